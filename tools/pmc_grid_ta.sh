#!/bin/bash
# texture-addresser / L1 counters of the grid search while the registration is still moving (1M x 1M, first 40 iterations)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-pmc_ta}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE -d $O/ta -o p -- python3 $R/tools/one_size.py 1000000 1000000 40 > $O/ta.out 2> $O/ta.err || { tail -5 $O/ta.err; exit 1; }
python3 $R/tools/rocprof_summary.py --pmc $O/ta/p_results.db | grep -A8 "search_grid_kernel\|cov_rec_kernel" > $O/pmc_ta.txt
rm -rf $O/ta
cat $O/pmc_ta.txt
