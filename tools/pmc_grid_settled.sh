#!/bin/bash
# kernel trace + SQ counters of a long 1M x 1M registration (400 iterations: most of them settled)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-pmc_settled}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 $R/tools/one_size.py 1000000 1000000 400 > $O/trace.out 2> $O/trace.err || { tail -5 $O/trace.err; exit 1; }
python3 $R/tools/rocprof_summary.py $O/trace/t_results.db > $O/kernel_trace_stats.txt
rm -rf $O/trace
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/sq -o p -- python3 $R/tools/one_size.py 1000000 1000000 400 > $O/sq.out 2> $O/sq.err || { tail -5 $O/sq.err; exit 1; }
python3 $R/tools/rocprof_summary.py --pmc $O/sq/p_results.db | grep -A12 "search_grid_settled_kernel\|apply_kernel" > $O/pmc_sq.txt
rm -rf $O/sq
cat $O/kernel_trace_stats.txt | head -30
cat $O/pmc_sq.txt
