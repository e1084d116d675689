#!/bin/bash
# PMC passes over the grid iteration at 1M x 1M (counters only, one pass per run)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-pmc_grid}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() {  # name, counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d $O/$name -o p -- python3 $R/tools/one_size.py 1000000 1000000 40 > $O/$name.out 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python3 $R/tools/rocprof_summary.py --pmc $O/$name/p_results.db | grep -A14 "search_grid_kernel\|cov_kernel\|apply_kernel" > $O/pmc_$name.txt
  rm -rf $O/$name
}
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE && \
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS && \
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum && \
pass fetch FETCH_SIZE && pass write WRITE_SIZE
cat $O/pmc_*.txt
