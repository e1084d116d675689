#!/bin/bash
# PMC passes over the first 40 grid iterations at 1M x 1M (the clouds still move) for several builds / settings of the search
# usage: tools/pmc_grid_ab.sh <tag> "<name>:<VAR=val ...>" ...
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-pmc_ab}
shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() {  # variant-name, env settings, pass-name, counters...
  local vn=$1 cfg=$2 name=$3; shift 3
  ( export $cfg; timeout -s KILL 120 rocprofv3 --pmc "$@" -d $O/$vn.$name -o p -- python3 $R/tools/one_size.py 1000000 1000000 40 > $O/$vn.$name.out 2> $O/$vn.$name.err ) || { tail -5 $O/$vn.$name.err; return 1; }
  python3 $R/tools/rocprof_summary.py --pmc $O/$vn.$name/p_results.db | grep -A14 "search_grid_kernel" > $O/pmc_${vn}_$name.txt
  rm -rf $O/$vn.$name
}
for spec in "$@"; do
  vn=${spec%%:*}; cfg=${spec#*:}
  pass $vn "$cfg" sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE || exit 1
  pass $vn "$cfg" sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM || exit 1
  pass $vn "$cfg" ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE || exit 1
done
for f in $O/pmc_*.txt; do echo "#### $f"; cat $f; done
