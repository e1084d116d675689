#!/bin/bash
# PMC pass over the brute-force search on one size (counters only): matrix-pipe busy cycles beside the usual SQ set.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_mfma
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export ICP_AMD_SEARCH=f32
for G in ${GROUPS_LIST:-4 0}; do
  export ICP_AMD_MFMA=$G   # 16 = the fp16 K=16 filter (groups: ICP_AMD_MFMA_G)
  timeout -s KILL 150 rocprofv3 --pmc ${COUNTERS:-SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE} -d $O/g$G -o p -- python3 $R/tools/one_size.py 1000000 1000000 3 > $O/g$G.out 2> $O/g$G.err
  python3 $R/tools/rocprof_summary.py --pmc $O/g$G/p_results.db | grep -A12 "search_" > $O/pmc_g$G.txt
  rm -rf $O/g$G
  echo "G=$G done"
done
