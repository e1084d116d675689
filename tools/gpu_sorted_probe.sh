#!/bin/bash
# tools/sorted_source_probe.py under its variants (unsorted baseline; source sorted on the host with the engine's own permutation on / off; the fused storage-order kernel forced), old library by default
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-sorted}
mkdir -p $O
cd $R
export ICP_AMD_LIB=${PROBE_LIB:-$R/icp_amd/libicp_amd_old.so}
run() { timeout -k 10 200 env "$@" python -u tools/sorted_source_probe.py >> $O/out.txt 2>&1 || { tail $O/out.txt; exit 1; }; }
run PROBE_SORT=0
run PROBE_SORT=1 ICP_AMD_GRID_ORDER=0
run PROBE_SORT=1 ICP_AMD_GRID_ORDER=0 ICP_AMD_GRID_ORDER_SWITCH=1e30
run PROBE_SORT=1 ICP_AMD_GRID_ORDER=1
run PROBE_SORT=0
run PROBE_SORT=1 ICP_AMD_GRID_ORDER=0
run PROBE_SORT=1 ICP_AMD_GRID_ORDER=0 ICP_AMD_GRID_ORDER_SWITCH=1e30
cat $O/out.txt
