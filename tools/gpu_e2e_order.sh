#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-e2e_order}
mkdir -p $O
cd $R
timeout -k 10 200 python -u tools/e2e_probe.py > $O/e2e_order1.txt 2>&1 || { tail $O/e2e_order1.txt; exit 1; }
ICP_AMD_GRID_ORDER=0 timeout -k 10 200 python -u tools/e2e_probe.py > $O/e2e_order0.txt 2>&1 || { tail $O/e2e_order0.txt; exit 1; }
grep auto $O/e2e_order1.txt; echo "-- ICP_AMD_GRID_ORDER=0"; grep auto $O/e2e_order0.txt
