#!/bin/bash
# the settled-registration threshold (ICP_AMD_DEBUG=grid_order_switch, in cell half-edges) against the whole 1M x 1M registration
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-order_sweep}
mkdir -p $O
cd $R
for v in 0.125 0.5 1 2 4 8 1e30 0.125; do
  ICP_AMD_DEBUG="grid_order_switch=$v" ICP_AMD_GRID_ORDER_SWITCH=$v timeout -k 10 200 python -u tools/order_switch_probe.py >> $O/sweep.txt 2>&1 || { tail $O/sweep.txt; exit 1; }
done
cat $O/sweep.txt
