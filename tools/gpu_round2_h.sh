#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r2h}
mkdir -p $O
cd $R
for sw in 0 0.03 0.125 0.5 2 1e30; do
  ICP_AMD_GRID_ORDER_SWITCH=$sw timeout -k 10 120 python -u tools/order_switch_probe.py >> $O/order_switch.txt 2>&1 || { tail $O/order_switch.txt; exit 1; }
done
ICP_AMD_GRID_ORDER=0 timeout -k 10 120 python -u tools/order_switch_probe.py >> $O/order_switch.txt 2>&1 || exit 1
cat $O/order_switch.txt
