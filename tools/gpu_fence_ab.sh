#!/bin/bash
# settled search without the device-scope fence in its walking path: tests, then old / new on the registration and on the
# small-cloud probe with the fused storage-order kernel forced (grid_order_switch=1e30) and by default
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-fence}
mkdir -p $O
cd $R
timeout -k 10 400 python -u -m pytest tests/test_gpu_grid.py tests/test_gpu_fullsize.py tests/test_gpu_hooks.py -m gpu -x -q --timeout 200 --timeout-method thread > $O/pytest.txt 2>&1
rc=$?
tail -3 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
for l in old new old new; do
  L=$R/icp_amd/libicp_amd.so; [ $l = old ] && L=$R/icp_amd/libicp_amd_old.so
  for sw in default 1e30; do
    echo "== $l switch=$sw" >> $O/ab.txt
    if [ $sw = default ]; then
      ICP_AMD_LIB=$L timeout -k 10 200 python -u tools/order_switch_probe.py >> $O/ab.txt 2>&1 || { tail $O/ab.txt; exit 1; }
      ICP_AMD_LIB=$L timeout -k 10 200 python -u tools/order_small_probe.py >> $O/ab.txt 2>&1 || { tail $O/ab.txt; exit 1; }
    else
      ICP_AMD_LIB=$L ICP_AMD_GRID_ORDER_SWITCH=$sw timeout -k 10 200 python -u tools/order_switch_probe.py >> $O/ab.txt 2>&1 || { tail $O/ab.txt; exit 1; }
      ICP_AMD_LIB=$L ICP_AMD_GRID_ORDER_SWITCH=$sw timeout -k 10 200 python -u tools/order_small_probe.py >> $O/ab.txt 2>&1 || { tail $O/ab.txt; exit 1; }
    fi
  done
done
cat $O/ab.txt
