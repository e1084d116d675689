#!/bin/bash
# A/B of the grid search between icp_amd/libicp_amd_old.so (see tools/gpu_ab_lib.sh for how it is built) and the tree's library:
# grid tests on the tree's, then the whole 1M x 1M registration, the surface clouds and all sizes of grid_probe per library
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-grid_ab}
mkdir -p $O
cd $R
timeout -k 10 400 python -u -m pytest tests/test_gpu_grid.py tests/test_gpu_fullsize.py tests/test_gpu_hooks.py -m gpu -x -q --timeout 200 --timeout-method thread > $O/pytest.txt 2>&1
rc=$?
tail -4 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
for l in old new old new; do
  L=$R/icp_amd/libicp_amd.so; [ $l = old ] && L=$R/icp_amd/libicp_amd_old.so
  echo "== $l" >> $O/ab.txt
  ICP_AMD_LIB=$L timeout -k 10 200 python -u tools/order_switch_probe.py >> $O/ab.txt 2>&1 || { tail $O/ab.txt; exit 1; }
  ICP_AMD_LIB=$L timeout -k 10 300 python -u tools/surface_probe.py >> $O/ab.txt 2>&1 || { tail $O/ab.txt; exit 1; }
  ICP_AMD_LIB=$L timeout -k 10 300 python -u tools/grid_probe.py 2>&1 | grep "grid" >> $O/ab.txt || { tail $O/ab.txt; exit 1; }
done
cat $O/ab.txt
