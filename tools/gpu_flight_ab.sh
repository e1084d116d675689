#!/bin/bash
# record loads in flight per walking lane (ICP_AMD_DEBUG=grid_flight=4|6|8): grid tests, then volumes (whole 1M x 1M registration,
# all sizes of grid_probe) and surfaces (surface_probe) per setting
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-flight}
mkdir -p $O
cd $R
timeout -k 10 400 python -u -m pytest tests/test_gpu_grid.py tests/test_gpu_fullsize.py -m gpu -x -q --timeout 200 --timeout-method thread > $O/pytest.txt 2>&1
rc=$?
tail -4 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
for f in 4 6 8 4 6; do
  echo "== grid_flight=$f" >> $O/ab.txt
  ICP_AMD_DEBUG="grid_flight=$f" timeout -k 10 200 python -u tools/order_switch_probe.py >> $O/ab.txt 2>&1 || { tail $O/ab.txt; exit 1; }
  ICP_AMD_DEBUG="grid_flight=$f" timeout -k 10 300 python -u tools/surface_probe.py >> $O/ab.txt 2>&1 || { tail $O/ab.txt; exit 1; }
  ICP_AMD_DEBUG="grid_flight=$f" timeout -k 10 300 python -u tools/grid_probe.py 2>&1 | grep "grid" >> $O/ab.txt || { tail $O/ab.txt; exit 1; }
done
cat $O/ab.txt
