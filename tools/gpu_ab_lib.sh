#!/bin/bash
# A/B of two builds of the library on one box: icp_amd/libicp_amd_old.so (ICP_AMD_LIB) against the tree's own -- the batch
# bench, the small-cloud probe, a 1M x 1M registration by tens of iterations.   usage: tools/gpu_ab_lib.sh <tag>
# The baseline is built beforehand, here on the CPU box, from the commit to compare against:
#   git worktree add /tmp/base <commit> && make -C /tmp/base icp_amd/libicp_amd.so && cp /tmp/base/icp_amd/libicp_amd.so icp_amd/libicp_amd_old.so
# (git-ignored; it travels to the GPU box with the tree)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-ab_lib}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -m gpu -x -q --timeout 200 --timeout-method thread > $O/pytest.txt 2>&1
rc=$?
tail -4 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
for lib in old new old new; do
  L=$R/icp_amd/libicp_amd.so; [ $lib = old ] && L=$R/icp_amd/libicp_amd_old.so
  echo "== $lib" >> $O/ab.txt
  ICP_AMD_LIB=$L timeout -k 10 200 python bench.py --workload batch --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch ms_per_step', d['ms_per_step'], 'reg/s', d.get('registrations_per_sec'))" >> $O/ab.txt || { tail $O/ab.txt; exit 1; }
  ICP_AMD_LIB=$L timeout -k 10 200 python -u tools/small_probe.py >> $O/ab.txt 2>&1 || { tail $O/ab.txt; exit 1; }
  ICP_AMD_LIB=$L timeout -k 10 200 python -u tools/order_switch_probe.py >> $O/ab.txt 2>&1 || { tail $O/ab.txt; exit 1; }
done
cat $O/ab.txt
