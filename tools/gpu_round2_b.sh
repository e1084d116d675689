#!/bin/bash
# A/B of the matrix-filter kernel builds (icp_amd/alt), then the GPU suite on the tree's own library
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2b
mkdir -p $O
cd $R
cp icp_amd/libicp_amd.so /tmp/libicp_amd_main.so
for v in L0_R64 L1_R64 L0_R128 L1_R128 L0_R64 L1_R64; do
  cp icp_amd/alt/libicp_amd_$v.so icp_amd/libicp_amd.so || exit 1
  timeout -k 10 120 python tools/ab_probe.py 1000000 1000000 8 $v >> $O/ab.txt 2>&1 || { echo "ab $v failed"; tail -5 $O/ab.txt; cp /tmp/libicp_amd_main.so icp_amd/libicp_amd.so; exit 1; }
done
cp /tmp/libicp_amd_main.so icp_amd/libicp_amd.so
cat $O/ab.txt
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest.txt 2>&1
rc=$?
tail -15 $O/pytest.txt
exit $rc
