#!/bin/bash
# the GPU suite K times in a row (fresh process each), to catch tests that sit on a threshold; stops at the first failure
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-repeat}
K=${2:-3}
mkdir -p $O
cd $R
for k in $(seq 1 $K); do
  timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -p no:cacheprovider > $O/pytest_$k.txt 2>&1
  rc=$?
  tail -3 $O/pytest_$k.txt
  [ $rc -ne 0 ] && { grep -n "Error\|FAILED" $O/pytest_$k.txt | head; exit $rc; }
done
exit 0
