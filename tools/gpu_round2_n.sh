#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r2n}
mkdir -p $O
cd $R
timeout -k 10 400 python -u -m pytest tests/test_gpu_grid.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest_grid.txt 2>&1
rc=$?
tail -15 $O/pytest_grid.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 420 python -u tests/fuzz/fuzz_variants.py 300 61 > $O/fuzz_variants.txt 2>&1 || { echo FUZZ FAILED; tail -8 $O/fuzz_variants.txt; exit 1; }
tail -1 $O/fuzz_variants.txt
