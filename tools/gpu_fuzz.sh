#!/bin/bash
# the differential fuzzers on the tree's library (tests/fuzz), each under its own limit; stops at the first failure
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-fuzz}
mkdir -p $O
cd $R
timeout -k 10 300 python -u -m pytest tests/test_gpu_fullsize.py -m gpu -x -q --timeout 280 --timeout-method thread > $O/fullsize.txt 2>&1 || { tail -20 $O/fullsize.txt; exit 1; }
tail -2 $O/fullsize.txt
run() { # name seconds seed
  timeout -k 10 $(( $2 + 120 )) python -u tests/fuzz/$1.py $2 $3 > $O/$1.txt 2>&1 || { echo "$1 FAILED"; tail -8 $O/$1.txt; exit 1; }
  echo "$1: $(tail -1 $O/$1.txt)"
}
S=${2:-20}
run fuzz_variants 200 $((S+1)) && run fuzz_big 260 $((S+2)) && run fuzz_sharded 100 $((S+3)) && run fuzz_hooks 60 $((S+4)) && run fuzz_batch 60 $((S+5))
