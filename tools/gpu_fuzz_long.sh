#!/bin/bash
# longer differential fuzz of the search variants on the tree's library: fuzz_big, then fuzz_variants (seeds from $2)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-fuzz_long}
S=${2:-70}
mkdir -p $O
cd $R
timeout -k 10 720 python -u tests/fuzz/fuzz_big.py 600 $((S+1)) > $O/fuzz_big.txt 2>&1 || { echo "fuzz_big FAILED"; tail -8 $O/fuzz_big.txt; exit 1; }
echo "fuzz_big: $(tail -1 $O/fuzz_big.txt)"
timeout -k 10 420 python -u tests/fuzz/fuzz_variants.py 300 $((S+2)) > $O/fuzz_variants.txt 2>&1 || { echo "fuzz_variants FAILED"; tail -8 $O/fuzz_variants.txt; exit 1; }
echo "fuzz_variants: $(tail -1 $O/fuzz_variants.txt)"
