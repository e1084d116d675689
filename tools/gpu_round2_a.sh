#!/bin/bash
# first GPU call of round 2: GPU suite, the allocation-free entry point's test, the MFMA loop microbenchmark, a short bench
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2a
mkdir -p $O
cd $R
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $O/pytest.txt 2>&1; echo "pytest rc $?" | tee -a $O/progress.txt
timeout -k 10 120 python -u -m pytest tests/wip_no_alloc.py -m gpu -x -v --timeout 100 --timeout-method thread > $O/wip.txt 2>&1; echo "wip rc $?" | tee -a $O/progress.txt
timeout -k 10 200 $R/tools/ubench_mfma > $O/ubench_mfma.txt 2>&1; echo "ubench rc $?" | tee -a $O/progress.txt
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" | tee -a $O/progress.txt
tail -5 $O/pytest.txt; tail -5 $O/wip.txt
