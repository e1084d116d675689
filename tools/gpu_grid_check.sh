#!/bin/bash
# the grid search after a change: its tests, then the per-ten-iteration timing of a 1M x 1M registration and the short bench
# usage: tools/gpu_grid_check.sh <tag>
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-gridcheck}
shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_grid.py tests/test_gpu_hooks.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q --timeout 200 --timeout-method thread "$@" > $O/pytest.txt 2>&1
rc=$?
tail -8 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python -u tools/order_switch_probe.py > $O/probe.txt 2>&1 || { tail $O/probe.txt; exit 1; }
cat $O/probe.txt
ICP_AMD_GRID_STATS=1 timeout -k 10 200 python -u tools/one_size.py 1000000 1000000 12 > $O/stats.txt 2>&1 || { tail $O/stats.txt; exit 1; }
tail -14 $O/stats.txt
timeout -k 10 300 python bench.py --no-cpu-baseline --no-fp64-variants --no-batch > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/bench.json"))
g=d["grid_search"]
print("settled", g["ms_per_step"], g["search_kernel_avg_ms"], "first_steps", g["first_steps"]["ms_per_step"], g["first_steps"]["ms_per_step_by_10"], "first", g["first_step_ms"])
PY
