#!/bin/bash
# GPU suite + default bench on the tree's library; each step under its own limit, nothing runs after a failed step
# usage: tools/gpu_suite.sh <tag> [extra pytest arguments, e.g. --deselect ...]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-suite}
shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 300 --timeout-method thread --durations=12 "$@" > $O/pytest.txt 2>&1
rc=$?
tail -25 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 100 python -u -m pytest tests/test_zz_no_alloc.py -m gpu -x -q --timeout 90 --timeout-method thread > $O/noalloc_alone.txt 2>&1 || { tail -20 $O/noalloc_alone.txt; exit 1; }
tail -2 $O/noalloc_alone.txt
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","first_step_ms")}, d["roofline"]["avg_launch_ms"], d["grid_search"]["ms_per_step"],
      d["grid_search"]["first_steps"]["ms_per_step"], d.get("batch_mode", {}).get("registrations_per_sec"))
PY
