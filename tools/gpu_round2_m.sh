#!/bin/bash
# full GPU suite + batch bench + grid probe (after a change to the shared solve code)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r2m}
mkdir -p $O
cd $R
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread --durations=5 > $O/pytest.txt 2>&1
rc=$?
tail -4 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u tools/grid_probe.py > $O/grid_probe.txt 2>&1 || { tail $O/grid_probe.txt; exit 1; }
grep grid $O/grid_probe.txt
timeout -k 10 200 python bench.py --workload batch --steps 5 --warmup 2 > $O/bench_batch.json 2> $O/bench_batch.err || { tail -5 $O/bench_batch.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/bench_batch.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d.get("roofline",{}).get("frac"))
PY
