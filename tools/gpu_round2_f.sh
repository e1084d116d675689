#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r2f}
mkdir -p $O
cd $R
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread --durations=5 > $O/pytest.txt 2>&1
rc=$?
tail -12 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u tools/grid_probe.py > $O/grid_probe.txt 2>&1 || { tail $O/grid_probe.txt; exit 1; }
ICP_AMD_FUSE=0 timeout -k 10 300 python -u tools/grid_probe.py > $O/grid_probe_nofuse.txt 2>&1 || exit 1
grep grid $O/grid_probe.txt; grep grid $O/grid_probe_nofuse.txt
timeout -k 10 300 python bench.py --no-cpu-baseline --no-fp64-variants > $O/bench_short.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/bench_short.json"))
print({k:d[k] for k in ("value","ms_per_step","first_step_ms")}, {k:d["grid_search"][k] for k in ("ms_per_step","ms_per_step_by_10","first_step_ms")})
PY
