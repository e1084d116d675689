#!/bin/bash
# full GPU suite, grid probes (also with the spatial order off), short bench, kernel trace of the grid iteration
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r2g}
mkdir -p $O
cd $R
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread --durations=5 > $O/pytest.txt 2>&1
rc=$?
tail -12 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u tools/grid_probe.py > $O/grid_probe.txt 2>&1 || { tail $O/grid_probe.txt; exit 1; }
ICP_AMD_GRID_ORDER=0 timeout -k 10 300 python -u tools/grid_probe.py > $O/grid_probe_noorder.txt 2>&1 || { tail $O/grid_probe_noorder.txt; exit 1; }
timeout -k 10 200 python -u tools/early_iters.py > $O/early.txt 2>&1 || exit 1
grep grid $O/grid_probe.txt; echo "-- order off"; grep grid $O/grid_probe_noorder.txt; grep cube $O/early.txt
timeout -k 10 300 python bench.py --no-cpu-baseline --no-fp64-variants > $O/bench_short.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/bench_short.json"))
print({k:d[k] for k in ("value","ms_per_step","first_step_ms")}, {k:d["grid_search"][k] for k in ("ms_per_step","ms_per_step_by_10","first_step_ms")})
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o grid -- python3 $R/tools/grid_probe.py > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
cd $R
f=$(ls $O/trace/*kernel_stats.csv $O/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cut -d, -f1-7 "$f" | head -24 | tee $O/grid_kernel_stats.txt
