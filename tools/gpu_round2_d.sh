#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2d
mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_grid.py tests/test_gpu_hooks.py tests/test_gpu_sharded.py tests/test_gpu_fullsize.py -m gpu -x -q --timeout 300 --timeout-method thread --durations=5 > $O/pytest.txt 2>&1
rc=$?
tail -25 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u tools/grid_probe.py > $O/grid_probe.txt 2>&1 || { tail $O/grid_probe.txt; exit 1; }
ICP_AMD_GRID_CERT=0 timeout -k 10 300 python -u tools/grid_probe.py > $O/grid_probe_nocert.txt 2>&1 || exit 1
timeout -k 10 200 python -u tools/early_iters.py > $O/early.txt 2>&1 || { tail $O/early.txt; exit 1; }
cat $O/grid_probe.txt $O/grid_probe_nocert.txt $O/early.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 $R/tools/one_size.py 1000000 1000000 60 > $O/one_size.txt 2>&1 || { tail $O/one_size.txt; exit 1; }
python3 $R/tools/rocprof_summary.py $O/trace/t_results.db > $O/kernel_trace_grid.txt
python3 $R/tools/rocprof_summary.py --timeline $O/trace/t_results.db > $O/timeline_grid.txt
rm -rf $O/trace
cat $O/kernel_trace_grid.txt
