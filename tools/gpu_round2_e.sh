#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r2e}
mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_grid.py tests/test_gpu_hooks.py tests/test_gpu_sharded.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread --durations=5 > $O/pytest.txt 2>&1
rc=$?
tail -12 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u tools/grid_probe.py > $O/grid_probe.txt 2>&1 || { tail $O/grid_probe.txt; exit 1; }
timeout -k 10 200 python -u tools/early_iters.py > $O/early.txt 2>&1 || { tail $O/early.txt; exit 1; }
grep grid $O/grid_probe.txt; cat $O/early.txt
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD -d $O/sq -o p -- python3 $R/tools/one_size.py 1000000 1000000 40 > $O/sq.out 2> $O/sq.err || { tail -5 $O/sq.err; exit 1; }
python3 $R/tools/rocprof_summary.py --pmc $O/sq/p_results.db | grep -A9 "search_grid_kernel" > $O/pmc_sq.txt
rm -rf $O/sq
cat $O/pmc_sq.txt
