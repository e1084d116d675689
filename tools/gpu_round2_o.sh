#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r2o}
mkdir -p $O
cd $R
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest.txt 2>&1
rc=$?
tail -4 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 python -u tools/phase_probe.py 1000000 20 2>&1 | tee $O/phase.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 $R/tools/phase_probe.py 1000000 20 > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
python3 $R/tools/rocprof_summary.py $O/trace/t_results.db | head -30 | tee $O/phase_kernels.txt
rm -rf $O/trace
