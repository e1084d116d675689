#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the multi-registration kernel (bench.py --workload batch), one counter per run
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-pmc_batch}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c -d $O/$c -o p -- python3 $R/bench.py --workload batch --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/$c.err || { tail -5 $O/$c.err; exit 1; }
  python3 $R/tools/rocprof_summary.py --pmc $O/$c/p_results.db > $O/pmc_$c.txt
  rm -rf $O/$c
done
grep -A2 "batch_icp_kernel" $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt | cut -c1-200
