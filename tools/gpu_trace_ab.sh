#!/bin/bash
# per-kernel times of the first N iterations of the 1M x 1M registration, icp_amd/libicp_amd_old.so against the tree's library
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-trace_ab}
N=${2:-40}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for l in old new; do
  L=$R/icp_amd/libicp_amd.so; [ $l = old ] && L=$R/icp_amd/libicp_amd_old.so
  export ICP_AMD_LIB=$L
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/$l -o t -- python3 $R/tools/one_size.py 1000000 1000000 $N > $O/$l.log 2>&1 || { tail $O/$l.log; exit 1; }
  f=$(find $O/$l -name "*kernel_stats.csv" | head -1)
  echo "== $l" >> $O/summary.txt
  python3 - "$f" >> $O/summary.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-60s calls %5s  avg %9.1f us  total %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
done
cat $O/summary.txt
find $O -name "*.csv" -size +1M -delete
