#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-phase}
mkdir -p $O
cd $R
for a in "48485 20" "100000 30" "1000000 20"; do
  timeout -k 10 120 python -u tools/phase_probe.py $a >> $O/phase.txt 2>&1 || { tail $O/phase.txt; exit 1; }
done
cat $O/phase.txt
timeout -k 10 200 python -u tools/e2e_probe.py > $O/e2e.txt 2>&1 || { tail $O/e2e.txt; exit 1; }
cat $O/e2e.txt
