#!/bin/bash
# per-ten-iteration timing of a 1M x 1M registration for a list of "VAR=val VAR=val" settings (one process each)
# usage: tools/gpu_grid_matrix.sh <tag> "<settings 1>" "<settings 2>" ...
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-matrix}
shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
: > $O/matrix.txt
for cfg in "$@"; do
  echo "== $cfg" >> $O/matrix.txt
  env $cfg timeout -k 10 120 python -u tools/order_switch_probe.py >> $O/matrix.txt 2>&1 || { tail $O/matrix.txt; exit 1; }
done
cat $O/matrix.txt
