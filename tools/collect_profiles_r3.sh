#!/bin/bash
# Round-3 profile set on the GPU box (through gpurun): everything tools/collect_profiles.sh gathers, plus the counters of the
# settled grid search and of the batch kernel, and the stream microbenchmark.  Summaries land in gpurun_out/<tag>/.
set -o pipefail
TAG=${1:-r3_v1}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
bash $R/tools/collect_profiles.sh $TAG || exit 1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c -d $O/g$c -o p -- python3 $R/tools/one_size.py 1000000 1000000 400 > /dev/null 2> $O/grid_$c.err || { tail -5 $O/grid_$c.err; exit 1; }
  python3 $R/tools/rocprof_summary.py --pmc $O/g$c/p_results.db | grep -A3 "search_grid_settled_kernel\|search_grid_kernel\|apply_kernel\|cov_rec_kernel" > $O/grid_pmc_$c.txt
  rm -rf $O/g$c
done
echo "grid pmc done" >> $O/progress.txt
bash $R/tools/pmc_grid_settled.sh $TAG/settled > /dev/null || exit 1
cp $O/settled/kernel_trace_stats.txt $O/grid_settled_kernel_trace_stats.txt
cp $O/settled/pmc_sq.txt $O/grid_settled_pmc_sq.txt
rm -rf $O/settled
bash $R/tools/pmc_batch.sh $TAG/batch > /dev/null || exit 1
cp $O/batch/pmc_FETCH_SIZE.txt $O/batch_pmc_fetch.txt
cp $O/batch/pmc_WRITE_SIZE.txt $O/batch_pmc_write.txt
rm -rf $O/batch
echo "batch pmc done" >> $O/progress.txt
cd $R
timeout -k 10 120 ./tools/ubench_stream > $O/ubench_stream.txt 2>&1 || exit 1
timeout -k 10 200 python3 -u tools/order_switch_probe.py > $O/registration_probe.txt 2>&1 || exit 1
echo "all done" >> $O/progress.txt
ls $O
