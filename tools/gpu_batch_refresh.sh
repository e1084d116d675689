#!/bin/bash
# After a change to the multi-registration kernel (through gpurun): its tests, a short differential fuzz, the FETCH / WRITE /
# SQ counter passes, the traffic record (icp_amd/measured_traffic.json, returned as gpurun_out/<tag>/measured_traffic.json)
# and the two bench lines that quote it.    bash tools/gpu_batch_refresh.sh r3_v9
set -o pipefail
TAG=${1:-batch_refresh}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O $R/profiles
cd $R
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_batch.py -x -q --timeout 200 --timeout-method thread > $O/batch_tests.txt 2>&1 || { tail -n 20 $O/batch_tests.txt; exit 1; }
timeout -k 10 200 python3 -u tests/fuzz/fuzz_batch.py 60 2718 > $O/fuzz_batch.txt 2>&1 || { tail -n 5 $O/fuzz_batch.txt; exit 1; }
echo "tests + fuzz done" > $O/progress.txt
bash tools/pmc_batch.sh $TAG/batch > /dev/null || exit 1
cp $O/batch/pmc_FETCH_SIZE.txt profiles/${TAG}_batch_pmc_fetch.txt
cp $O/batch/pmc_WRITE_SIZE.txt profiles/${TAG}_batch_pmc_write.txt
cp $O/batch/pmc_FETCH_SIZE.txt $O/batch_pmc_fetch.txt
cp $O/batch/pmc_WRITE_SIZE.txt $O/batch_pmc_write.txt
rm -rf $O/batch
python3 tools/update_traffic.py --batch profiles/${TAG}_batch_pmc_fetch.txt profiles/${TAG}_batch_pmc_write.txt 4096 1000 20 > /dev/null || exit 1
cp icp_amd/measured_traffic.json $O/measured_traffic.json
echo "pmc tcc done" >> $O/progress.txt
( cd /tmp && export TMPDIR=/tmp && timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/sq -o p -- python3 $R/bench.py --workload batch --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/sq.err ) || { tail -n 5 $O/sq.err; exit 1; }
python3 tools/rocprof_summary.py --pmc $O/sq/p_results.db | grep -A10 "batch_icp_kernel" | cut -c1-200 > $O/batch_pmc_sq.txt
rm -rf $O/sq
echo "pmc sq done" >> $O/progress.txt
timeout -k 10 200 python3 bench.py --workload batch --steps 5 --warmup 2 > $O/bench_batch.json 2> $O/bench_batch.err || exit 1
echo "bench batch done" >> $O/progress.txt
timeout -k 10 500 python3 bench.py > $O/bench.json 2> $O/bench.err || exit 1
echo "all done" >> $O/progress.txt
tail -n 1 $O/batch_tests.txt $O/fuzz_batch.txt
