#!/bin/bash
# Round profile collection on the GPU box (run through gpurun): bench JSON, rocprofv3 kernel-trace stats of the same
# command, and the PMC passes (separate runs, counters only).  Summaries land in gpurun_out/<tag>/ as text.
set -e -o pipefail
TAG=${1:-v4}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SHORT="--no-cpu-baseline --no-fp64-variants"
timeout -k 10 500 python3 $R/bench.py > $O/bench.json 2> $O/bench.err
echo "bench done" > $O/progress.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 $R/bench.py --steps 10 --warmup 2 $SHORT > $O/trace_bench.json 2> $O/trace.err
python3 $R/tools/rocprof_summary.py $O/trace/t_results.db > $O/kernel_trace_stats.txt
echo "trace done" >> $O/progress.txt
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $O/pmc_sq -o p -- python3 $R/bench.py --steps 2 --warmup 1 $SHORT > /dev/null 2> $O/pmc_sq.err
python3 $R/tools/rocprof_summary.py --pmc $O/pmc_sq/p_results.db > $O/pmc_sq.txt
echo "pmc sq done" >> $O/progress.txt
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma -o p -- python3 $R/bench.py --steps 2 --warmup 1 $SHORT > /dev/null 2> $O/pmc_mfma.err
python3 $R/tools/rocprof_summary.py --pmc $O/pmc_mfma/p_results.db > $O/pmc_mfma.txt
echo "pmc mfma done" >> $O/progress.txt
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- python3 $R/bench.py --steps 2 --warmup 1 $SHORT > /dev/null 2> $O/pmc_fetch.err
python3 $R/tools/rocprof_summary.py --pmc $O/pmc_fetch/p_results.db > $O/pmc_fetch.txt
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o p -- python3 $R/bench.py --steps 2 --warmup 1 $SHORT > /dev/null 2> $O/pmc_write.err
python3 $R/tools/rocprof_summary.py --pmc $O/pmc_write/p_results.db > $O/pmc_write.txt
echo "pmc tcc done" >> $O/progress.txt
timeout -k 10 300 python3 -u $R/tools/grid_probe.py > $O/grid_probe.txt 2>&1
timeout -k 10 300 python3 -u $R/tools/small_probe.py > $O/small_probe.txt 2>&1
ICP_AMD_SEARCH=f32 timeout -k 10 300 python3 -u $R/tools/small_probe.py > $O/small_probe_f32.txt 2>&1
timeout -k 10 200 python3 $R/bench.py --workload batch --steps 5 --warmup 2 > $O/bench_batch.json 2> $O/bench_batch.err
echo "probes done" >> $O/progress.txt
rm -rf $O/trace $O/pmc_sq $O/pmc_mfma $O/pmc_fetch $O/pmc_write
