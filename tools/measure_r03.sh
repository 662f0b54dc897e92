# Round-3 measurement session (one gpurun call): bench under rocprofv3, PMC passes, default bench line, kernel table, config-5 step.
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/prof gpurun_out/r03
TAG=r03 bash tools/prof_bench.sh > gpurun_out/prof/r03_prof_bench.txt 2>&1
TAG=r03 bash tools/pmc_bench.sh > gpurun_out/prof/r03_pmc.txt 2>&1
cd $GRAFT_REPO_ROOT
( while true; do sleep 60; echo "[alive $(date +%T)]"; done ) & HB=$!
timeout -k 10 600 python3 bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err
N=12 timeout -k 10 300 python3 tools/bench_kernels.py > gpurun_out/r03/kernels.jsonl 2> gpurun_out/r03/kernels.err
timeout -k 10 300 python3 tools/bench_v5.py > gpurun_out/r03/v5_step.json 2> gpurun_out/r03/v5_step.err
kill $HB
tail -3 gpurun_out/prof/r03_prof_bench.txt | cut -c1-300; tail -8 gpurun_out/prof/r03_pmc.txt; cut -c1-400 gpurun_out/r03/bench_default.json; tail -2 gpurun_out/r03/v5_step.json | cut -c1-400
