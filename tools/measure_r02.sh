# Round-2 measurement session (one gpurun call): bench under rocprofv3 (fused and dense routes), PMC passes, kernel table.
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/prof
TAG=r02 bash tools/prof_bench.sh > gpurun_out/prof/r02_prof_bench.txt 2>&1
TAG=r02dense EXTRA="--no-fused-head-backward" bash tools/prof_bench.sh > gpurun_out/prof/r02dense_prof_bench.txt 2>&1
TAG=r02 bash tools/pmc_bench.sh > gpurun_out/prof/r02_pmc.txt 2>&1
TAG=r02dense EXTRA="--no-fused-head-backward" bash tools/pmc_bench.sh > gpurun_out/prof/r02dense_pmc.txt 2>&1
N=12 python3 tools/bench_kernels.py > gpurun_out/prof/r02_kernels.jsonl 2> gpurun_out/prof/r02_kernels.err
bash tools/pmc_kernels.sh > gpurun_out/prof/r02_pmc_kernels.txt 2>&1
python3 tools/bench_retina_step.py > gpurun_out/prof/r02_retina_step.json 2> gpurun_out/prof/r02_retina_step.err
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/prof/r02_prof_bench.txt | cut -c1-400; cat gpurun_out/prof/r02_pmc.txt | tail -12; cat gpurun_out/prof/r02dense_pmc.txt | tail -12; cat gpurun_out/prof/r02_kernels.jsonl | cut -c1-260
