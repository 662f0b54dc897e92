#!/bin/bash
# Round-4 measurement pass (one gpurun call): default bench line (cpu_baseline + secondary configs), the RCCL path at world size 1,
# the kernel table, PMC traffic of the loss-path kernels, SQ counters of the instruction-bound kernels, the copy yardstick under rocprofv3.
mkdir -p gpurun_out/final
( while true; do sleep 60; echo "[alive $(date +%T)]"; done ) & HB=$!
trap "kill $HB" EXIT
set -x
timeout -k 10 600 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err || exit 1
timeout -k 10 300 python bench.py --force-ddp --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > gpurun_out/final/bench_force_ddp.json 2> gpurun_out/final/bench_force_ddp.err || exit 1
timeout -k 10 300 python bench.py --torch-adam --no-cpu-baseline --no-secondary --steps 30 > gpurun_out/final/bench_torch_adam.json 2> gpurun_out/final/bench_torch_adam.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 30 > gpurun_out/final/bench_again.json 2> gpurun_out/final/bench_again.err || exit 1
timeout -k 10 400 python tools/bench_kernels.py > gpurun_out/final/kernels.jsonl 2> gpurun_out/final/kernels.err || exit 1
TAG=r04 bash tools/pmc_bench.sh > gpurun_out/final/pmc_bench.log 2>&1 || exit 1
bash tools/pmc_instr.sh > gpurun_out/final/pmc_instr.log 2>&1 || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probe_copy.hip -o /tmp/probe_copy || exit 1
rm -rf /tmp/prof_copy; ( cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_copy -- /tmp/probe_copy > $GRAFT_REPO_ROOT/gpurun_out/final/probe_copy.csv 2> $GRAFT_REPO_ROOT/gpurun_out/final/probe_copy.err )
F=$(find /tmp/prof_copy -name '*kernel_stats.csv' 2>/dev/null | sort | sed -n 1p)
test -n "$F" && cp "$F" gpurun_out/final/probe_copy_rocprof_stats.csv
set +x
cut -c1-300 gpurun_out/final/bench_default.json
