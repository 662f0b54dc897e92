# end-of-round measurement pass: default bench line (with cpu_baseline), RCCL path at world size 1, kernel table
mkdir -p gpurun_out/final
( while true; do sleep 60; echo "[alive $(date +%T)]"; done ) & HB=$!
timeout -k 10 600 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err && \
timeout -k 10 300 python bench.py --force-ddp --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/final/bench_force_ddp.json 2> gpurun_out/final/bench_force_ddp.err && \
timeout -k 10 300 python tools/bench_kernels.py > gpurun_out/final/kernels.jsonl 2> gpurun_out/final/kernels.err
RC=$?
kill $HB
cut -c1-600 gpurun_out/final/bench_default.json; cut -c1-200 gpurun_out/final/bench_force_ddp.json
exit $RC
