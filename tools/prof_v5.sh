# per-kernel table of the config-5 step (YOLOv5-l 640^2 bs 32, bf16 autocast, MIOpen immediate mode) under rocprofv3
mkdir -p gpurun_out/prof
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_v5
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v5 -- python3 $ROOT/tools/bench_v5.py --steps 6 --warmup 3 > $ROOT/gpurun_out/prof/v5_bench.log 2> $ROOT/gpurun_out/prof/v5_bench.err
T=$(find /tmp/prof_v5 -name '*kernel_trace.csv' | head -1)
python3 $ROOT/tools/trace_summary.py $T --steps 6 --marker v5_finalize_kernel > $ROOT/gpurun_out/prof/r02_v5_steady_state.csv
head -45 $ROOT/gpurun_out/prof/r02_v5_steady_state.csv | cut -c1-150
