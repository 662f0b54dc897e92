# per-kernel table of the config-4 step (RetinaNet-R50-FPN 800x1333 bs 4, fp32, MIOpen immediate mode) under rocprofv3
mkdir -p gpurun_out/prof
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_r4
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r4 -- python3 $ROOT/tools/bench_retina_step.py --steps 6 --warmup 3 $EXTRA > $ROOT/gpurun_out/prof/r4_bench.log 2> $ROOT/gpurun_out/prof/r4_bench.err
T=$(find /tmp/prof_r4 -name '*kernel_trace.csv' | head -1)
python3 $ROOT/tools/trace_summary.py $T --steps 6 --marker retina_finalize_kernel > $ROOT/gpurun_out/prof/${TAG:-r03}_retina_steady_state.csv
head -4 $ROOT/gpurun_out/prof/${TAG:-r03}_retina_steady_state.csv | cut -c1-150
grep -v "igemm\|^ck::\|Cijk\|^#\|^kernel" $ROOT/gpurun_out/prof/${TAG:-r03}_retina_steady_state.csv | head -40 | cut -c1-150
