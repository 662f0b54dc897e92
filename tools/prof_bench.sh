# rocprofv3 kernel-trace of the bench command; steady-state summary -> gpurun_out/prof/<TAG>_*.csv
TAG=${TAG:-default}
mkdir -p gpurun_out/prof
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
( while true; do sleep 60; echo "[alive $(date +%T)]"; done ) &
HB=$!
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_raw_$TAG -- python3 $ROOT/bench.py --steps ${STEPS:-10} --warmup ${WARMUP:-3} --no-cpu-baseline --no-secondary $EXTRA > $ROOT/gpurun_out/prof/${TAG}_bench.log 2> $ROOT/gpurun_out/prof/${TAG}_bench.err
echo exit=$? >> $ROOT/gpurun_out/prof/${TAG}_bench.log
kill $HB
T=$(find /tmp/prof_raw_$TAG -name '*kernel_trace.csv' | head -1)
python3 $ROOT/tools/trace_summary.py $T --steps ${STEPS:-10} > $ROOT/gpurun_out/prof/${TAG}_steady_state.csv
S=$(find /tmp/prof_raw_$TAG -name '*kernel_stats.csv' | head -1)
grep -E "yolo_|headgrad_|^\"Name\"" $S | sed -E 's/\(anonymous namespace\):://; s/_ZN12_GLOBAL__N_1[0-9]+//' > $ROOT/gpurun_out/prof/${TAG}_rocprof_stats_yolo.csv
head -14 $ROOT/gpurun_out/prof/${TAG}_steady_state.csv | cut -c1-150
grep -E "yolo|headgrad|Cijk|direct_copy" $ROOT/gpurun_out/prof/${TAG}_steady_state.csv | cut -c1-150
tail -2 $ROOT/gpurun_out/prof/${TAG}_bench.log | cut -c1-1200
