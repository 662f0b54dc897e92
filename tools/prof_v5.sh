# rocprofv3 kernel trace of the config-5 step (tools/bench_v5.py); per-kernel totals of the whole run, top 40
cd /tmp && export TMPDIR=/tmp
( while true; do sleep 60; echo "[alive $(date +%T)]"; done ) & HB=$!
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v5 -- python3 $GRAFT_REPO_ROOT/tools/bench_v5.py --steps 10 --warmup 3 $EXTRA > $GRAFT_REPO_ROOT/gpurun_out/prof_v5_bench.log 2>&1
kill $HB
T=$(find /tmp/prof_v5 -name '*kernel_trace.csv' | head -1)
python3 $GRAFT_REPO_ROOT/tools/trace_summary.py $T --steps 10 --marker v5_finalize_kernel --top 45 > $GRAFT_REPO_ROOT/gpurun_out/prof_v5_steady_state.csv
cut -c1-150 $GRAFT_REPO_ROOT/gpurun_out/prof_v5_steady_state.csv
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_v5_bench.log | cut -c1-300
