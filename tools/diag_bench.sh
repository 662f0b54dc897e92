mkdir -p gpurun_out
export MIOPEN_ENABLE_LOGGING_CMD=0
( while true; do sleep 60; echo "[alive $(date +%T)] cache: $(du -sh ~/.cache/miopen 2>/dev/null | cut -f1) $(du -sh ~/.config/miopen 2>/dev/null | cut -f1)"; done ) &
HB=$!
timeout -k 10 1000 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench2.log 2> gpurun_out/bench2.err
echo exit=$? >> gpurun_out/bench2.log
kill $HB
tail -3 gpurun_out/bench2.log; grep bench gpurun_out/bench2.err | tail -8
du -sh ~/.cache/miopen ~/.config/miopen 2>/dev/null
