mkdir -p gpurun_out
( while true; do sleep 60; echo "[alive]"; done ) & HB=$!
for v in "" "--fused-adam" "--miopen-find" "--miopen-find --fused-adam --channels-last"; do
  echo "== bench $v"; timeout -k 10 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline $v 2> gpurun_out/bv.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_us'], d['roofline']['frac'])"
  grep "warmup step 1/" gpurun_out/bv.err
done
kill $HB
