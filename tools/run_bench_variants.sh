mkdir -p gpurun_out
for v in "yolo_fill_nt=0" "yolo_fill_nt=1"; do
  echo "== $v"; timeout -k 10 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --tune $v 2> gpurun_out/bv.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline'])"
done
echo "== cpu baseline"; ( while true; do sleep 60; echo "[alive]"; done ) & HB=$!
timeout -k 10 600 python bench.py --steps 10 --warmup 3 2> gpurun_out/bench3.err > gpurun_out/bench3.log; echo exit=$?; kill $HB
tail -1 gpurun_out/bench3.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('cpu_baseline'))"
tail -3 gpurun_out/bench3.err
