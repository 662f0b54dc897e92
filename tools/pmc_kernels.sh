# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes, kernel-trace only) of the secondary dense kernels driven
# by tools/bench_kernels.py: retina_main, v5_dense (fp32 and bf16 launches appear as separate rows by order), nms prep, decode.
mkdir -p gpurun_out/prof
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C /tmp/pmck_$C
  ( while true; do sleep 60; echo "[alive $(date +%T)]"; done ) & HB=$!
  N=6 timeout -k 10 600 rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "retina_main|v5_expand|v5_obj|yolo_nms_prep|yolo_decode|greedy_nms_prep|yolo_expand|yolo_loss" --output-format csv -d /tmp/pmck_$C -- python3 $ROOT/tools/bench_kernels.py > $ROOT/gpurun_out/prof/pmck_${C}.log 2> $ROOT/gpurun_out/prof/pmck_${C}.err
  echo "$C exit=$?"; kill $HB
  F=$(find /tmp/pmck_$C -name '*counter_collection.csv' | head -1)
  python3 - "$F" "$C" > $ROOT/gpurun_out/prof/pmck_${C}.csv <<'PY'
import csv, sys
from collections import defaultdict
agg = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != sys.argv[2]:
        continue
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    agg[(n, r.get("Grid_Size", r.get("Grid_Size_X", "")))].append(float(r["Counter_Value"]))
print("kernel,grid,launches,%s_KiB_median,min,max" % sys.argv[2])
for (k, g), v in agg.items():
    v.sort()
    print(f"{k},{g},{len(v)},{v[len(v) // 2]:.1f},{v[0]:.1f},{v[-1]:.1f}")
PY
  cat $ROOT/gpurun_out/prof/pmck_${C}.csv
done
