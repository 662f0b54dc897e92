# HBM traffic of the yolo kernels from PMC counters (separate passes, kernel-trace only), per MI355X_MICROARCH.md
mkdir -p gpurun_out/prof
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C /tmp/pmck_$C
  ( while true; do sleep 60; echo "[alive $(date +%T)]"; done ) & HB=$!
  timeout -k 10 600 rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "yolo_|headgrad_" --output-format csv -d /tmp/pmc_$C -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary $EXTRA > $ROOT/gpurun_out/prof/pmc_${C}_bench.log 2> $ROOT/gpurun_out/prof/pmc_${C}_bench.err
  echo "$C exit=$?"; kill $HB
  F=$(find /tmp/pmc_$C -name '*counter_collection.csv' | head -1)
  python3 - "$F" "$C" > $ROOT/gpurun_out/prof/${TAG:-r02}_pmc_${C}.csv <<'PY'
import csv, sys
from collections import defaultdict
agg = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != sys.argv[2]:
        continue
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    agg[n].append(float(r["Counter_Value"]))
print("kernel,launches,%s_avg,%s_min,%s_max" % ((sys.argv[2],) * 3))
for k, v in agg.items():
    print(f"{k},{len(v)},{sum(v) / len(v):.1f},{min(v):.1f},{max(v):.1f}")
PY
  cat $ROOT/gpurun_out/prof/${TAG:-r02}_pmc_${C}.csv
done
