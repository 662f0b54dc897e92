#!/bin/bash
# SQ counters of the kernels DESIGN calls instruction-bound (one rocprofv3 --pmc pass, kernel-trace only): wave instructions issued,
# VALU instructions, waves, busy cycles -> gpurun_out/r04_pmc_instr.csv (per kernel and grid: medians over the launches)
ROOT=$GRAFT_REPO_ROOT
rm -rf /tmp/pmc_instr
( cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace \
    --kernel-include-regex "bn_reduce|bn_apply|bn_bwd_dx|spp_|s2d|focus|yolo_nms_prep|bias_act_bwd|v5_expand" --output-format csv -d /tmp/pmc_instr -- python3 $ROOT/tools/pmc_instr.py > $ROOT/gpurun_out/pmc_instr.log 2>&1 )
F=$(find /tmp/pmc_instr -name '*counter_collection.csv' 2>/dev/null | sort | sed -n 1p)
if [ -z "$F" ]; then echo "no counter_collection.csv"; tail -5 $ROOT/gpurun_out/pmc_instr.log; exit 1; fi
python3 - "$F" > $ROOT/gpurun_out/r04_pmc_instr.csv <<'PY'
import csv, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    key = (n, r.get("Grid_Size", r.get("Grid_Size_X", "")))
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY"]
print("kernel,grid,launches," + ",".join(names) + ",VALU_insts_per_wave")
for (k, g), d in sorted(agg.items()):
    med = {}
    for c in names:
        v = sorted(d.get(c, [0.0]))
        med[c] = v[len(v) // 2]
    per = med["SQ_INSTS_VALU"] / med["SQ_WAVES"] if med["SQ_WAVES"] else 0
    print(f"{k},{g},{len(d.get('SQ_WAVES', []))}," + ",".join(f"{med[c]:.0f}" for c in names) + f",{per:.1f}")
PY
cat $ROOT/gpurun_out/r04_pmc_instr.csv | cut -c1-220
