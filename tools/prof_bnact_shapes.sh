# per (kernel, launch grid) durations of the fused BN+act kernels over the six YOLOv4 stage shapes
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_bn2 -- python3 $GRAFT_REPO_ROOT/tools/bench_bnact.py > /dev/null 2>&1
T=$(find /tmp/prof_bn2 -name '*kernel_trace.csv' | head -1)
python3 - "$T" <<'PY'
import csv,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"]
    if "bn_" not in n: continue
    n=n.replace("(anonymous namespace)::","").split("(")[0].replace("void ","")
    g=int(r.get("Grid_Size_X",r.get("Grid_Size",0)) or 0)
    d[(n,g)].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for (n,g),v in sorted(d.items()):
    v.sort()
    print(f"{n:40s} grid={g:9d} calls={len(v):3d} median_us={v[len(v)//2]:8.1f} min_us={v[0]:8.1f}")
PY
