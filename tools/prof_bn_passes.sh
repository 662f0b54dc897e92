# per-shape durations of the four BN launches under rocprofv3, for a list of knob settings: "TB:SB TB:SB ..."
cd /tmp && export TMPDIR=/tmp
for cfg in ${CFGS:-1024:4096}; do
  TBv=${cfg%%:*}; SBv=${cfg##*:}
  for s in ${SHAPES:-0 1 2 3 4 5}; do
    rm -rf /tmp/pbn
    SHAPE=$s TB=$TBv SB=$SBv RT=${RT:-2048} timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/pbn -- python3 $GRAFT_REPO_ROOT/tools/bench_bn_passes.py > /dev/null 2>&1
    T=$(find /tmp/pbn -name '*kernel_trace.csv' | head -1)
    python3 - "$T" "$cfg" "$s" <<'PY'
import csv,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"]
    if "bn_" not in n: continue
    n=n.replace("(anonymous namespace)::","").split("(")[0].replace("void ","")
    d[n].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
MB=[354,177,88.6,44.3,22.1,11.1][int(sys.argv[3])]
out=[]
for n,v in sorted(d.items()):
    v.sort(); m=v[len(v)//2]
    passes={"bn_reduce_kernel<false, false>":1,"bn_apply_kernel<false>":2,"bn_reduce_kernel<true, false>":2,"bn_bwd_dx_kernel<false>":3}.get(n)
    out.append("%s=%.1fus%s"%(n.replace("_kernel","").replace("<false, false>","_fwd").replace("<true, false>","_bwd").replace("<false>",""),m,(" (%.2fTB/s)"%(passes*MB/m) if passes else "")))
print("cfg",sys.argv[2],"shape",sys.argv[3],"MB",MB," ".join(out),flush=True)
PY
  done
done
