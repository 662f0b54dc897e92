# device durations (rocprofv3 kernel trace, cold caches) of the BN launches per shape: one-launch path vs three-launch path
cd /tmp && export TMPDIR=/tmp
for s in ${SHAPES:-16,512,13,13 16,1024,13,13 16,256,26,26 16,512,26,26 16,128,52,52 16,256,52,52}; do
  for f in 1 0; do
    rm -rf /tmp/pbn
    SHAPE=$s FUSED=$f timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/pbn -- python3 $GRAFT_REPO_ROOT/tools/bench_bn_passes.py > /dev/null 2>&1
    T=$(find /tmp/pbn -name '*kernel_trace.csv' | head -1)
    python3 - "$T" "$s" "$f" <<'PY'
import csv,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"]
    if "bn_" not in n: continue
    n=n.replace("(anonymous namespace)::","").split("(")[0].replace("void ","")
    d[n].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
out=[]; tot=0
for n,v in sorted(d.items()):
    v.sort(); m=v[len(v)//2]; tot+=m
    out.append("%s=%.1f"%(n.replace("_kernel",""),m))
print("shape",sys.argv[2],"fused",sys.argv[3],"sum_us=%.1f"%tot," ".join(out),flush=True)
PY
  done
done
