import csv, collections, glob, sys
f=glob.glob(sys.argv[1]+"/*/*kernel_trace.csv")[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "yolo" in n or "headgrad" in n or "v5_" in n or "retina" in n:
        k=n.split("(")[0].replace("void ","").replace("(anonymous namespace)::","")[:40]
        d[(k, r["Grid_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"])].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(d.items()):
    v.sort(); print(k, len(v), "median %.1f us"%(v[len(v)//2]/1e3), "min %.1f"%(v[0]/1e3))
