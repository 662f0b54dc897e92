cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bn -- python3 $GRAFT_REPO_ROOT/tools/bench_bnact.py > /dev/null 2>&1
S=$(find /tmp/prof_bn -name '*kernel_stats.csv' | head -1)
python3 - "$S" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    n=r["Name"].replace("(anonymous namespace)::","")[:70]
    print(f'{n:70s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:9.1f} max_us={float(r["MaxNs"])/1e3:9.1f} pct={r["Percentage"]}')
PY
