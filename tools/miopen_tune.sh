# Exhaustive MIOpen tuning of the YOLOv4 step's convolutions (MIOPEN_FIND_ENFORCE=SEARCH_DB_UPDATE) into a copy of miopen_db/;
# the tuned user dbs come back under gpurun_out/miopen_tuned/ for inspection / commit.  One forward+backward triggers every problem.
ROOT=$GRAFT_REPO_ROOT
D=$ROOT/gpurun_out/miopen_tuned
rm -rf $D; mkdir -p $D/cache; cp -r $ROOT/miopen_db/* $D/
( while true; do sleep 60; echo "[alive $(date +%T)] $(wc -l $D/*.udb.txt 2>/dev/null | tail -1)"; done ) &
HB=$!
MIOPEN_USER_DB_PATH=$D MIOPEN_CUSTOM_CACHE_DIR=$D/cache MIOPEN_FIND_ENFORCE=${ENFORCE:-4} timeout -k 10 ${TMO:-1000} python $ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $ROOT/gpurun_out/miopen_tune.json 2> $ROOT/gpurun_out/miopen_tune.err
echo "bench exit=$?"
kill $HB
wc -l $D/*.txt; du -sh $D
MIOPEN_USER_DB_PATH=$D MIOPEN_CUSTOM_CACHE_DIR=$D/cache timeout -k 10 300 python $ROOT/bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('tuned db:', d['value'], d['ms_per_step'])"
