#!/bin/bash
# One-off per image / architecture: add SSD300-VGG16's convolution problems (bs 32, fp32, channels-last; no BASELINE config, but a model
# family of the reference) to the committed MIOpen user db, as tools/extend_miopen_db.sh does for configs 4 and 5.  Without records
# the immediate mode falls back to MIOpen's own ranking -- and its first step runs the naive reference kernels for half a minute.
# Run on the GPU box from the repo root; brings the result home under gpurun_out/miopen_db_new.
set -o pipefail
mkdir -p gpurun_out
WORK=$PWD/gpurun_out/miopen_db_new
rm -rf "$WORK"; cp -r miopen_db "$WORK"; mkdir -p "$WORK/cache"
( while true; do sleep 60; echo "[alive $(date +%T)]"; done ) & HB=$!
trap "kill $HB" EXIT
echo "== SSD300, immediate mode on the committed db (no SSD records)"; timeout -k 10 600 python tools/bench_ssd_step.py --steps 8 --warmup 3 2> gpurun_out/dbssd_before.err | tee gpurun_out/dbssd_before.json | cut -c1-300
export MIOPEN_USER_DB_PATH=$WORK MIOPEN_CUSTOM_CACHE_DIR=$WORK/cache
echo "== SSD300, solver search"; timeout -k 10 1000 python tools/bench_ssd_step.py --miopen-find --steps 8 --warmup 3 2> gpurun_out/dbssd_find.err | tee gpurun_out/dbssd_find.json | cut -c1-300 || exit 1
echo "== SSD300, immediate mode on the new db"; timeout -k 10 300 python tools/bench_ssd_step.py --steps 8 --warmup 3 2> gpurun_out/dbssd_imm.err | tee gpurun_out/dbssd_imm.json | cut -c1-300 || exit 1
du -sh "$WORK" "$WORK"/* ; wc -l "$WORK"/*.txt
