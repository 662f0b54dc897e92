#!/bin/bash
# One-off per image / architecture: add the convolution problems of BASELINE configs 4 (RetinaNet-R50-FPN 800x1333 bs 4, fp32) and
# 5 (YOLOv5-l 640^2 bs 32, bf16) to the committed MIOpen user db (find-db records + perf-db + compiled-kernel cache), so that the
# IMMEDIATE mode (torch.backends.cudnn.benchmark = False) picks the measured-best solver for them without any search at start-up,
# identically in every rank process.  Run on the GPU box from the repo root; brings the result home under gpurun_out/miopen_db_new.
set -o pipefail
mkdir -p gpurun_out
WORK=$PWD/gpurun_out/miopen_db_new
rm -rf "$WORK"; cp -r miopen_db "$WORK"; mkdir -p "$WORK/cache"
export MIOPEN_USER_DB_PATH=$WORK MIOPEN_CUSTOM_CACHE_DIR=$WORK/cache
( while true; do sleep 60; echo "[alive $(date +%T)]"; done ) & HB=$!
trap "kill $HB" EXIT
echo "== config 4, solver search"; timeout -k 10 900 python tools/bench_retina_step.py --miopen-find --steps 8 --warmup 3 2> gpurun_out/db4_find.err | tee gpurun_out/db4_find.json | cut -c1-300 || exit 1
echo "== config 5, solver search"; timeout -k 10 900 python tools/bench_v5.py --miopen-find --steps 8 --warmup 3 2> gpurun_out/db5_find.err | tee gpurun_out/db5_find.json | cut -c1-300 || exit 1
echo "== config 4, immediate mode on the new db"; timeout -k 10 300 python tools/bench_retina_step.py --steps 8 --warmup 3 2> gpurun_out/db4_imm.err | tee gpurun_out/db4_imm.json | cut -c1-300 || exit 1
echo "== config 5, immediate mode on the new db"; timeout -k 10 300 python tools/bench_v5.py --steps 8 --warmup 3 2> gpurun_out/db5_imm.err | tee gpurun_out/db5_imm.json | cut -c1-300 || exit 1
echo "== config 4, immediate mode WITHOUT a user db (what round 3 called immediate mode)"
MIOPEN_USER_DB_PATH=/tmp/empty_db_$$ MIOPEN_CUSTOM_CACHE_DIR=/tmp/empty_db_$$/cache timeout -k 10 600 python tools/bench_retina_step.py --steps 8 --warmup 3 2> gpurun_out/db4_imm_nodb.err | tee gpurun_out/db4_imm_nodb.json | cut -c1-300
du -sh "$WORK" "$WORK"/* ; wc -l "$WORK"/*.txt
