#!/bin/bash
# One-off per image / architecture: the convolution problems of the reference's other YOLO families (no BASELINE config beyond config 1's
# shape) into the committed MIOpen user db: YOLOv3 416^2 bs 16 (80 classes), YOLOv2 416^2 bs 16 and 288^2 bs 2 (20 classes: config 1).
# Run on the GPU box from the repo root; brings the result home under gpurun_out/miopen_db_new.
set -o pipefail
mkdir -p gpurun_out
WORK=$PWD/gpurun_out/miopen_db_new
rm -rf "$WORK"; cp -r miopen_db "$WORK"; mkdir -p "$WORK/cache"
( while true; do sleep 60; echo "[alive $(date +%T)]"; done ) & HB=$!
trap "kill $HB" EXIT
CASES=("--model YOLOv3 --img 416 --batch 16 --classes 80" "--model YOLOv2 --img 416 --batch 16 --classes 20" "--model YOLOv2 --img 288 --batch 2 --classes 20")
: > gpurun_out/db_yolo23.jsonl
for c in "${CASES[@]}"; do
  echo "== $c: committed db"; timeout -k 10 600 python tools/bench_model_step.py $c --steps 8 --warmup 3 2>> gpurun_out/db_yolo23.err | tee -a gpurun_out/db_yolo23.jsonl | cut -c1-200 || exit 1
done
export MIOPEN_USER_DB_PATH=$WORK MIOPEN_CUSTOM_CACHE_DIR=$WORK/cache
for c in "${CASES[@]}"; do
  echo "== $c: solver search"; timeout -k 10 1000 python tools/bench_model_step.py $c --miopen-find --steps 8 --warmup 3 2>> gpurun_out/db_yolo23.err | tee -a gpurun_out/db_yolo23.jsonl | cut -c1-200 || exit 1
done
for c in "${CASES[@]}"; do
  echo "== $c: immediate mode on the new db"; timeout -k 10 300 python tools/bench_model_step.py $c --steps 8 --warmup 3 2>> gpurun_out/db_yolo23.err | tee -a gpurun_out/db_yolo23.jsonl | cut -c1-200 || exit 1
done
du -sh "$WORK"; wc -l "$WORK"/*.txt
