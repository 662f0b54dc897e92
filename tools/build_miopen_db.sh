# one-off (per image/arch): run the bench with the exhaustive MIOpen search and bring the user db home
mkdir -p gpurun_out
( while true; do sleep 60; echo "[alive $(date +%T)]"; done ) & HB=$!
rm -rf miopen_db; mkdir -p miopen_db/cache
MIOPEN_USER_DB_PATH=$PWD/miopen_db MIOPEN_CUSTOM_CACHE_DIR=$PWD/miopen_db/cache timeout -k 10 1000 python bench.py --steps 20 --warmup 4 --no-cpu-baseline > gpurun_out/bench_find.log 2> gpurun_out/bench_find.err; echo exit=$?
kill $HB
tail -c 1500 gpurun_out/bench_find.log; grep "warmup step 1/" gpurun_out/bench_find.err
du -sh miopen_db miopen_db/* | head; ls -la miopen_db | head -20
rm -rf gpurun_out/miopen_db; cp -r miopen_db gpurun_out/miopen_db
echo "== second run (warm db)"
timeout -k 10 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline 2> gpurun_out/bench_find2.err | tail -c 600; grep "warmup step 1/" gpurun_out/bench_find2.err
