#!/bin/bash
# rocprofv3 per-kernel totals of a secondary training step: tools/prof_step.sh retina|v5 -> gpurun_out/r04_<which>_kernel_stats.csv
# (whole run: warm-up included; MIOpen immediate mode on the committed find-db, so there is no solver search in it)
ROOT=$GRAFT_REPO_ROOT
W=${1:-retina}
if [ "$W" = retina ]; then PROG="$ROOT/tools/bench_retina_step.py --steps 10 --warmup 3"; elif [ "$W" = ssd ]; then PROG="$ROOT/tools/bench_ssd_step.py --steps 10 --warmup 3"; else PROG="$ROOT/tools/bench_v5.py --steps 10 --warmup 3"; fi
rm -rf /tmp/prof_step
( cd /tmp && TMPDIR=/tmp timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_step -- python3 $PROG > $ROOT/gpurun_out/prof_step_$W.log 2>&1 )
f=$(find /tmp/prof_step -name "*kernel_stats.csv" 2>/dev/null | sort | sed -n 1p)
if [ -z "$f" ]; then echo "no kernel_stats.csv (see gpurun_out/prof_step_$W.log)"; tail -5 $ROOT/gpurun_out/prof_step_$W.log; exit 1; fi
cp "$f" $ROOT/gpurun_out/r04_${W}_kernel_stats.csv
tail -1 $ROOT/gpurun_out/prof_step_$W.log | cut -c1-300
sed -n 1,45p $ROOT/gpurun_out/r04_${W}_kernel_stats.csv | cut -c1-160
