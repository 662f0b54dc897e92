#!/bin/bash
# rocprofv3 per-kernel averages of the eval-side NMS launches (tools/prof_nms.py) -> gpurun_out/nms_kernel_stats.csv
ROOT=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_nms
( cd /tmp && TMPDIR=/tmp ONE_WG=${ONE_WG:-} timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_nms -- python3 $ROOT/tools/prof_nms.py > $ROOT/gpurun_out/prof_nms.log 2>&1 )
f=$(find /tmp/prof_nms -name "*kernel_stats.csv" 2>/dev/null | sort | sed -n 1p)
if [ -z "$f" ]; then echo "no kernel_stats.csv (see gpurun_out/prof_nms.log)"; tail -5 $ROOT/gpurun_out/prof_nms.log; exit 1; fi
cp "$f" $ROOT/gpurun_out/nms_kernel_stats.csv
sed -n 1,10p $ROOT/gpurun_out/nms_kernel_stats.csv | cut -c1-150
