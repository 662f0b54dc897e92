#!/bin/bash
# rocprofv3 per-kernel averages of tools/pmc_instr.py's launches (the small kernels DESIGN quotes below 0.70 of HBM) -> stdout
ROOT=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_k
( cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -- python3 $ROOT/tools/pmc_instr.py > $ROOT/gpurun_out/prof_kernels.log 2>&1 )
f=$(find /tmp/prof_k -name "*kernel_stats.csv" 2>/dev/null | sort | sed -n 1p)
if [ -z "$f" ]; then echo "no kernel_stats.csv"; tail -5 $ROOT/gpurun_out/prof_kernels.log; exit 1; fi
grep -E "${1:-bias_act|bn_|spp|focus|v5_expand|nms_prep}" "$f" | cut -d, -f1-4,6,7 | cut -c1-200
