# Does the step gain from keeping MIOpen off the asm igemm backward kernels (which zero-fill dX first: SubTensorOpWithScalar1d, 1.8 ms/step)?
run() { label=$1; shift; env "$@" python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
  run "default          " A=1
  run "no-asm-bwd       " MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC=0
  run "no-asm-wrw       " MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_XDLOPS_NHWC=0
  run "no-asm-bwd+wrw   " MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC=0 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_XDLOPS_NHWC=0
done
