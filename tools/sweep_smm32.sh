#!/bin/bash
# Developer sweep for the fp32 32^3 kernels (run on the GPU box): grid occupancy and MFMA variants.
# Usage: tools/sweep_smm32.sh > gpurun_out/sweep.log
set -u
run() { # label, env assignments...
  local label="$1"; shift
  local line
  line=$(env "$@" timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-secondary --no-cpu ${BENCH_ARGS:-} 2>/dev/null | tail -1)
  echo "$label $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(r["kernel"], "ms", r["launch_ms_avg"], "min", r["launch_ms_min"], "GB/s", r["achieved"], "frac", r["frac"], "ceil", r.get("stream_ceiling_gbs"))' 2>/dev/null || echo "FAILED: $line")"
}
for bpc in 2 3 4 5 6 8; do run "mfma  v0 bpc=$bpc" XSMM_SMM32_BPC=$bpc XSMM_SMM32_VARIANT=0; done
for bpc in 2 3 4 5 6 8; do run "mfma  v1 bpc=$bpc" XSMM_SMM32_BPC=$bpc XSMM_SMM32_VARIANT=1; done
BENCH_ARGS="--mfma 0"
for bpc in 2 3 4 6; do run "fma      bpc=$bpc" XSMM_SMM32_BPC=$bpc; done
BENCH_ARGS="--mode index"
run "mfma v0 index-mode" XSMM_SMM32_BPC=4
for s in 4 8 16 32; do XSMM_STREAM_BPC=$s run "stream bpc=$s" XSMM_STREAM_BPC=$s; done
