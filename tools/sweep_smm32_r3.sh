run() { local label="$1"; shift; local line; line=$(env "$@" timeout -k 10 200 python bench.py --steps 15 --warmup 3 --no-secondary --no-cpu 2>/dev/null | tail -1); echo "$label $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("ms", r["launch_ms_avg"], "min", r["launch_ms_min"], "frac", r["frac"])' 2>/dev/null || echo FAILED)"; }
for rep in 1 2; do
for v in 0 1; do for bpc in 3 4 5; do run "variant=$v bpc=$bpc" XSMM_SMM32_BPC=$bpc XSMM_SMM32_VARIANT=$v; done; done
done
run "variant=1 bpc=4 nt=0" XSMM_SMM32_BPC=4 XSMM_SMM32_VARIANT=1 XSMM_SMM32_NT=0
run "variant=0 bpc=3 nt=0" XSMM_SMM32_BPC=3 XSMM_SMM32_VARIANT=0 XSMM_SMM32_NT=0
