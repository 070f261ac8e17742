#!/bin/bash
# knob sweep of the wave-per-item kernels on the config-1/5 shapes (work-groups per CU, flat vs global accesses, items per wave)
cd "$GRAFT_REPO_ROOT" || exit 1
export LIBXSMM_AMD_CACHE=/tmp/sweep_cache
for shp in 23x23x23 13x23x32 13x13x13; do
  for prec in f64 f32; do
    for knobs in "" "XSMM_SMMJIT_BPC=2" "XSMM_SMMJIT_BPC=3" "XSMM_SMMJIT_BPC=4" "XSMM_SMMJIT_BPC=6" "XSMM_SMMJIT_FLAT=0" "XSMM_SMMJIT_PACK=2" "XSMM_SMMJIT_PACK=4"; do
      echo -n "[$knobs] "; env $knobs DENSE_SHAPES=$shp timeout -k 5 60 python tools/bench_dense.py $prec 5 2>&1 | grep -v amdgpu | tail -n 1
    done
  done
done
