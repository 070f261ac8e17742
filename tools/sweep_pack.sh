#!/bin/bash
# items per wave pass (XSMM_SMMJIT_PACK) and work-groups per CU of the wave-per-item kernels on the config-1/5 shapes
cd "$GRAFT_REPO_ROOT" || exit 1
export LIBXSMM_AMD_CACHE=/tmp/sweep_cache
for shp in ${SHAPES:-23x23x23 13x23x32 13x13x13}; do
  for prec in f64 f32; do
    for knobs in "XSMM_SMMJIT_PACK=1" "XSMM_SMMJIT_PACK=2" "XSMM_SMMJIT_PACK=4" "XSMM_SMMJIT_PACK=1 XSMM_SMMJIT_BPC=3" "XSMM_SMMJIT_PACK=1 XSMM_SMMJIT_BPC=6" "XSMM_SMMJIT_PACK=2 XSMM_SMMJIT_BPC=3"; do
      echo -n "[$knobs] "; env $knobs DENSE_SHAPES=$shp timeout -k 5 90 python tools/bench_dense.py $prec 5 2>&1 | grep -v amdgpu | tail -n 1
    done
  done
done
