#!/bin/bash
# deferred stores of the wave-per-item kernels (XSMM_SMMJIT_DEFER) against the form that stores right behind the arithmetic,
# with generic (FLAT) and global accesses, on the config-1/5 shapes
cd "$GRAFT_REPO_ROOT" || exit 1
export LIBXSMM_AMD_CACHE=/tmp/sweep_cache
for shp in ${SHAPES:-23x23x23 13x23x32 13x13x13 5x5x5 32x32x32}; do
  for prec in f64 f32; do
    for knobs in "XSMM_SMMJIT_DEFER=0" "XSMM_SMMJIT_DEFER=1" "XSMM_SMMJIT_DEFER=1 XSMM_SMMJIT_FLAT=0" "XSMM_SMMJIT_DEFER=0 XSMM_SMMJIT_FLAT=0"; do
      echo -n "[$knobs] "; env $knobs DENSE_SHAPES=$shp timeout -k 5 90 python tools/bench_dense.py $prec 5 2>&1 | grep -v amdgpu | tail -n 1
    done
  done
done
