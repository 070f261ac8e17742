#!/bin/bash
# the shapes beyond 32 on the matrix cores: one wave per item (default) against the work-group form
cd "$GRAFT_REPO_ROOT" || exit 1
export LIBXSMM_AMD_CACHE=/tmp/sweep_cache
for shp in ${SHAPES:-40x40x40 48x48x48 56x56x56 64x64x64 64x64x32}; do
  for prec in f64 f32; do
    for knobs in "XSMM_SMMJIT_MFMA_WAVE=1" "XSMM_SMMJIT_MFMA_WAVE=0"; do
      echo -n "[$knobs] "; env $knobs DENSE_SHAPES=$shp timeout -k 5 90 python tools/bench_dense.py $prec 5 2>&1 | grep -v amdgpu | tail -n 1
    done
  done
done
