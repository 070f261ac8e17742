#!/bin/bash
# runs one command once per A/B library: tools/ab_run.sh "v0 v1" python tools/bench_sparse.py spmdm 10
names=$1; shift
for n in $names; do
  echo "=== $n"
  LIBXSMM_AMD_LIBRARY=$PWD/libxsmm-1_amd/lib/ab/libxsmm_$n.so timeout -k 10 200 "$@" 2>&1 | grep -v amdgpu.ids || exit 1
done
