#!/bin/bash
# rocprofv3 kernel statistics of the matrix-core work-group kernels (64^3 and 48^3 strided batches, fp32 and fp64).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export DENSE_SHAPES=64x64x64,48x48x48
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_d64 -o d64 -- python3 $R/tools/bench_dense.py all 7 > $R/gpurun_out/prof_d64.log 2>&1
grep "mfma=1" $R/gpurun_out/prof_d64.log
ls $R/gpurun_out/prof_d64
