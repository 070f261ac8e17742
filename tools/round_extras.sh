#!/bin/bash
# Secondary measurements for profiles/: dense shape sweep, sparse phases, CP2K stacks (1 and many streams), spmdm PMC.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 500 python tools/bench_dense.py all 7 2>&1 | grep -v amdgpu.ids > gpurun_out/dense_shapes.txt &&
(timeout -k 10 200 python tools/bench_sparse.py spmdm 10; timeout -k 10 200 python tools/bench_sparse.py fsspmdm 10) 2>&1 | grep -v amdgpu.ids > gpurun_out/sparse_phases.txt &&
(echo "== libxsmm_gemm_batch (sums in batch order), GPU_MAX_HW_QUEUES=16"; GPU_MAX_HW_QUEUES=16 timeout -k 10 200 python tools/bench_cp2k.py 524288 5 0;
 echo "== libxsmm_gemm_batch_omp (order relaxed, as in the reference's multi-threaded path), GPU_MAX_HW_QUEUES=16"; GPU_MAX_HW_QUEUES=16 timeout -k 10 200 python tools/bench_cp2k.py 524288 5 1;
 echo "== full config 5 on one GPU (4 194 304 products), default queues"; timeout -k 10 200 python tools/bench_cp2k.py 4194304 5 0; timeout -k 10 200 python tools/bench_cp2k.py 4194304 5 1) 2>&1 | grep -v amdgpu.ids > gpurun_out/cp2k_stacks.txt &&
timeout -k 10 300 bash tools/pmc_spmdm.sh
# the callers and data formats either side of the hot path: blocked GEMM, one large GEMM, the per-block spmdm interface,
# SOA kernels, the generic kernel next to the specialised ones, low-precision kernels
(for a in "2048 32 f32" "2048 32 f64" "2048 64 f32" "2048 64 f64" "4096 32 f32" "4096 64 f32" "4096 64 f64"; do timeout -k 10 100 python3 tools/bench_blocked.py $a 2>&1 | tail -n 1; done;
 timeout -k 10 200 python3 tools/bench_gemm_single.py 256 1024 2048 4096; LIBXSMM_AMD_BLAS=0 timeout -k 10 200 python3 tools/bench_gemm_single.py 2048;
 timeout -k 10 200 python3 tools/bench_spmdm_api.py 2048 0.15 2;
 timeout -k 10 200 python3 tools/bench_soa.py;
 timeout -k 10 200 python3 tools/bench_generic.py;
 timeout -k 10 200 python3 tools/bench_dense.py lowp 5;
 timeout -k 10 100 python3 tools/bench_host.py; timeout -k 10 100 python3 tools/bench_misc.py | tail -n 1; timeout -k 10 100 python3 tools/bench_autobatch.py | tail -n 1) 2>&1 | grep -v amdgpu.ids > gpurun_out/other_paths.txt
tail -n 4 gpurun_out/dense_shapes.txt gpurun_out/sparse_phases.txt gpurun_out/cp2k_stacks.txt
