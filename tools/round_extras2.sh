#!/bin/bash
# Round-2 additions to the evidence: PMC passes of the dense matrix-core kernels and of the reference-API spmdm path,
# per-call bursts, and the callers either side of the hot path (re-run).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 500 bash tools/pmc_dense_mfma.sh > /dev/null 2>&1
timeout -k 10 300 bash tools/pmc_spmdm_api.sh > /dev/null 2>&1
gcc -O2 -I include tools/bench_calls.c -o /tmp/bench_calls -L libxsmm-1_amd/lib -lxsmm -Wl,-rpath,$PWD/libxsmm-1_amd/lib &&
 (echo "--- default: a launch per call (what an unchanged caller gets) ---"; timeout -k 10 120 /tmp/bench_calls; echo "--- LIBXSMM_AMD_DEFER=1 (opt-in bursts; in code: libxsmm_amd_defer_begin/end) ---"; LIBXSMM_AMD_DEFER=1 timeout -k 10 120 /tmp/bench_calls) > gpurun_out/bench_calls.txt 2>&1
gcc -O2 -I include tools/bench_panels.c -o /tmp/bench_panels -L libxsmm-1_amd/lib -lxsmm -Wl,-rpath,$PWD/libxsmm-1_amd/lib &&
 (echo "--- default: a launch per panel ---"; timeout -k 10 120 /tmp/bench_panels; echo "--- LIBXSMM_AMD_DEFER=1 (opt-in bursts) ---"; LIBXSMM_AMD_DEFER=1 timeout -k 10 120 /tmp/bench_panels) > gpurun_out/bench_panels.txt 2>&1
gcc -O2 -I include -I/opt/rocm/include tools/bench_call_latency.c -o /tmp/bench_call_latency -L libxsmm-1_amd/lib -lxsmm -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/libxsmm-1_amd/lib &&
 (echo "--- default ---"; timeout -k 10 120 /tmp/bench_call_latency; echo "--- LIBXSMM_AMD_DEFER=1 ---"; LIBXSMM_AMD_DEFER=1 timeout -k 10 120 /tmp/bench_call_latency) >> gpurun_out/bench_calls.txt 2>&1
SHAPES="33x33x33 40x40x40 45x45x45 48x48x48 56x56x56 64x40x19 64x64x32 64x64x64" timeout -k 10 400 bash tools/sweep_mfma_shapes.sh > gpurun_out/sweep_mfma_shapes.txt 2>&1
(timeout -k 10 200 python3 tools/bench_gemm_single.py 256 1024 2048 4096; LIBXSMM_AMD_BLAS=0 timeout -k 10 200 python3 tools/bench_gemm_single.py 2048) 2>&1 | grep -v amdgpu.ids > gpurun_out/gemm_single.txt
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
tail -n 6 gpurun_out/dense_mfma_pmc.txt; cat gpurun_out/gemm_single.txt gpurun_out/bench_calls.txt
