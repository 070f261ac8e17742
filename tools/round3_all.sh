#!/bin/bash
# Round-3 evidence in one call on the MI355X box (about fifteen minutes): GPU tests, the bench line + rocprofv3 kernel statistics of
# the same command, PMC passes (headline traffic, grouped CP2K kernel, fsspmdm operator kernel), CP2K stacks, call benches,
# small batches, the callers either side of the hot path. Outputs under gpurun_out/ (copy the summaries to profiles/).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
bash tools/round_profile.sh > gpurun_out/round_profile.log 2>&1 || { tail -n 30 gpurun_out/round_profile.log; exit 1; }
echo "[round3] tests + bench + rocprof stats done"
timeout -k 10 400 bash tools/pmc_headline.sh && echo "[round3] headline pmc done"
(echo "== one grouped call, batch order, matrix cores on (default)"; timeout -k 10 200 python tools/bench_cp2k.py 524288 7;
 echo "== CP2K_MFMA=0: register-tiled run form"; CP2K_MFMA=0 timeout -k 10 200 python tools/bench_cp2k.py 524288 7 0 0 1;
 echo "== order relaxed (libxsmm_gemm_batch_omp semantics)"; timeout -k 10 200 python tools/bench_cp2k.py 524288 7 1 0 1;
 echo "== XSMM_SMMJIT_GROUPED_WPE=3"; XSMM_SMMJIT_GROUPED_WPE=3 timeout -k 10 200 python tools/bench_cp2k.py 524288 7 0 0 1;
 echo "== XSMM_SMMJIT_GROUPED_INLINE=0"; XSMM_SMMJIT_GROUPED_INLINE=0 timeout -k 10 200 python tools/bench_cp2k.py 524288 7 0 0 1;
 echo "== XSMM_SMMJIT_HANDWAIT=0: the compiler's wait counts in the run form"; XSMM_SMMJIT_HANDWAIT=0 timeout -k 10 200 python tools/bench_cp2k.py 524288 7;
 echo "== XSMM_SMMJIT_TILESPLIT=0: one call per shape with a wave per run"; XSMM_SMMJIT_TILESPLIT=0 timeout -k 10 200 python tools/bench_cp2k.py 524288 7;
 echo "== full config 5 on one GPU (4 194 304 products)"; timeout -k 10 200 python tools/bench_cp2k.py 4194304 5 0 0 1; timeout -k 10 200 python tools/bench_cp2k.py 4194304 5 1 0 1) 2>&1 | grep -v amdgpu.ids > gpurun_out/cp2k_stacks.txt
echo "[round3] cp2k stacks done"
timeout -k 10 300 bash tools/pmc_cp2k.sh > /dev/null 2>&1; echo "[round3] cp2k pmc done"
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cp2k -o cp -- python3 $GRAFT_REPO_ROOT/tools/bench_cp2k.py 524288 5 0 0 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_cp2k.log 2>&1)
timeout -k 10 600 bash tools/pmc_fsspmdm.sh > /dev/null 2>&1; echo "[round3] fsspmdm pmc done"
gcc -O2 -I include tools/bench_calls.c -o /tmp/bench_calls -L libxsmm-1_amd/lib -lxsmm -Wl,-rpath,$PWD/libxsmm-1_amd/lib &&
 (echo "--- default: a launch per call (what an unchanged caller gets) ---"; timeout -k 10 120 /tmp/bench_calls; timeout -k 10 120 /tmp/bench_calls 32 32 32 20000;
  echo "--- LIBXSMM_AMD_DEFER=1 (opt-in bursts; in code: libxsmm_amd_defer_begin/end) ---"; LIBXSMM_AMD_DEFER=1 timeout -k 10 120 /tmp/bench_calls; LIBXSMM_AMD_DEFER=1 timeout -k 10 120 /tmp/bench_calls 32 32 32 20000) > gpurun_out/bench_calls.txt 2>&1
gcc -O2 -I include tools/bench_panels.c -o /tmp/bench_panels -L libxsmm-1_amd/lib -lxsmm -Wl,-rpath,$PWD/libxsmm-1_amd/lib &&
 (echo "--- default: a launch per panel ---"; timeout -k 10 120 /tmp/bench_panels; echo "--- LIBXSMM_AMD_DEFER=1 (opt-in bursts) ---"; LIBXSMM_AMD_DEFER=1 timeout -k 10 120 /tmp/bench_panels) > gpurun_out/bench_panels.txt 2>&1
gcc -O2 -I include -I/opt/rocm/include tools/bench_call_latency.c -o /tmp/bench_call_latency -L libxsmm-1_amd/lib -lxsmm -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/libxsmm-1_amd/lib &&
 (echo "--- default ---"; timeout -k 10 120 /tmp/bench_call_latency; echo "--- LIBXSMM_AMD_DEFER=1 ---"; LIBXSMM_AMD_DEFER=1 timeout -k 10 120 /tmp/bench_call_latency) >> gpurun_out/bench_calls.txt 2>&1
echo "[round3] call benches done"
timeout -k 10 300 python3 tools/bench_small_batches.py 2>&1 | grep -v amdgpu.ids > gpurun_out/small_batches.txt
timeout -k 10 300 python3 tools/bench_tile_split.py 2>&1 | grep -v amdgpu.ids > gpurun_out/tile_split.txt
timeout -k 10 500 python tools/bench_dense.py all 7 2>&1 | grep -v amdgpu.ids > gpurun_out/dense_shapes.txt
(timeout -k 10 200 python tools/bench_sparse.py spmdm 10; timeout -k 10 200 python tools/bench_sparse.py fsspmdm 10; SP_BATCH=262144 timeout -k 10 200 python tools/bench_sparse.py fsspmdm 10) 2>&1 | grep -v amdgpu.ids > gpurun_out/sparse_phases.txt
echo "[round3] dense + sparse done"
(for a in "2048 32 f32" "2048 32 f64" "2048 64 f32" "2048 64 f64" "4096 32 f32" "4096 32 f64" "4096 64 f64"; do timeout -k 10 100 python3 tools/bench_blocked.py $a 2>&1 | tail -n 1; done;
 timeout -k 10 200 python3 tools/bench_spmdm_api.py 2048 0.15 10;
 timeout -k 10 200 python3 tools/bench_soa.py;
 echo "== bench_generic (matrix cores off: register-tiled forms)"; timeout -k 10 200 python3 tools/bench_generic.py;
 echo "== bench_generic (matrix cores on, the default policy)"; XSMM_BENCH_MFMA=1 timeout -k 10 200 python3 tools/bench_generic.py;
 timeout -k 10 200 python3 tools/bench_dense.py lowp 5;
 timeout -k 10 100 python3 tools/bench_host.py; timeout -k 10 100 python3 tools/bench_misc.py | tail -n 1; timeout -k 10 100 python3 tools/bench_autobatch.py | tail -n 1) 2>&1 | grep -v amdgpu.ids > gpurun_out/other_paths.txt
echo "[round3] other paths done"
tail -n 3 gpurun_out/gpu_tests.log; cut -c1-600 gpurun_out/bench_n1.json
