#!/bin/bash
# Round-end evidence run on the MI355X box: full GPU test suite, the bench line, and the rocprofv3 kernel statistics of
# the same bench command. Outputs under gpurun_out/ (copy the summaries to profiles/).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/gpu_tests.log 2>&1 || { tail -n 30 gpurun_out/gpu_tests.log; exit 1; }
tail -n 3 gpurun_out/gpu_tests.log
timeout -k 10 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err || { tail gpurun_out/bench_n1.err; exit 1; }
cat gpurun_out/bench_n1.json
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o smm32 -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 2 --no-cpu --no-variants > "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.err") || { tail gpurun_out/prof_bench.err; exit 1; }
ls gpurun_out/prof
