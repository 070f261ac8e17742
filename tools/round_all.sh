#!/bin/bash
# Everything the profiles/ of a round are made of, in one call on the MI355X box (about ten minutes).
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/round_profile.sh > gpurun_out/round_profile.log 2>&1 || { tail -n 20 gpurun_out/round_profile.log; exit 1; }
cp gpurun_out/bench_n1.json gpurun_out/bench_n1_first.json
timeout -k 10 400 bash tools/pmc_headline.sh
timeout -k 10 1200 bash tools/round_extras.sh > gpurun_out/round_extras.log 2>&1
bash tools/round_extras2.sh > gpurun_out/round_extras2.log 2>&1
tail -n 3 gpurun_out/gpu_tests.log; cut -c1-400 gpurun_out/bench_n1.json
