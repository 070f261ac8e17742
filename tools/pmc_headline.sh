#!/bin/bash
# HBM traffic of the headline kernel from the TCC counters: FETCH_SIZE and WRITE_SIZE in separate passes (they do not fit
# one pass), --kernel-trace only (no other trace domains next to --pmc). Outputs under gpurun_out/pmc_{fetch,write}.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-secondary > $R/gpurun_out/pmc_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o w -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-secondary > $R/gpurun_out/pmc_write.log 2>&1
