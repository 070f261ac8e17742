cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 -L > gpurun_out/rocprof_counters.txt 2>&1 || true
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_sp1 -o sp -- python3 tools/bench_sparse.py spmdm 3 > gpurun_out/pmc_sp1.log 2>&1 &&
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/pmc_sp2 -o sp -- python3 tools/bench_sparse.py spmdm 3 > gpurun_out/pmc_sp2.log 2>&1
