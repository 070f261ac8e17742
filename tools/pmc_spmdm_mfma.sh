# MFMA pipe / LDS / issue counters of the spmdm compute kernel (one --pmc pass per group, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d gpurun_out/pmc_spm1 -o sp -- python3 tools/bench_sparse.py spmdm 3 > gpurun_out/pmc_spm1.log 2>&1 &&
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_spm2 -o sp -- python3 tools/bench_sparse.py spmdm 3 > gpurun_out/pmc_spm2.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("pmc_spm1", "pmc_spm2"):
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "spmdm_compute" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in sorted(agg): print(d, k, "launches=%d mean=%.4g" % (len(agg[k]), sum(agg[k]) / len(agg[k])))
PY
