# PMC passes on the spmdm compute kernel (BASELINE config 4 shape): one --pmc group per run, --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d gpurun_out/pmc_sp1 -o sp -- python3 tools/bench_sparse.py spmdm 3 > gpurun_out/pmc_sp1.log 2>&1 &&
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_sp2 -o sp -- python3 tools/bench_sparse.py spmdm 3 > gpurun_out/pmc_sp2.log 2>&1
python3 - > gpurun_out/spmdm_pmc.txt <<PY
import csv, glob, collections
print("rocprofv3 --pmc passes on the spmdm compute kernels (BASELINE config 4 shape, 131072 items, beta=0 and beta=1 launches mixed); tools/pmc_spmdm.sh")
print("units: SQ_*_CYCLES/ACTIVE_* per wave or SIMD in quad-cycles as rocprof reports them, summed over the device; SQ_BUSY_CU_CYCLES and SQ_LDS_* summed over 256 CUs;")
print("SQ_VALU_MFMA_BUSY_CYCLES summed over 1024 SIMDs. Launches that returned at once (the kernel not chosen for the batch's density) are listed separately.")
for d in ("pmc_sp1", "pmc_sp2"):
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "spmdm_compute" in r["Kernel_Name"]:
                kind = "mfma" if "mfma" in r["Kernel_Name"] else "wg_lds"
                agg[(kind, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k in sorted(agg): print("%s  %-6s %-26s launches=%d mean=%.4g" % (d, k[0], k[1], len(agg[k]), sum(agg[k]) / len(agg[k])))
PY
cat gpurun_out/spmdm_pmc.txt
