# PMC passes on the grouped CP2K run kernel (BASELINE config 5, the per-GPU shard: 524 288 products of 27 shapes in ONE
# libxsmm_amd_gemm_batch_groups call, sums per C block in batch order). One --pmc group per run, --kernel-trace only; the
# program itself after "--". Output: gpurun_out/cp2k_pmc.txt (copied to profiles/ by hand).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
P=${CP2K_PRODUCTS:-524288}
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_cp1 -o cp -- python3 tools/bench_cp2k.py $P 3 0 0 1 > gpurun_out/pmc_cp1.log 2>&1 &&
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d gpurun_out/pmc_cp2 -o cp -- python3 tools/bench_cp2k.py $P 3 0 0 1 > gpurun_out/pmc_cp2.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_cp3 -o cp -- python3 tools/bench_cp2k.py $P 3 0 0 1 > gpurun_out/pmc_cp3.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/pmc_cp4 -o cp -- python3 tools/bench_cp2k.py $P 3 0 0 1 > gpurun_out/pmc_cp4.log 2>&1
python3 - > gpurun_out/cp2k_pmc.txt <<PY
import csv, glob, collections
print("rocprofv3 --pmc passes on the grouped CP2K run kernel (config 5, $P products of 27 shapes, one libxsmm_amd_gemm_batch_groups call, batch order); tools/pmc_cp2k.sh")
print("SQ_* summed over the device as rocprofv3 reports them (cycle counters in quad-cycles per wave / SIMD); FETCH_SIZE in KB, to be doubled for wide streaming reads (gfx950)")
for d in ("pmc_cp1", "pmc_cp2", "pmc_cp3", "pmc_cp4"):
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "grouped" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in sorted(agg): print("%s  %-28s launches=%d mean=%.5g" % (d, k, len(agg[k]), sum(agg[k]) / len(agg[k])))
    for f in glob.glob("gpurun_out/%s/**/*kernel_trace.csv" % d, recursive=True):
        ds = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "grouped" in r["Kernel_Name"]]
        if ds: print("%s  kernel duration under the counters: launches=%d mean=%.1f us min=%.1f us" % (d, len(ds), sum(ds) / len(ds) / 1e3, min(ds) / 1e3))
PY
cat gpurun_out/cp2k_pmc.txt
