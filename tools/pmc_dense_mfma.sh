#!/bin/bash
# PMC passes on the dense matrix-core kernels (fp32 32^3 headline kernel, the one-wave-per-item kernels on 40^3 .. 56^3, the work-group kernels):
# one --pmc group per run, --kernel-trace only (no other trace domains next to --pmc). Summary: gpurun_out/dense_mfma_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
export DENSE_SHAPES=32x32x32,40x40x40,48x48x48,56x56x56,64x64x64
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d gpurun_out/pmc_dm1 -o dm -- python3 tools/bench_dense.py all 3 > gpurun_out/pmc_dm1.log 2>&1 &&
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_dm2 -o dm -- python3 tools/bench_dense.py all 3 > gpurun_out/pmc_dm2.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_dm3 -o dm -- python3 tools/bench_dense.py all 3 > gpurun_out/pmc_dm3.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_dm4 -o dm -- python3 tools/bench_dense.py all 3 > gpurun_out/pmc_dm4.log 2>&1
python3 - > gpurun_out/dense_mfma_pmc.txt <<PY
import csv, glob, collections
print("rocprofv3 --pmc passes on the dense matrix-core kernels (tools/pmc_dense_mfma.sh; bench_dense.py, strided batches of ~6 GB traffic, beta=1).")
print("Counter values are means per launch, summed over the device (SQ_VALU_MFMA_BUSY_CYCLES over 1024 SIMDs, SQ_BUSY_CU_CYCLES over 256 CUs).")
print("mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES): share of the time a CU is busy during which each of its four matrix pipes works.")
print("FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE shows half of the bytes of a streaming read (see r1_pmc_summary.json), WRITE_SIZE is exact.")
def tag(name, row=None):
    # the hiprtc kernels all carry one name; the one-wave-per-item kernels are the 64-thread ones (dynamic LDS: the trace shows 0):
    # told apart by their register count and grid (8 waves per CU: fp32 40^3 / 48^3; 4 waves per CU: fp32 56^3, fp64 40^3 / 48^3)
    if row is not None and "xsmm_smm_op" in name and 64 == int(row["Workgroup_Size"]): return "smm_mfma_wave_jit (vgpr %s, %d waves/CU)" % (row["VGPR_Count"], int(row["Grid_Size"]) // 64 // 256)
    if row is not None and "xsmm_smm_op" in name and 256 == int(row["Workgroup_Size"]) and 32768 <= int(row["LDS_Block_Size"]) <= 33792 and int(row["VGPR_Count"]) >= 68: return "smm_f64_mfma_wg_jit (56^3, 64^3; vgpr %s)" % row["VGPR_Count"]
    if "smm32_f32_mfma" in name: return "smm_f32_32x32x32_mfma"
    if "smm64_f32_mfma" in name: return "smm_f32_64x64x64_mfma"
    if "smm_f32_mfma_wg" in name: return "smm_f32_mfma_wg (48^3)"
    if "smm_f64_mfma_wg" in name: return "smm_f64_mfma_wg (64^3, 48^3)"
    return None
table = collections.defaultdict(dict)
for d in ("pmc_dm1", "pmc_dm2", "pmc_dm3", "pmc_dm4"):
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            t = tag(r["Kernel_Name"], r)
            if t: agg[(t, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k in sorted(agg):
            table[k[0]][k[1]] = sum(agg[k]) / len(agg[k])
            print("%s  %-30s %-26s launches=%d mean=%.4g" % (d, k[0], k[1], len(agg[k]), table[k[0]][k[1]]))
for t in sorted(table):
    v = table[t]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("SQ_BUSY_CU_CYCLES"):
        line = "%-30s mfma_util = %.1f %%" % (t, 100.0 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * v["SQ_BUSY_CU_CYCLES"]))
        if v.get("SQ_LDS_IDX_ACTIVE"): line += "   LDS bank-conflict cycles / LDS active cycles = %.3f" % (v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"])
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v: line += "   HBM bytes per launch = %.4g (2 x FETCH_SIZE + WRITE_SIZE, KiB -> B)" % ((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0)
        print(line)
PY
cat gpurun_out/dense_mfma_pmc.txt
