# PMC passes on the fsspmdm operator kernel (BASELINE config 3: fp64 M=K=35, N=96, ~15 % nnz) at 65 536 and at the
# configuration's own 262 144 items: address translation (UTCL1 / UTCL2), L2 hit rate and fetched bytes -- why does the larger
# batch lose 9-13 points? One --pmc group per run, --kernel-trace only. Output: gpurun_out/fsspmdm_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for B in 65536 262144; do
  export SP_BATCH=$B
  rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum --kernel-trace --output-format csv -d gpurun_out/pmc_fs1_$B -o fs -- python3 tools/bench_sparse.py fsspmdm 3 > gpurun_out/pmc_fs1_$B.log 2>&1
  rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_fs2_$B -o fs -- python3 tools/bench_sparse.py fsspmdm 3 > gpurun_out/pmc_fs2_$B.log 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/pmc_fs3_$B -o fs -- python3 tools/bench_sparse.py fsspmdm 3 > gpurun_out/pmc_fs3_$B.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/pmc_fs4_$B -o fs -- python3 tools/bench_sparse.py fsspmdm 3 > gpurun_out/pmc_fs4_$B.log 2>&1
done
python3 - > gpurun_out/fsspmdm_pmc.txt <<PY
import csv, glob, collections
print("rocprofv3 --pmc passes on fsspmdm_f64_jit_operator (config 3 shape, beta = 1 and beta = 0 launches mixed, fp64 and fp32 kernels listed apart); tools/pmc_fsspmdm.sh")
for B in (65536, 262144):
    for d in ("pmc_fs1", "pmc_fs2", "pmc_fs3", "pmc_fs4"):
        for f in glob.glob("gpurun_out/%s_%d/**/*counter_collection.csv" % (d, B), recursive=True):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if "fsspmdm" in r["Kernel_Name"] or "xsmm_fsspmdm" in r["Kernel_Name"] or "operator" in r["Kernel_Name"]:
                    agg[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
            for k in sorted(agg): print("items=%-7d %s  %-40s %-32s launches=%d mean=%.5g" % (B, d, k[0], k[1], len(agg[k]), sum(agg[k]) / len(agg[k])))
        for f in glob.glob("gpurun_out/%s_%d/**/*kernel_trace.csv" % (d, B), recursive=True):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if "fsspmdm" in r["Kernel_Name"] or "operator" in r["Kernel_Name"]: agg[r["Kernel_Name"][:40]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
            for k in sorted(agg): print("items=%-7d %s  %-40s duration under the counters: launches=%d mean=%.1f us min=%.1f us" % (B, d, k, len(agg[k]), sum(agg[k]) / len(agg[k]) / 1e3, min(agg[k]) / 1e3))
PY
cat gpurun_out/fsspmdm_pmc.txt
