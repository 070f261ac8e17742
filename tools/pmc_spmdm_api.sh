# rocprofv3 on the reference-API spmdm path (one 2048^3 problem, 15 % non-zeros): kernel statistics, then two --pmc groups
# (one group per run, --kernel-trace only). usage (GPU box): bash tools/pmc_spmdm_api.sh [n=2048] [density=0.15]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
N=${1:-2048}; D=${2:-0.15}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/spapi_stats -o sp -- python3 tools/bench_spmdm_api.py $N $D 10 > gpurun_out/spapi_stats.log 2>&1 &&
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/spapi_pmc1 -o sp -- python3 tools/bench_spmdm_api.py $N $D 3 > gpurun_out/spapi_pmc1.log 2>&1 &&
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/spapi_pmc2 -o sp -- python3 tools/bench_spmdm_api.py $N $D 3 > gpurun_out/spapi_pmc2.log 2>&1
python3 - > gpurun_out/spmdm_api_pmc.txt <<PY
import csv, glob, collections
print("rocprofv3 on tools/bench_spmdm_api.py $N $D (reference-API spmdm, one problem); tools/pmc_spmdm_api.sh")
for f in glob.glob("gpurun_out/spapi_stats/**/*kernel_stats.csv", recursive=True):
    print("kernel statistics (--kernel-trace --stats):")
    for r in csv.DictReader(open(f)):
        if "spmdm" in r["Name"]: print("  %-60s calls=%s avg=%.1f us min=%.1f us max=%.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
print("counters (summed over the device, mean per launch; SQ_*CYCLES / ACTIVE_* / WAIT_* in quad-cycles as rocprofv3 reports them):")
for d in ("spapi_pmc1", "spapi_pmc2"):
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "spmdm" in r["Kernel_Name"]:
                kind = "tiled_all" if ("tiled" in r["Kernel_Name"] and int(r["Grid_Size"]) > 64 * 512) else ("tiled_block" if "tiled" in r["Kernel_Name"] else ("create_all" if int(r["Grid_Size"]) > 1024 else "create_block"))
                agg[(kind, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k in sorted(agg): print("  %-12s %-26s launches=%d mean=%.4g" % (k[0], k[1], len(agg[k]), sum(agg[k]) / len(agg[k])))
PY
cat gpurun_out/spmdm_api_pmc.txt
