#!/usr/bin/env python3
"""Summary of tools/pmc_headline.sh: HBM bytes per launch of the headline kernel (and of the pure streaming kernel that
calibrates the counter) from the two rocprofv3 --pmc passes.  usage: python3 tools/pmc_summarise.py ROUND > profiles/rROUND_pmc_summary.json
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a coalesced streaming read
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section; re-checked here on stream_abc_kernel), WRITE_SIZE is exact."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csv.field_size_limit(1 << 30)
ALGO = 16384 * 1048576  # bytes per launch of the headline workload (and of the stream probe: 3 x 4 GiB read + 4 GiB written)


def collect(path, counter):
    out = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            key = "smm_f32_32x32x32_mfma" if "smm32_f32_mfma_kernel" in name else ("stream_abc_kernel" if "stream_abc_kernel" in name else None)
            if key is None:
                continue
            out.setdefault(key, []).append(float(row["Counter_Value"]))
    return out


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "2"
    fetch = collect(os.path.join(ROOT, "gpurun_out", "pmc_fetch", "f_counter_collection.csv"), "FETCH_SIZE")
    write = collect(os.path.join(ROOT, "gpurun_out", "pmc_write", "w_counter_collection.csv"), "WRITE_SIZE")
    res = {"note": "rocprofv3 --pmc passes (separate runs, --kernel-trace only) on MI355X, round %s. FETCH_SIZE/WRITE_SIZE are in KiB. On gfx950 "
                   "FETCH_SIZE reports half of the bytes of a coalesced streaming read (MI355X_MICROARCH.md, HBM): read bytes = 2 x FETCH_SIZE x 1024, "
                   "checked on stream_abc_kernel (3 x 4 GiB read). WRITE_SIZE is exact." % rnd,
           "commands": ["bash tools/pmc_headline.sh", "python3 tools/pmc_summarise.py %s" % rnd]}
    for key in sorted(set(fetch) | set(write)):
        fv, wv = fetch.get(key, []), write.get(key, [])
        big_f = [v for v in fv if v > 0.5 * max(fv)] if fv else []
        big_w = [v for v in wv if v > 0.5 * max(wv)] if wv else []
        e = {}
        if big_f:
            e["FETCH_SIZE_KiB_mean"] = sum(big_f) / len(big_f); e["launches_fetch"] = len(big_f)
            e["read_bytes_per_launch"] = 2.0 * 1024.0 * e["FETCH_SIZE_KiB_mean"]
        if big_w:
            e["WRITE_SIZE_KiB_mean"] = sum(big_w) / len(big_w); e["launches_write"] = len(big_w)
            e["write_bytes_per_launch"] = 1024.0 * e["WRITE_SIZE_KiB_mean"]
        if big_f and big_w:
            e["traffic_bytes_per_launch"] = e["read_bytes_per_launch"] + e["write_bytes_per_launch"]
        e["algorithmic_bytes_per_launch"] = float(ALGO)
        res[key] = e
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
