#!/usr/bin/env python3
"""Sum FETCH_SIZE / WRITE_SIZE of the conv family's dispatches of a bench.py run (--mode net --steps 1 --warmup 1; the number
of forwards in the run = the number of seg-final kernel dispatches) and write traffic.json.
usage: traffic_summary.py <dir> [fp16|fp32s|fp32]"""
import collections
import csv
import glob
import json
import sys

out = sys.argv[1]
# kernel family by engine: fp16 (default) | fp32s | fp32
FAMILY = {"fp16": ("conv_igemm_kernel", "conv_halo_kernel", "conv_halo2_kernel", "conv_halo3_kernel", "c3_fused_kernel", "c3b_kernel"), "fp32s": ("conv_split_kernel", "conv_split_halo_kernel", "stem_split_kernel", "conv_f32_mfma_kernel"),
          "fp32": ("conv_f32_mfma_kernel",)}[sys.argv[2] if len(sys.argv) > 2 else "fp16"]
tot = collections.defaultdict(lambda: [0.0, 0])
stems = collections.defaultdict(int)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            if any(k in r["Kernel_Name"] for k in FAMILY):
                tot[c][0] += float(r["Counter_Value"])
                tot[c][1] += 1
            if "seg_final" in r["Kernel_Name"]:          # one per forward in every engine
                stems[c] += 1
n_fwd = max(stems["FETCH_SIZE"], 1)
res = {
    "counter_unit": "KB",
    "forwards_in_run": n_fwd,
    "igemm_dispatches": tot["FETCH_SIZE"][1],
    "FETCH_SIZE_KB_per_forward": tot["FETCH_SIZE"][0] / n_fwd,
    "WRITE_SIZE_KB_per_forward": tot["WRITE_SIZE"][0] / n_fwd,
}
# MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads
res["hbm_bytes_per_forward_corrected"] = (2 * res["FETCH_SIZE_KB_per_forward"] + res["WRITE_SIZE_KB_per_forward"]) * 1024
res["hbm_bytes_per_forward_uncorrected"] = (res["FETCH_SIZE_KB_per_forward"] + res["WRITE_SIZE_KB_per_forward"]) * 1024
json.dump(res, open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps(res))
