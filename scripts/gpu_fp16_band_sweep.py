"""The fp16 engine's "bounded and explained" claim (tests/test_gpu_accept.py `fp16_band`) over MANY pages (round 6): for page after
page -- its maps within EPS of the oracle's, every flipped pixel of the DB bitmap / the mask at level 127 inside the threshold band
of the ORACLE's map, every differing line / block attributed (oracle/accept.py explain_geometry), nothing unexplained.
BAND_N pages (default 30), BAND_SIZE (512), BAND_SEED0 (300)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import pkg          # noqa: E402
import test_gpu_accept as TA       # noqa: E402

p = pkg()
n = int(os.environ.get("BAND_N", "30"))
size = int(os.environ.get("BAND_SIZE", "512"))
seed0 = int(os.environ.get("BAND_SEED0", "300"))
ck = TA.blob_ckpt()
det = p.detector.TextDetector(ck, input_size=size, device="cuda", precision="fp16")
rng = np.random.RandomState(seed0)
bad = 0
tot = {"lines": 0, "lines_id": 0, "blocks": 0, "blocks_id": 0, "flips": 0, "max_prob": 0.0, "max_mask": 0.0, "band": 0.0}
for i in range(n):
    shape = (size, size) if i % 3 else (int(rng.randint(300, 900)), int(rng.randint(300, 900)))
    page = p.synth.text_like_page(shape, seed0 + i, n_blocks=int(rng.randint(4, 12)))
    TA._ORACLE.clear()                                   # the test module caches the oracle's result per (size, page shape)
    got = det(page, refine_mode=0, keep_undetected_mask=False)
    rep, geo = TA.fp16_band(p, ck, det, page, size, got)
    ref = TA.oracle_result(ck, page, size)
    cmp_ = TA.accept.compare(got, ref)
    ok = (rep["prob_max_abs_delta"] < TA.EPS_FP16 and rep["mask_max_abs_delta"] < TA.EPS_FP16 and rep["bitmap_flips_out_of_band"] == 0 and
          rep["mask127_flips_out_of_band"] == 0 and geo["lines_unexplained"] == 0 and geo["blocks_unexplained"] == 0)
    bad += not ok
    tot["lines"] += cmp_["lines"]["ref"]; tot["lines_id"] += cmp_["lines"]["identical"]
    tot["blocks"] += cmp_["blocks"]["ref"]; tot["blocks_id"] += cmp_["blocks"]["identical"]
    tot["flips"] += rep["bitmap_flips"]
    tot["max_prob"] = max(tot["max_prob"], rep["prob_max_abs_delta"]); tot["max_mask"] = max(tot["max_mask"], rep["mask_max_abs_delta"])
    tot["band"] = max(tot["band"], rep["bitmap_in_band_frac"])
    if not ok:
        print(f"page {seed0 + i} {shape}: {rep} {geo}", flush=True)
print(f"fp16 band sweep, input {size}: {n} pages, {bad} with something out of band or unexplained; lines identical {tot['lines_id']} / {tot['lines']}, "
      f"blocks {tot['blocks_id']} / {tot['blocks']}; bitmap pixels flipped {tot['flips']} (all inside the band); max |map - oracle| {tot['max_prob']:.2e} "
      f"(shrink map) {tot['max_mask']:.2e} (mask) against EPS {TA.EPS_FP16}; widest band {tot['band']:.4f} of a page's pixels")
sys.exit(1 if bad else 0)
