"""GPU box: dumps what tests/test_gpu_accept.py's band test compares (fp16 engine vs oracle on the three acceptance
pages) for offline analysis: lines / blocks of both sides, the flipped bitmap pixels, the float NMS detections of both
sides.  -> gpurun_out/band_dump.pkl"""
import importlib
import os
import pickle
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("comic-text-detector_amd")
from oracle import accept, cv_ref as cv, postproc_ref as R      # noqa: E402
from oracle.net_ref import OracleNet                            # noqa: E402

out = []
ck = pkg.synth.make_blob_checkpoint(0)
torch.set_num_threads(16)
for size, shape in ((512, (512, 512)), (1024, (1024, 1024)), (512, (700, 495))):
    page = pkg.synth.text_like_page(shape, 3, n_blocks=8)
    lb, ratio, (dw, dh) = cv.letterbox(page, (size, size))
    x = torch.from_numpy(np.ascontiguousarray(lb.transpose(2, 0, 1)[None])).float() / 255
    ob, om, ol = OracleNet(ck)(x)
    ref = R.detector_tail(page, ob.numpy(), om.numpy(), ol.numpy(), input_size=(size, size), dw=dw, dh=dh, refine_mode=0,
                          keep_undetected_mask=False)
    det = pkg.detector.TextDetector(ck, input_size=size, device="cuda", precision="fp16")
    got = det(page, refine_mode=0, keep_undetected_mask=False)
    blks, mask, lines = det.net.forward_u8(det._prepare([page])[0])
    torch.cuda.synchronize()
    dets, counts = pkg.backend.nms(blks, 0.4, 0.35)
    rd = R.non_max_suppression(ob.numpy(), 0.4, 0.35)[0]
    rep = accept.band_report(ol[0, 0].numpy(), om[0, 0].numpy(), det.net.bitmap[0].cpu().numpy(), det.net.mask_u8[0].cpu().numpy(), 4e-3)
    flips = rep.pop("_flips")
    # tail on the ORACLE's maps through the product tail (-> isolates the network's contribution)
    ob_boxes, ob_scores = R.seg_rep((size, size), ol.numpy())
    x8 = det._prepare([page])[0]
    ext = det.tail_batch([page], blks, det.net.mask_u8, lines[:, 0].contiguous(), det.net.bitmap,
                         metas=[(page.shape[0], page.shape[1], dw, dh)], want_extras=True)[0][3]
    out.append(dict(size=size, shape=shape, dwdh=(dw, dh), ref_db=(np.asarray(ob_boxes[0]), np.asarray(ob_scores[0])),
                    got_db=(ext["db_boxes"], ext["db_scores"]),
                    got=[(list(map(int, b.xyxy)), [np.asarray(l).tolist() for l in b.lines], b.language, bool(b.vertical)) for b in got[2]],
                    ref=[(list(map(int, b.xyxy)), [np.asarray(l).tolist() for l in b.lines], b.language, bool(b.vertical)) for b in ref[2]],
                    flips=np.argwhere(flips).astype(np.int16), dets=dets[0, : int(counts[0])].cpu().numpy(), ref_dets=np.asarray(rd),
                    prob_ref=ol[0, 0].numpy().astype(np.float16), prob=lines[0, 0].cpu().numpy().astype(np.float16), rep=rep))
    print(size, shape, rep, len(got[2]), len(ref[2]))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "band_dump.pkl"), "wb") as f:
    pickle.dump(out, f)
