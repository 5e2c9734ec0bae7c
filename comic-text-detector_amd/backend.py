"""`HipTextDetBackend`: the third backend behind the reference's seam

        blks, mask, lines_map = self.net(img_in)            (reference inference.py:146)

next to `TextDetBase` (torch, reference basemodel.py:222-244) and
`TextDetBaseDNN` (OpenCV-DNN, reference basemodel.py:246-256).  Same call
contract: `img_in` f32 (B,3,H,W) in [0,1] on the device -> `blks` (B,N,5+nc),
`mask` (B,1,H,W), `lines_map` (B,2,H,W), all f32 on the device.

PyTorch is used for device memory and streams only; every FLOP runs in
libctd_hip.so.  There is no CPU path: constructing the backend without a GPU
or without the built library raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple, Union

import numpy as np
import torch

from . import _lib as L
from . import graph


def _load_ckpt(model: Union[str, dict]) -> dict:
    if isinstance(model, dict):
        return model
    # the reference's own loader does a plain torch.load of the dict (basemodel.py:212)
    return torch.load(model, map_location="cpu", weights_only=False)


class HipTextDetBackend:
    def __init__(self, model: Union[str, dict], device: Union[str, int, torch.device] = "cuda",
                 precision: str = "fp16", act: str = "leaky", bitmap_thresh: float = 0.3, step_eval: bool = False,
                 db_k: float = 50.0, outputs: str = "all"):
        if not torch.cuda.is_available():
            raise L.CtdError("HipTextDetBackend needs a ROCm GPU (MI355X); there is no CPU fallback")
        dev = torch.device(device)
        if dev.type != "cuda":
            raise L.CtdError(f"device must be a cuda/hip device, got {device!r}")
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        self.prec = {"fp16": L.PREC_F16, "fp32": L.PREC_F32, "fp32s": L.PREC_F32S}[precision]
        self.precision = precision
        self._lib = L.lib()
        # `DBHead.forward(step_eval=True)` (reference basemodel.py:121-122): `lines_map` becomes the (B,1,H,W)
        # differentiable-binarisation map step_function(shrink, thresh) (:159-160, k = 50)
        self.step_eval, self.db_k, self.bitmap_thresh = bool(step_eval), float(db_k), float(bitmap_thresh)
        # outputs="all": the seam's contract (blks, mask (B,1,H,W) f32, lines_map (B,2,H,W) f32).
        # outputs="detector": what `TextDetector.__call__` consumes (reference inference.py:146-161 uses the u8
        # mask and `lines_map[:, 0]` only): the DB threshold branch is not lowered, `lines_map` is (B,1,H,W) and
        # the f32 mask is not written (`mask` is returned as None) -- same values, 0.2 ms and 270 MB less per 32 pages
        if outputs not in ("all", "detector"):
            raise ValueError("outputs must be 'all' or 'detector'")
        if outputs == "detector" and step_eval:
            raise ValueError("step_eval needs the threshold branch: outputs='all'")
        self.outputs = outputs
        ckpt = _load_ckpt(model)
        self.program = graph.lower(ckpt, self.prec, act=act, bitmap_thresh=bitmap_thresh, db_thresh=outputs == "all")
        T, O, blob = graph.to_ctypes(self.program)
        self._blob = blob
        h = C.c_void_p()
        L.check(self._lib.ctd_engine_create(C.byref(h), T, len(T), O, len(O),
                                            blob.ctypes.data_as(C.POINTER(C.c_float)), blob.size, self.prec,
                                            self.device.index), "ctd_engine_create")
        self._h = h
        self.no = self.program.meta["no"]
        self._offset_budget = 2 ** 31 - 1            # bytes addressable inside one activation tensor (see _run)
        self.mask_u8: Optional[torch.Tensor] = None      # fused (uint8)(mask*255), reference inference.py:96-99
        self.bitmap: Optional[torch.Tensor] = None       # fused lines_map[:,0] > 0.3, reference db_utils.py:71-72

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                self._lib.ctd_engine_destroy(h)
            except Exception:
                pass
            self._h = None

    # -- shapes -----------------------------------------------------------------
    def blks_rows(self, H: int, W: int) -> int:
        rows, no = C.c_int32(), C.c_int32()
        L.check(self._lib.ctd_engine_blks_shape(self._h, H, W, C.byref(rows), C.byref(no)), "blks_shape")
        return rows.value

    def _outputs(self, B: int, H: int, W: int):
        dev = self.device
        blks = torch.empty((B, self.blks_rows(H, W), self.no), dtype=torch.float32, device=dev)
        mask = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev) if self.outputs == "all" else None
        lines = torch.empty((B, self.program.meta.get("line_planes", 2), H, W), dtype=torch.float32, device=dev)
        mask_u8 = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
        bitmap = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
        return blks, mask, lines, mask_u8, bitmap

    def _run(self, inp: torch.Tensor, fmt: int, B: int, H: int, W: int, profile: bool = False, outs=None):
        if H % 64 or W % 64:
            raise ValueError("H and W must be multiples of 64 (stride-32 backbone + AvgPool2d(2))")
        # The MFMA kernels use 32-bit byte offsets inside a tensor: the largest activation (64 fp16 channels
        # at half resolution = 32 B per input pixel, 40 with slack) must stay below 2 GiB, so larger batches
        # run as consecutive sub-batches (pages are independent) that write into slices of ONE set of outputs.
        max_b = max(1, self._offset_budget // (H * W * 40)) if self.prec == L.PREC_F16 else B
        part = outs is not None                       # a sub-batch of the loop below: the caller finishes
        if outs is None:
            outs = self._outputs(B, H, W)
        if B > max_b and not profile:
            for i in range(0, B, max_b):
                j = min(B, i + max_b)
                self._run(inp[i:j], fmt, j - i, H, W, outs=tuple(None if t is None else t[i:j] for t in outs))
            return self._finish(outs, B, H, W)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        args = [self._h, inp.data_ptr(), fmt, B, H, W] + [0 if t is None else t.data_ptr() for t in outs] + [stream]
        if profile:
            n = self._lib.ctd_engine_n_ops(self._h)
            ms = (C.c_float * n)()
            L.check(self._lib.ctd_engine_profile(*args, ms), "ctd_engine_profile")
            self.last_op_ms = np.array(ms[:], dtype=np.float64)
        else:
            L.check(self._lib.ctd_engine_forward(*args), "ctd_engine_forward")
        return None if part else self._finish(outs, B, H, W)

    def _finish(self, outs, B: int, H: int, W: int):
        self._last_bhw = (B, H, W)
        self.mask_u8, self.bitmap = outs[3], outs[4]
        if self.step_eval:
            stream = torch.cuda.current_stream(self.device).cuda_stream
            step = torch.empty((B, 1, H, W), dtype=torch.float32, device=self.device)
            L.check(self._lib.ctd_db_step(outs[2].data_ptr(), B, H, W, self.db_k, step.data_ptr(), outs[4].data_ptr(),
                                          self.bitmap_thresh, stream), "ctd_db_step")
            return outs[0], outs[1], step
        return outs[0], outs[1], outs[2]

    # -- the seam ---------------------------------------------------------------
    def __call__(self, img_in: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        if img_in.dim() != 4 or img_in.shape[1] != 3:
            raise ValueError("img_in must be (B,3,H,W)")
        x = img_in.to(device=self.device, dtype=torch.float32).contiguous()
        B, _, H, W = x.shape
        return self._run(x, L.IN_NCHW_F32, B, H, W)

    forward = __call__

    def forward_u8(self, pages: torch.Tensor):
        """pages: (B,H,W,3) uint8 letterboxed, channel order as the net consumes it
        (SURVEY App. C-1).  Saves the f32 NCHW round trip (12.6 MB -> 3 MB per page)."""
        if pages.dim() != 4 or pages.shape[3] != 3 or pages.dtype != torch.uint8:
            raise ValueError("pages must be (B,H,W,3) uint8")
        x = pages.to(self.device).contiguous()
        B, H, W, _ = x.shape
        return self._run(x, L.IN_NHWC_U8, B, H, W)

    # -- hipGraph replay ------------------------------------------------------------
    def capture(self, B: int, H: int, W: int, fmt: str = "u8"):
        """Captures one forward for a fixed (B,H,W) into a hipGraph (through torch's stream
        capture: the engine launches on the capturing stream and never synchronises).  Returns
        (static_input, replay) where replay() re-launches the ~100 kernels with one graph launch
        and returns (blks, mask, lines_map) views of static output buffers; `mask_u8` / `bitmap`
        are static too.  Worth it for small batches, where launch overhead rivals kernel time."""
        shape, dtype, code = ((B, H, W, 3), torch.uint8, L.IN_NHWC_U8) if fmt == "u8" else \
                             ((B, 3, H, W), torch.float32, L.IN_NCHW_F32)
        static_in = torch.zeros(shape, dtype=dtype, device=self.device)
        self._run(static_in, code, B, H, W)              # plans the arena outside the capture
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            outs = self._run(static_in, code, B, H, W)
        side = (self.mask_u8, self.bitmap)
        gen = self.arena_generation()

        def replay():
            # the graph holds the arena's addresses of capture time: a later, larger shape reallocates the arena
            if self.arena_generation() != gen:
                raise L.CtdError("this captured forward is stale: the engine's arena was reallocated for a larger "
                                 "(B,H,W) after the capture; capture the largest shape first, or capture again")
            g.replay()
            self.mask_u8, self.bitmap = side
            self._last_bhw = (B, H, W)
            return outs

        replay.static_in = static_in        # the graph reads this buffer: it must live as long as the graph
        return static_in, replay

    # -- measurement helpers ------------------------------------------------------
    def profile(self, img_in: torch.Tensor):
        """One forward with a hipEvent pair around every op; returns dict(ms, flops, bytes, cls, names)."""
        x = img_in.contiguous()
        if x.dtype == torch.uint8:
            B, H, W, _ = x.shape
            self._run(x, L.IN_NHWC_U8, B, H, W, profile=True)
        else:
            B, _, H, W = x.shape
            self._run(x.float(), L.IN_NCHW_F32, B, H, W, profile=True)
        n = self._lib.ctd_engine_n_ops(self._h)
        fl, by, cl = (C.c_double * n)(), (C.c_double * n)(), (C.c_int32 * n)()
        L.check(self._lib.ctd_engine_op_work(self._h, fl, by, cl), "op_work")
        return dict(ms=self.last_op_ms, flops=np.array(fl[:]), bytes=np.array(by[:]), cls=np.array(cl[:]),
                    names=[o["name"] for o in self.program.ops])

    def op_kernels(self) -> list:
        """[(op name, kernel it launches under the current plan and tuning)] in program order, for the shape of the last
        forward (`ctd_engine_op_kernel`); "(fused)" = another op's launch does this op's work.  (The two heads' inner
        layers share names -- `upconv4.conv.1` exists in the UNet and in the DB head -- hence a list, not a dict.)"""
        n = self._lib.ctd_engine_n_ops(self._h)
        out, buf = [], C.create_string_buffer(64)
        for i in range(n):
            L.check(self._lib.ctd_engine_op_kernel(self._h, i, buf, 64), "ctd_engine_op_kernel")
            out.append((self.program.ops[i]["name"] or f"op{i}", buf.value.decode()))
        return out

    def read_tensor(self, name_or_id) -> np.ndarray:
        """Debug: activation tensor of the last forward as f32 (B,H,W,C).  Only
        meaningful with CTD_NO_REUSE=1 (otherwise the arena slot may have been reused)."""
        tid = self.program.taps[name_or_id] if isinstance(name_or_id, str) else int(name_or_id)
        c, down, _ = self.program.tensors[tid]
        B, H, W = self._last_bhw
        shape = (B, H >> down, W >> down, c)
        out = np.empty(shape, np.float32)
        L.check(self._lib.ctd_engine_read_tensor(self._h, tid, out.ctypes.data_as(C.POINTER(C.c_float)), out.size),
                "read_tensor")
        return out

    def arena_generation(self) -> int:
        return int(self._lib.ctd_engine_arena_generation(self._h))

    def workspace_bytes(self) -> int:
        return int(self._lib.ctd_engine_workspace_bytes(self._h))


class HipTextDetDNN:
    """The reference's OTHER backend object behind the same seam: `TextDetBaseDNN` (reference
    basemodel.py:246-256), the OpenCV-DNN runner of the exported ONNX file whose tensors are named
    `images` -> `blk`, `seg`, `det` (utils/export.py:43-44).  Same constructor and call contract:
    `TextDetBaseDNN(input_size, model_path)(im_in)` takes the letterboxed uint8 HWC image that
    `preprocess_img(..., to_tensor=False)` hands it (inference.py:72-83), scales it by 1/255 like
    `cv2.dnn.blobFromImage` (no channel swap, no mean) and returns numpy `(blks, mask, lines_map)`.
    `model_path` is the reference's checkpoint dict / .pt file (there is no ONNX parser here: the graph
    is the one `graph.lower` builds from the checkpoint, which is what the ONNX file was exported from)."""

    output_names = ("blk", "seg", "det")          # utils/export.py:44
    input_name = "images"                         # utils/export.py:43

    def __init__(self, input_size: int, model_path: Union[str, dict], device="cuda", precision: str = "fp32",
                 act: str = "leaky"):
        self.input_size = input_size
        self.net = HipTextDetBackend(model_path, device=device, precision=precision, act=act)
        self.uoln = list(self.output_names)       # getUnconnectedOutLayersNames()

    def forward_named(self, images: np.ndarray) -> dict:
        """{'images': (B,3,H,W) float32 in [0,1]} -> {'blk': .., 'seg': .., 'det': ..} as numpy arrays."""
        x = torch.from_numpy(np.ascontiguousarray(images, np.float32)).to(self.net.device)
        outs = self.net(x)
        torch.cuda.current_stream(self.net.device).synchronize()
        return {k: v.cpu().numpy() for k, v in zip(self.output_names, outs)}

    def __call__(self, im_in: np.ndarray):
        if im_in.dtype != np.uint8 or im_in.ndim != 3 or im_in.shape[2] != 3:
            raise ValueError("im_in must be a uint8 (H,W,3) image")
        if im_in.shape[0] != self.input_size or im_in.shape[1] != self.input_size:
            raise ValueError("blobFromImage would rescale: pass the letterboxed input_size x input_size image")
        blob = (im_in.astype(np.float32) * np.float32(1 / 255.0)).transpose(2, 0, 1)[None]     # scalefactor = 1/255
        out = self.forward_named(blob)
        return out["blk"], out["seg"], out["det"]


# ---------------------------------------------------------------------------
# device -> host copies through cached pinned buffers
# ---------------------------------------------------------------------------

_pinned: dict = {}


def to_host(t: torch.Tensor, key: str) -> np.ndarray:
    """Synchronous D2H copy of `t` into a cached page-locked buffer (pageable `.cpu()` copies of the
    label / probability maps ran at a few GB/s and dominated the host tail).  The returned array
    aliases the buffer of `key`: it is valid until the next `to_host` call with the same key."""
    t = t.contiguous()
    n = t.numel() * t.element_size()
    buf = _pinned.get(key)
    if buf is None or buf.numel() < n:
        buf = torch.empty(max(n, 1 << 20), dtype=torch.uint8, pin_memory=True)
        _pinned[key] = buf
    view = buf[:n].view(t.dtype).view(t.shape)
    view.copy_(t)
    return view.numpy()


# ---------------------------------------------------------------------------
# post-processing kernels
# ---------------------------------------------------------------------------

def nms(blks: torch.Tensor, conf_thres: float = 0.4, iou_thres: float = 0.35, max_det: int = 300,
        max_nms: int = 30000, max_wh: float = 4096.0):
    """HIP replacement of `non_max_suppression` (reference utils/yolov5_utils.py:124-218).
    blks (B,rows,no) f32 cuda -> (dets (B,max_det,6) [xyxy,conf,cls], counts (B,) i32)."""
    lib = L.lib()
    if not blks.is_cuda:
        raise L.CtdError("nms: blks must live on the GPU")
    blks = blks.contiguous().float()
    B, rows, no = blks.shape
    dets = torch.empty((B, max_det, 6), dtype=torch.float32, device=blks.device)
    counts = torch.empty((B,), dtype=torch.int32, device=blks.device)
    nbytes = lib.ctd_nms_workspace_bytes(B, rows)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=blks.device)
    stream = torch.cuda.current_stream(blks.device).cuda_stream
    L.check(lib.ctd_nms(blks.data_ptr(), B, rows, no, conf_thres, iou_thres, max_det, max_nms, max_wh,
                        dets.data_ptr(), counts.data_ptr(), ws.data_ptr(), nbytes, stream), "ctd_nms")
    return dets, counts


def connected_components(img: torch.Tensor, thresh: int = 0, connectivity: int = 8, max_labels: int = 4096):
    """HIP replacement of `cv2.connectedComponentsWithStats` (reference utils/textmask.py:93,113,138).
    img (B,H,W) or (H,W) u8 cuda; foreground = img > thresh.
    Returns labels (B,H,W) i32 (0 = background, 1..n raster order of first pixel),
    n (B,) i32, stats (B,max_labels,5) i32 [x,y,w,h,area] for labels 1..n."""
    lib = L.lib()
    if not img.is_cuda or img.dtype != torch.uint8:
        raise L.CtdError("connected_components: img must be a uint8 GPU tensor")
    if img.dim() == 2:
        img = img[None]
    img = img.contiguous()
    B, H, W = img.shape
    labels = torch.empty((B, H, W), dtype=torch.int32, device=img.device)
    n = torch.empty((B,), dtype=torch.int32, device=img.device)
    stats = torch.empty((B, max_labels, 5), dtype=torch.int32, device=img.device)
    nbytes = lib.ctd_ccl_workspace_bytes(B, H, W)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=img.device)
    stream = torch.cuda.current_stream(img.device).cuda_stream
    L.check(lib.ctd_ccl(img.data_ptr(), B, H, W, thresh, connectivity, labels.data_ptr(), n.data_ptr(),
                        stats.data_ptr(), max_labels, ws.data_ptr(), nbytes, stream), "ctd_ccl")
    return labels, n, stats


def connected_components_dual(img: torch.Tensor, thresh: int = 0, max_labels: int = 4096):
    """Foreground (8-connected) and background (4-connected) components of (img > thresh) in one pass -- the two
    labellings behind `cv2.findContours(bitmap, RETR_LIST)` (reference utils/db_utils.py:136).
    Returns labels (B,H,W) i32 signed (+id foreground / -id background), (n_f, n_b), (stats_f, stats_b),
    (first_f, first_b)."""
    lib = L.lib()
    if not img.is_cuda or img.dtype != torch.uint8:
        raise L.CtdError("connected_components_dual: img must be a uint8 GPU tensor")
    if img.dim() == 2:
        img = img[None]
    img = img.contiguous()
    B, H, W = img.shape
    dev = img.device
    labels = torch.empty((B, H, W), dtype=torch.int32, device=dev)
    n = [torch.empty((B,), dtype=torch.int32, device=dev) for _ in range(2)]
    stats = [torch.empty((B, max_labels, 5), dtype=torch.int32, device=dev) for _ in range(2)]
    first = [torch.empty((B, max_labels), dtype=torch.int32, device=dev) for _ in range(2)]
    nbytes = lib.ctd_ccl_workspace_bytes(B, H, W)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    L.check(lib.ctd_ccl_dual(img.data_ptr(), B, H, W, thresh, labels.data_ptr(), n[0].data_ptr(), n[1].data_ptr(),
                             stats[0].data_ptr(), stats[1].data_ptr(), first[0].data_ptr(), first[1].data_ptr(),
                             max_labels, ws.data_ptr(), nbytes, stream), "ctd_ccl_dual")
    return labels, tuple(n), tuple(stats), tuple(first)


def resize_linear_u8(src: torch.Tensor, dst_hw, canvas_hw=None) -> torch.Tensor:
    """HIP replacement of cv2.resize(INTER_LINEAR) for uint8 (H,W) or (H,W,3) GPU tensors
    (reference utils/imgproc_utils.py:113, inference.py:165); with `canvas_hw` larger than
    `dst_hw` the result sits in the top-left corner of a zero canvas = the reference's letterbox."""
    lib = L.lib()
    if not src.is_cuda or src.dtype != torch.uint8:
        raise L.CtdError("resize_linear_u8: src must be a uint8 GPU tensor")
    src = src.contiguous()
    C = 1 if src.dim() == 2 else src.shape[2]
    dH, dW = dst_hw
    cH, cW = canvas_hw if canvas_hw is not None else dst_hw
    shape = (cH, cW) if src.dim() == 2 else (cH, cW, C)
    dst = torch.empty(shape, dtype=torch.uint8, device=src.device)
    stream = torch.cuda.current_stream(src.device).cuda_stream
    L.check(lib.ctd_resize_linear_u8(src.data_ptr(), src.shape[0], src.shape[1], C, dst.data_ptr(), dH, dW, cH, cW,
                                     stream), "ctd_resize_linear_u8")
    return dst
