"""Python face of the CPU oracle (ctypes over oracle/libfav_oracle.so + numpy file formats).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product path never imports this module.

Parity status: the consistency mask (A3/A4) is pinned against the reference's own compiled
consistencyChecker (oracle/_ref, tests/golden/mask_*); the "stn" warp (A2, the reference's GPU path) is
pinned against the reference's own kernel stnbdhw/BilinearSamplerBDHW.cu:1-109 compiled for gfx950
(oracle/_ref/libwarp_ref*.so, tests/golden/warp_*.npz); everything that restates un-vendored Torch7
packages (nn, image: A5-A9, the "cpu" warp) is "parity unpinned" -- the reference ships no golden
vectors, weights or tests and Lua/Torch7 cannot run offline.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
REF_CHECKER = os.path.join(_HERE, "_ref", "consistencyChecker")


def build(force: bool = False) -> None:
    so = os.path.join(_HERE, "libfav_oracle.so")
    src = os.path.join(_HERE, "fav_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libfav_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/consistencyChecker") and (force or not os.path.exists(REF_CHECKER)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/stnbdhw") and (force or not warp_ref_available()):
        subprocess.check_call(["make", "-C", _HERE, "warp_ref"], stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        build()
        _LIB = C.CDLL(os.path.join(_HERE, "libfav_oracle.so"))
        _LIB.orc_avg.restype = C.c_float
    return _LIB


def set_threads(n: int) -> None:
    """Cap the OpenMP team (tests on many-core hosts: the problems are tiny, fork/join would dominate)."""
    lib().orc_set_threads(int(n))


def _f(a):
    a = np.ascontiguousarray(a, np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _u8(a):
    a = np.ascontiguousarray(a, np.uint8)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint8))


# ------------------------------------------------------------------------------------------ formats
def read_flo(path: str) -> np.ndarray:
    """Middlebury .flo -> [H][W][2] (u, v) float32.  Tag read but not validated
    (flowFileLoader.lua:17, consistencyChecker.cpp:24)."""
    with open(path, "rb") as f:
        f.read(4)
        w, h = np.frombuffer(f.read(8), "<i4")
        return np.frombuffer(f.read(int(w) * int(h) * 8), "<f4").reshape(int(h), int(w), 2).copy()


def write_flo(path: str, uv: np.ndarray) -> None:
    h, w, _ = uv.shape
    with open(path, "wb") as f:
        f.write(np.float32(202021.25).tobytes()); f.write(np.array([w, h], "<i4").tobytes())
        f.write(np.ascontiguousarray(uv, "<f4").tobytes())


def flo_to_lua(uv: np.ndarray) -> np.ndarray:
    """[H][W][(u,v)] -> the Lua loader's [2][H][W] with [0]=v(dy), [1]=u(dx) (flowFileLoader.lua:27-29)."""
    return np.ascontiguousarray(np.stack([uv[..., 1], uv[..., 0]]), np.float32)


def flo_to_planar_uv(uv: np.ndarray) -> np.ndarray:
    """[H][W][(u,v)] -> the checker's planar [2][H][W], plane 0 = u (consistencyChecker.cpp:31-32)."""
    return np.ascontiguousarray(np.stack([uv[..., 0], uv[..., 1]]), np.float32)


def write_pnm(path: str, a: np.ndarray) -> None:
    a = np.ascontiguousarray(a, np.uint8)
    with open(path, "wb") as f:
        if a.ndim == 2:
            f.write(b"P5\n%d %d\n255\n" % (a.shape[1], a.shape[0]))
        else:
            f.write(b"P6\n%d %d\n255\n" % (a.shape[1], a.shape[0]))
        f.write(a.tobytes())


def read_pnm(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        data = f.read()
    toks, p = [], 0
    while len(toks) < 4:
        while data[p:p + 1].isspace(): p += 1
        if data[p:p + 1] == b"#":
            while data[p:p + 1] != b"\n": p += 1
            continue
        q = p
        while not data[q:q + 1].isspace(): q += 1
        toks.append(data[p:q]); p = q
    p += 1
    w, h = int(toks[1]), int(toks[2])
    ch = 3 if toks[0] == b"P6" else 1
    a = np.frombuffer(data, np.uint8, w * h * ch, p)
    return a.reshape(h, w, 3).copy() if ch == 3 else a.reshape(h, w).copy()


# ------------------------------------------------------------------------------------------ ops
def warp(img: np.ndarray, flow_lua: np.ndarray, border: str = "stn") -> np.ndarray:
    """border: "stn" (the reference's GPU kernel, sum rounded once from double), "stn_f32" (the same in fp32 exactly as
    written: bit-identical to the reference kernel built without contraction), "cpu" (image.warp, recalled)."""
    img, pi = _f(img); flow_lua, pf = _f(flow_lua)
    c, h, w = img.shape; _, ho, wo = flow_lua.shape
    out = np.empty((c, ho, wo), np.float32)
    fn = {"stn": lib().orc_warp_stn, "stn_f32": lib().orc_warp_stn_f32, "cpu": lib().orc_warp_cpu}[border]
    fn(pi, pf, out.ctypes.data_as(C.POINTER(C.c_float)), c, h, w, ho, wo)
    return out


_WARP_REF = {}


def warp_ref_available() -> bool:
    return all(os.path.exists(os.path.join(_HERE, "_ref", n)) for n in ("libwarp_ref.so", "libwarp_ref_nofma.so"))


def warp_ref_gpu(img_t, flow_t, contract: bool = True):
    """The reference's OWN warp kernel (stnbdhw/BilinearSamplerBDHW.cu:48-109, compiled for gfx950 by oracle/Makefile into
    oracle/_ref/) on torch device tensors: img [B][C][H][W], flow [B][2][Ho][Wo] -> [B][C][Ho][Wo].  contract=True is the
    default-flags build (FMA contraction allowed, as nvcc's default), False the -ffp-contract=off build."""
    import torch
    name = "libwarp_ref.so" if contract else "libwarp_ref_nofma.so"
    if name not in _WARP_REF:
        _WARP_REF[name] = C.CDLL(os.path.join(_HERE, "_ref", name))
    b, c, h, w = img_t.shape; ho, wo = flow_t.shape[2], flow_t.shape[3]
    assert img_t.is_contiguous() and flow_t.is_contiguous() and img_t.dtype == torch.float32 and flow_t.dtype == torch.float32
    out = torch.full((b, c, ho, wo), float("nan"), dtype=torch.float32, device=img_t.device)
    rc = _WARP_REF[name].warp_ref_bdhw(C.c_void_p(img_t.data_ptr()), C.c_void_p(flow_t.data_ptr()), C.c_void_p(out.data_ptr()),
                                       b, c, h, w, ho, wo, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, f"reference warp kernel launch failed: hip error {rc}"
    torch.cuda.synchronize()
    return out


def corners(rgb_hwc_u8: np.ndarray) -> np.ndarray:
    h, w, _ = rgb_hwc_u8.shape
    planar, pp = _u8(np.transpose(rgb_hwc_u8, (2, 0, 1)))
    out = np.empty((h, w), np.float32)
    lib().orc_corners(pp, out.ctypes.data_as(C.POINTER(C.c_float)), w, h)
    return out


def consistency(flow1_uv: np.ndarray, flow2_uv: np.ndarray, rgb_hwc_u8: Optional[np.ndarray] = None) -> np.ndarray:
    """flow*_uv: [H][W][2] as read from .flo.  Returns the PGM payload [H][W] u8 in {0,255}."""
    h, w, _ = flow1_uv.shape
    f1, p1 = _f(flo_to_planar_uv(flow1_uv)); f2, p2 = _f(flo_to_planar_uv(flow2_uv))
    out = np.empty((h, w), np.uint8)
    if rgb_hwc_u8 is not None:
        st, ps = _f(corners(rgb_hwc_u8))
    else:
        st, ps = None, None
    lib().orc_consistency(p1, p2, ps, out.ctypes.data_as(C.POINTER(C.c_uint8)), w, h)
    return out


def min_filter(cert: np.ndarray, r: int = 7) -> np.ndarray:
    cert, pc = _f(cert); h, w = cert.shape
    out = np.empty_like(cert)
    lib().orc_min_filter(pc, out.ctypes.data_as(C.POINTER(C.c_float)), h, w, r)
    return out


def preprocess(rgb: np.ndarray) -> np.ndarray:
    rgb, p = _f(rgb); out = np.empty_like(rgb)
    lib().orc_preprocess(p, out.ctypes.data_as(C.POINTER(C.c_float)), rgb.shape[1], rgb.shape[2])
    return out


def deprocess(bgr: np.ndarray) -> np.ndarray:
    bgr, p = _f(bgr); out = np.empty_like(bgr)
    lib().orc_deprocess(p, out.ctypes.data_as(C.POINTER(C.c_float)), bgr.shape[1], bgr.shape[2])
    return out


def assemble(frame_rgb: np.ndarray, warped_rgb: Optional[np.ndarray], cert: Optional[np.ndarray]) -> np.ndarray:
    frame_rgb, pf = _f(frame_rgb); _, h, w = frame_rgb.shape
    out = np.empty((7, h, w), np.float32)
    if warped_rgb is None:
        lib().orc_assemble(pf, None, None, out.ctypes.data_as(C.POINTER(C.c_float)), h, w)
    else:
        warped_rgb, pw = _f(warped_rgb); cert, pc = _f(cert)
        lib().orc_assemble(pf, pw, pc, out.ctypes.data_as(C.POINTER(C.c_float)), h, w)
    return out


def to_u8_hwc(rgb: np.ndarray) -> np.ndarray:
    rgb, p = _f(rgb); _, h, w = rgb.shape
    out = np.empty((h, w, 3), np.uint8)
    lib().orc_to_u8_hwc(p, out.ctypes.data_as(C.POINTER(C.c_uint8)), h, w)
    return out


# ------------------------------------------------------------------------------------------ network
def conv2d(x, w, b, stride, pad):
    x, px = _f(x); w, pw = _f(w)
    cin, h, ww = x.shape; cout, _, kh, kw = w.shape
    oh, ow = (h + 2 * pad - kh) // stride + 1, (ww + 2 * pad - kw) // stride + 1
    out = np.empty((cout, oh, ow), np.float32)
    if b is not None:
        b, pb = _f(b)
    else:
        pb = None
    lib().orc_conv2d(px, cin, h, ww, pw, pb, cout, kh, kw, stride, stride, pad, pad,
                     out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def full_conv2d(x, w, b, stride, pad, adj):
    x, px = _f(x); w, pw = _f(w)
    cin, h, ww = x.shape; _, cout, k, _ = w.shape
    oh, ow = (h - 1) * stride - 2 * pad + k + adj, (ww - 1) * stride - 2 * pad + k + adj
    out = np.empty((cout, oh, ow), np.float32)
    pb = _f(b)[1] if b is not None else None
    bb = _f(b)[0] if b is not None else None
    lib().orc_full_conv2d(px, cin, h, ww, pw, bb.ctypes.data_as(C.POINTER(C.c_float)) if bb is not None else None, cout, k, stride, pad, adj,
                          out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def batchnorm_eval_(x, mean, var, gamma, beta, eps=1e-5, relu=False):
    assert x.dtype == np.float32 and x.flags.c_contiguous
    m, pm = _f(mean); v, pv = _f(var); g, pg = _f(gamma); b, pb = _f(beta)
    lib().orc_batchnorm_eval(x.ctypes.data_as(C.POINTER(C.c_float)), x.shape[0], x.shape[1], x.shape[2], pm, pv, pg, pb,
                             C.c_float(eps), 1 if relu else 0)
    return x


def instnorm_(x, gamma, beta, eps=1e-5, relu=False):
    assert x.dtype == np.float32 and x.flags.c_contiguous
    g, pg = _f(gamma); b, pb = _f(beta)
    lib().orc_instnorm(x.ctypes.data_as(C.POINTER(C.c_float)), x.shape[0], x.shape[1], x.shape[2], pg, pb,
                       C.c_float(eps), 1 if relu else 0)
    return x


def reflect_pad(x, l, r, t, b):
    x, px = _f(x); c, h, w = x.shape
    out = np.empty((c, h + t + b, w + l + r), np.float32)
    lib().orc_reflect_pad(px, c, h, w, l, r, t, b, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def upsample(x, s):
    x, px = _f(x); c, h, w = x.shape
    out = np.empty((c, h * s, w * s), np.float32)
    lib().orc_upsample_nearest(px, c, h, w, s, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def shave_add(block, skip, s):
    block, pb = _f(block); skip, ps = _f(skip); c, h, w = skip.shape
    out = np.empty((c, h - 2 * s, w - 2 * s), np.float32)
    lib().orc_shave_add(pb, ps, c, h, w, s, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def bf16_round(a: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 (round to nearest even) -> fp32: the operand rounding of the product's optional fast mode."""
    b = np.ascontiguousarray(a, np.float32).view(np.uint32)
    r = ((b + np.uint32(0x7FFF) + ((b >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)) << np.uint32(16)
    return r.view(np.float32)


def _bf16_conv(L: dict) -> bool:
    """the convolutions the fast mode touches: 3x3, stride 1, 32..256 input channels in multiples of 32, <= 128 outputs
    padded to 64 or 128 (= conv3_halo_eligible in the product)"""
    cout, cin, kh, kw = L["w"].shape
    cp = (cout + 31) // 32 * 32
    return kh == 3 and kw == 3 and L["stride"] == 1 and cin % 32 == 0 and 32 <= cin <= 256 and cp in (64, 128)


def net_forward(layers: List[dict], x: np.ndarray, trace: Optional[list] = None, bf16_ops: bool = False) -> np.ndarray:
    """model:forward(input) for the layer list of fav_amd.t7.extract_layers
    (fast_artistic_video_core.lua:172; models_video.lua:55-140).  x: [7][H][W] -> [3][H'][W']."""
    x = np.ascontiguousarray(x, np.float32)
    i = 0
    while i < len(layers):
        L = layers[i]; t = L["type"]
        if t == "pad":
            if L.get("mode", "reflect") == "replicate":     # nn.SpatialReplicationPadding (models_video.lua:14-16,29-31,73-75): the edge pixel repeated
                x = np.ascontiguousarray(np.pad(x, ((0, 0), (L["t"], L["b"]), (L["l"], L["r"])), mode="edge"))
            else:                                           # nn.SpatialReflectionPadding
                x = reflect_pad(x, L["l"], L["r"], L["t"], L["b"])
        elif t == "conv":
            if bf16_ops and _bf16_conv(L):      # FAV_PRECISION_BF16_OPERANDS: both operands rounded, wide accumulation
                x = conv2d(bf16_round(x), bf16_round(L["w"]), L["b"], L["stride"], L["pad"])
            else:
                x = conv2d(x, L["w"], L["b"], L["stride"], L["pad"])
        elif t == "fullconv":
            x = full_conv2d(x, L["w"], L["b"], L["stride"], L["pad"], L["adj"])
        elif t == "bn":
            relu = i + 1 < len(layers) and layers[i + 1]["type"] == "relu"
            x = batchnorm_eval_(x, L["mean"], L["var"], L["gamma"], L["beta"], L["eps"], relu)
            if relu: i += 1
        elif t == "in":
            relu = i + 1 < len(layers) and layers[i + 1]["type"] == "relu"
            x = instnorm_(x, L["gamma"], L["beta"], L["eps"], relu)
            if relu: i += 1
        elif t == "relu":
            x = np.maximum(x, 0)
        elif t == "res":
            y = net_forward(L["block"], x, None, bf16_ops)
            x = shave_add(y, x, L["shave"]) if L["shave"] else y + x
        elif t == "up":
            x = upsample(x, L["s"])
        elif t == "tanh":
            x = np.ascontiguousarray(x); lib().orc_tanh_mul(x.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(x.size), C.c_float(1.0))
        elif t == "mul":
            x = (x * np.float32(L["k"])).astype(np.float32)
        elif t == "identity":
            pass
        else:
            raise ValueError(t)
        if trace is not None: trace.append((t, x.copy()))
        i += 1
    return x


def temporal_loss(prev_rgb, cur_rgb, backward_flow_uv, cert01, border: str = "stn") -> float:
    """fast_artistic_video.lua:128-151 (third number of func_eval): nn.MSECriterion between the flow-warped previous and the
    current stylised frame, both multiplied by the certainty; accumulated in fp64 here."""
    wpd = warp(prev_rgb, flo_to_lua(backward_flow_uv), border)
    c = np.asarray(cert01, np.float32)[None]
    d = (wpd * c).astype(np.float32) - (np.asarray(cur_rgb, np.float32) * c).astype(np.float32)
    return float(np.mean(d.astype(np.float64) ** 2))


class Stylizer:
    """Recurrent per-frame loop: fast_artistic_video_core.lua:189-229 with the video CLI's
    callbacks (fast_artistic_video.lua:93-172); fill_occlusions = vgg-mean."""

    def __init__(self, layers, border="stn", min_filter_r=7, invert_occlusion=False, fix_occlusions=False,
                 fill_random=False, seed=1):
        self.layers, self.border, self.r = layers, border, min_filter_r
        self.invert, self.fix = invert_occlusion, fix_occlusions
        self.fill_random, self.seed, self.count = fill_random, seed, 0      # -fill_occlusions uniform-random (core.lua:108-117)
        self.last = None        # last_frame_stylized: float RGB [3][H][W], unclamped (fav.lua:169)

    def _fill(self, cert):
        """generate_fill with the documented counter RNG (vr_oracle.fill_uniform) instead of the unseeded torch.rand."""
        from vr_oracle import fill_uniform
        _, h, w = cert.shape
        return preprocess(fill_uniform(self.seed, self.count, h, w)) * ((cert + np.float32(-1)) * np.float32(-1))

    def first(self, frame_rgb01, image_layers=None):
        self.count += 1
        if image_layers is not None:       # model_img:forward(pre) -- core.lua:146
            out = deprocess(net_forward(image_layers, preprocess(frame_rgb01)))
            self.last = out
            return out
        x = assemble(frame_rgb01, None, None)
        if self.fill_random:
            x[3:6] = self._fill(np.zeros((1,) + x.shape[1:], np.float32))
        out = deprocess(net_forward(self.layers, x))
        self.last = out
        return out

    def next(self, frame_rgb01, backward_flow_uv, cert01):
        self.count += 1
        cert01 = np.ascontiguousarray(cert01, np.float32)
        if self.invert:                                                       # fav.lua:104-106: cert:add(-1):mul(-1)
            cert01 = (cert01 + np.float32(-1)) * np.float32(-1)
        if self.fix:                                                          # fav.lua:79-86,107-110
            ones = np.ones((1,) + cert01.shape, np.float32)
            tmp = warp(ones, flo_to_lua(backward_flow_uv), self.border)[0]
            tmp = np.maximum(np.sign(tmp + np.float32(-0.5)), 0).astype(np.float32)
            cert01 = cert01 * tmp
        cert = min_filter(cert01, self.r)                                     # core:207
        warped = warp(self.last, flo_to_lua(backward_flow_uv), self.border)  # fav.lua:153-158
        x = assemble(frame_rgb01, warped, cert)
        if self.fill_random:
            x[3:6] = self._fill(cert[None]) + x[3:6]
        out = deprocess(net_forward(self.layers, x))
        self.last = out
        return out
