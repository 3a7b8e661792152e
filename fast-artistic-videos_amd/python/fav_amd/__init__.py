"""ctypes binding of libfav (include/fav.h) -- thin plumbing for the tests and bench.py.

The product is the C-ABI shared library ``fast-artistic-videos_amd/libfav.so`` (hand-written HIP for
gfx950 + C++ host code).  This module only moves pointers: device buffers are torch CUDA(=HIP)
tensors, ``tensor.data_ptr()`` and the current torch stream are handed to the C entry points.  There
is NO Python/CPU fallback: if the library is missing, or no HIP device is present, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # fast-artistic-videos_amd/
# FAV_AMD_LIB: another build of the library for THIS binding (tests / A-B scripts load libfav_diag.so, the build whose kernel-selection
# switches are live; the release library reads none of them)
LIB_PATH = os.environ.get("FAV_AMD_LIB") or os.path.join(_PKG, "libfav.so")
DIAG_LIB_PATH = os.path.join(_PKG, "libfav_diag.so")
BORDER_STN, BORDER_CPU = 0, 1

_lib = None


class FavError(RuntimeError):
    pass


def build(verbose: bool = False) -> None:
    """Compile libfav.so and the drop-in executables for gfx950 (hipcc cross-compiles without a GPU)."""
    import subprocess
    subprocess.check_call(["make", "-C", _PKG, "-j8", "all"], stdout=None if verbose else subprocess.DEVNULL)


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FavError(f"{LIB_PATH} is missing: build it with `make -C {_PKG}` "
                       "(python -c 'import __graft_entry__ as g; g.build()'); there is no fallback path")
    L = C.CDLL(LIB_PATH)
    L.fav_last_error.restype = C.c_char_p
    L.fav_consistency_workspace_bytes.restype = C.c_size_t
    L.fav_consistency_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
    L.fav_net_param_count.restype = C.c_longlong
    L.fav_net_param_count.argtypes = [C.c_void_p]
    L.fav_stream_last_mask.restype = C.c_void_p
    L.fav_stream_last_mask.argtypes = [C.c_void_p]
    L.fav_net_destroy.argtypes = [C.c_void_p]; L.fav_net_destroy.restype = None
    L.fav_stream_destroy.argtypes = [C.c_void_p]; L.fav_stream_destroy.restype = None
    L.fav_free_host.argtypes = [C.c_void_p]; L.fav_free_host.restype = None
    for f in (L.fav_png_capacity, L.fav_png_workspace_bytes):
        f.restype = C.c_size_t; f.argtypes = [C.c_int, C.c_int]
    L.fav_vr_destroy.argtypes = [C.c_void_p]; L.fav_vr_destroy.restype = None
    L.fav_png_crc32_combine_host.restype = C.c_uint32; L.fav_png_crc32_combine_host.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    _lib = L
    return L


EXPORTS = [
    "fav_last_error", "fav_version", "fav_device_count", "fav_warp_bdhw_f32", "fav_consistency_workspace_bytes",
    "fav_consistency_u8", "fav_min_filter_f32", "fav_assemble_input_f32", "fav_net_create", "fav_net_pack_host",
    "fav_net_create_from_blob", "fav_net_destroy", "fav_net_describe_host", "fav_t7_describe_host", "fav_net_param_count",
    "fav_net_output_size", "fav_net_forward", "fav_net_profile_enable", "fav_net_profile_read_host",
    "fav_conv2d_nchw_f32", "fav_stream_create", "fav_stream_destroy",
    "fav_stream_set_image_net", "fav_stream_first_frame", "fav_stream_next_frame_cert", "fav_stream_next_frame_flow", "fav_stream_prefetch_mask",
    "fav_stream_get_state",
    "fav_stream_set_state", "fav_stream_last_mask", "fav_stream_get_input_f32", "fav_stream_output_size", "fav_stream_set_host_ordered",
    "fav_png_capacity", "fav_png_workspace_bytes", "fav_png_encode_rgb8", "fav_png_encode_f32", "fav_stream_encode_png", "fav_stream_encode_png_async", "fav_stream_wait_png", "fav_png_tables_host", "fav_png_crc32_combine_host", "fav_read_flo_host", "fav_read_pnm_host", "fav_write_pgm_host",
    "fav_write_png_rgb8_host", "fav_free_host",
    "fav_vr_create", "fav_vr_destroy", "fav_vr_face", "fav_vr_finish_frame", "fav_vr_output_sizes", "fav_vr_get_f32",
    "fav_vr_map_host", "fav_temporal_loss_host", "fav_sequential_sum_f32", "fav_read_flo_into_host", "fav_read_pnm_into_host", "fav_net_set_precision", "fav_net_check", "fav_net_set_shared_device", "fav_net_forget_stream",
]


def _check(rc: int) -> None:
    if rc != 0:
        raise FavError(f"libfav error {rc}: {lib().fav_last_error().decode(errors='replace')}")


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise FavError("no HIP device visible to torch: libfav has no CPU fallback")
    return torch


def _p(t) -> C.c_void_p:
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream() -> C.c_void_p:
    return C.c_void_p(_torch().cuda.current_stream().cuda_stream)


def _chk_f32(t, name):
    torch = _torch()
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise FavError(f"{name} must be a contiguous float32 CUDA tensor")   # BilinearSamplerBDHW.lua:26-42


def device_count() -> int:
    n = lib().fav_device_count()
    if n < 0:
        _check(n)
    return n


# ------------------------------------------------------------------------------------------------ ops
def warp(img, flow, border: int = BORDER_STN):
    """nn.BilinearSamplerBDHW():forward({img, flow}) (stnbdhw/BilinearSamplerBDHW.lua:54-82): 3-D inputs get a
    batch dimension added and removed."""
    torch = _torch(); _chk_f32(img, "img"); _chk_f32(flow, "flow")
    squeeze = img.dim() == 3
    if squeeze:
        img, flow = img[None], flow[None]
    b, c, h, w = img.shape
    assert flow.shape[0] == b and flow.shape[1] == 2
    ho, wo = flow.shape[2], flow.shape[3]
    out = torch.empty((b, c, ho, wo), dtype=torch.float32, device=img.device)
    _check(lib().fav_warp_bdhw_f32(_p(img), _p(flow), _p(out), b, c, h, w, ho, wo, border, _stream()))
    return out[0] if squeeze else out


def consistency(flow1_flo, flow2_flo, rgb_hwc=None):
    """consistencyChecker flow1.flo flow2.flo out.pgm [img.ppm]: inputs are the .flo payloads [H][W][2]."""
    torch = _torch(); _chk_f32(flow1_flo, "flow1"); _chk_f32(flow2_flo, "flow2")
    h, w, _ = flow1_flo.shape
    out = torch.empty((h, w), dtype=torch.uint8, device=flow1_flo.device)
    ws_bytes = lib().fav_consistency_workspace_bytes(w, h, 1 if rgb_hwc is not None else 0)
    ws = torch.empty((max(ws_bytes, 1),), dtype=torch.uint8, device=flow1_flo.device)
    _check(lib().fav_consistency_u8(_p(flow1_flo), _p(flow2_flo), _p(rgb_hwc), _p(out), w, h, _p(ws),
                                    C.c_size_t(ws_bytes), _stream()))
    return out


def min_filter(cert, r: int = 7):
    torch = _torch(); _chk_f32(cert, "cert")
    out = torch.empty_like(cert)
    _check(lib().fav_min_filter_f32(_p(cert), _p(out), cert.shape[-2], cert.shape[-1], r, _stream()))
    return out


def assemble(frame_rgb, warped_rgb=None, cert=None):
    torch = _torch(); _chk_f32(frame_rgb, "frame")
    _, h, w = frame_rgb.shape
    out = torch.empty((7, h, w), dtype=torch.float32, device=frame_rgb.device)
    _check(lib().fav_assemble_input_f32(_p(frame_rgb), _p(warped_rgb), _p(cert), _p(out), h, w, _stream()))
    return out


def conv2d(x, weight, bias=None, stride=1, pad=0, gamma=None, beta=None, eps=1e-5, relu=False):
    torch = _torch(); _chk_f32(x, "x"); _chk_f32(weight, "weight")
    cin, h, w = x.shape; cout, _, k, _ = weight.shape
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    out = torch.empty((cout, oh, ow), dtype=torch.float32, device=x.device)
    _check(lib().fav_conv2d_nchw_f32(_p(x), cin, h, w, _p(weight), _p(bias), cout, k, stride, pad, _p(gamma), _p(beta),
                                     C.c_float(eps), 1 if relu else 0, _p(out), _stream()))
    return out


# ------------------------------------------------------------------------------------------------ network
class Net:
    def __init__(self, t7_path: Optional[str] = None, device: int = 0, blob: Optional[bytes] = None):
        h = C.c_void_p()
        if blob is not None:
            _check(lib().fav_net_create_from_blob(blob, C.c_size_t(len(blob)), device, C.byref(h)))
        else:
            _check(lib().fav_net_create(t7_path.encode(), device, C.byref(h)))
        self.h, self.device = h, device

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            try:
                lib().fav_net_destroy(self.h)
            except Exception:      # interpreter shutdown
                pass
            self.h = None

    __del__ = close

    def check(self):
        """fav_net_check: raises if a stream-K hand-off timed out since the last check (call after synchronising)"""
        _torch().cuda.synchronize()
        _check(lib().fav_net_check(self.h))

    def set_shared_device(self, shared: bool):
        """True: data-parallel convolution grids only (no stream-K hand-offs): for a GPU this process does not own exclusively"""
        _check(lib().fav_net_set_shared_device(self.h, 1 if shared else 0))

    def set_precision(self, bf16_operands: bool):
        """False: fp32 parity mode (default); True: bf16 operands in the 3x3 halo convolutions (fast mode, not bit-compatible)."""
        _check(lib().fav_net_set_precision(self.h, 1 if bf16_operands else 0))

    def describe(self) -> str:
        buf = C.create_string_buffer(1 << 16)
        _check(lib().fav_net_describe_host(self.h, buf, C.c_size_t(len(buf))))
        return buf.value.decode()

    def param_count(self) -> int:
        return int(lib().fav_net_param_count(self.h))

    def output_size(self, h: int, w: int) -> Tuple[int, int]:
        ho, wo = C.c_int(), C.c_int()
        _check(lib().fav_net_output_size(self.h, h, w, C.byref(ho), C.byref(wo)))
        return ho.value, wo.value

    def profile_enable(self, on: bool = True):
        _check(lib().fav_net_profile_enable(self.h, 1 if on else 0))

    def profile_read(self):
        """[(ms_sum, launches, useful_macs_per_launch, n_tile)] per convolution, network order"""
        cap = 256
        cnt = C.c_int(); ms = (C.c_double * cap)(); n = (C.c_int * cap)(); macs = (C.c_double * cap)(); tile = (C.c_int * cap)()
        _check(lib().fav_net_profile_read_host(self.h, cap, C.byref(cnt), ms, n, macs, tile))
        return [(ms[i], n[i], macs[i], tile[i]) for i in range(cnt.value)]

    def forward(self, in7):
        torch = _torch(); _chk_f32(in7, "in7")
        x = in7[0] if in7.dim() == 4 else in7
        _, h, w = x.shape
        ho, wo = self.output_size(h, w)
        out = torch.empty((3, ho, wo), dtype=torch.float32, device=x.device)
        _check(lib().fav_net_forward(self.h, _p(x), _p(out), h, w, _stream()))
        return out[None] if in7.dim() == 4 else out


def describe_t7(t7_path: str) -> str:
    """host-only: layer list as parsed by the product's C++ .t7 reader"""
    buf = C.create_string_buffer(1 << 16)
    _check(lib().fav_t7_describe_host(t7_path.encode(), buf, C.c_size_t(len(buf))))
    return buf.value.decode()


def describe_layers(layers, indent: int = 0) -> str:
    """the same text from the Python reader's layer list (fav_amd.t7.extract_layers)"""
    out = []
    pad = "  " * indent
    for L in layers:
        t = L["type"]
        if t == "pad": out.append(f"{pad}{'replicate-pad' if L.get('mode') == 'replicate' else 'pad'} {L['l']} {L['r']} {L['t']} {L['b']}")
        elif t == "fullconv":
            ci, co, k, _ = L["w"].shape
            out.append(f"{pad}fullconv {ci} {co} {k} {L['stride']} {L['pad']} adj={L['adj']} bias={0 if L['b'] is None else 1}")
        elif t == "bn": out.append(f"{pad}bn {len(L['mean'])}")
        elif t == "conv":
            co, ci, k, _ = L["w"].shape
            out.append(f"{pad}conv {ci} {co} {k} {L['stride']} {L['pad']} bias={0 if L['b'] is None else 1}")
        elif t == "in": out.append(f"{pad}in {len(L['gamma'])}")
        elif t == "relu": out.append(f"{pad}relu")
        elif t == "res":
            out.append(f"{pad}res shave={L['shave']}"); out.append(describe_layers(L["block"], indent + 1).rstrip("\n"))
        elif t == "up": out.append(f"{pad}up {L['s']}")
        elif t == "tanh": out.append(f"{pad}tanh")
        elif t == "mul": out.append(f"{pad}mul {L['k']:g}")
        else: out.append(f"{pad}identity")
    return "\n".join(out) + "\n"


def pack_checkpoint(t7_path: str) -> bytes:
    n = C.c_size_t()
    _check(lib().fav_net_pack_host(t7_path.encode(), None, C.c_size_t(0), C.byref(n)))
    buf = C.create_string_buffer(n.value)
    _check(lib().fav_net_pack_host(t7_path.encode(), buf, C.c_size_t(n.value), C.byref(n)))
    return buf.raw[:n.value]


class _Opts(C.Structure):
    _fields_ = [("border_mode", C.c_int), ("occlusions_min_filter", C.c_int), ("invert_occlusion", C.c_int),
                ("fix_occlusions", C.c_int), ("fill_random", C.c_int), ("seed", C.c_uint)]


class Stream:
    """One video stream: the recurrent per-frame pipeline (fast_artistic_video_core.lua:194-211)."""

    def __init__(self, net: Net, h: int, w: int, border: int = BORDER_STN, min_filter_r: int = 7,
                 invert_occlusion: bool = False, fix_occlusions: bool = False, fill_random: bool = False, seed: int = 1):
        self.net, self.H, self.W = net, h, w
        o = _Opts(border, min_filter_r, int(invert_occlusion), int(fix_occlusions), int(fill_random), seed)
        hd = C.c_void_p()
        _check(lib().fav_stream_create(net.h, h, w, C.byref(o), C.byref(hd)))
        self.h = hd
        ho, wo = C.c_int(), C.c_int()
        _check(lib().fav_stream_output_size(hd, C.byref(ho), C.byref(wo)))
        self.Ho, self.Wo = ho.value, wo.value          # size of the stylised frames (= H x W when both are multiples of 4)

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            try:
                lib().fav_stream_destroy(self.h)
            except Exception:      # interpreter shutdown
                pass
            self.h = None

    __del__ = close

    def set_image_net(self, img_net: Optional[Net]):
        self._img = img_net      # keep alive
        _check(lib().fav_stream_set_image_net(self.h, img_net.h if img_net is not None else None))

    def _outs(self, dev, want_f32, want_u8):
        torch = _torch()
        f = torch.empty((3, self.Ho, self.Wo), dtype=torch.float32, device=dev) if want_f32 else None
        u = torch.empty((self.Ho, self.Wo, 3), dtype=torch.uint8, device=dev) if want_u8 else None
        return f, u

    def first_frame(self, frame_u8_hwc, want_f32=True, want_u8=False, out_f32=None, out_u8=None):
        f, u = self._outs(frame_u8_hwc.device, want_f32 and out_f32 is None, want_u8 and out_u8 is None)
        f = out_f32 if out_f32 is not None else f; u = out_u8 if out_u8 is not None else u
        _check(lib().fav_stream_first_frame(self.h, _p(frame_u8_hwc), _p(f), _p(u), _stream()))
        return f, u

    def next_frame_cert(self, frame_u8_hwc, backward_flo, cert_u8, want_f32=True, want_u8=False, out_f32=None, out_u8=None):
        f, u = self._outs(frame_u8_hwc.device, want_f32 and out_f32 is None, want_u8 and out_u8 is None)
        f = out_f32 if out_f32 is not None else f; u = out_u8 if out_u8 is not None else u
        _check(lib().fav_stream_next_frame_cert(self.h, _p(frame_u8_hwc), _p(backward_flo), _p(cert_u8), _p(f), _p(u), _stream()))
        return f, u

    def next_frame_flow(self, frame_u8_hwc, backward_flo, forward_flo, use_structure=False, want_f32=True, want_u8=False,
                        out_f32=None, out_u8=None):
        f, u = self._outs(frame_u8_hwc.device, want_f32 and out_f32 is None, want_u8 and out_u8 is None)
        f = out_f32 if out_f32 is not None else f; u = out_u8 if out_u8 is not None else u
        _check(lib().fav_stream_next_frame_flow(self.h, _p(frame_u8_hwc), _p(backward_flo), _p(forward_flo),
                                                1 if use_structure else 0, _p(f), _p(u), _stream()))
        return f, u

    def prefetch_mask(self, frame_u8_hwc, backward_flo, forward_flo, use_structure=False):
        _check(lib().fav_stream_prefetch_mask(self.h, _p(frame_u8_hwc), _p(backward_flo), _p(forward_flo),
                                              1 if use_structure else 0, _stream()))

    def state(self):
        torch = _torch()
        out = torch.empty((3, self.Ho, self.Wo), dtype=torch.float32, device=f"cuda:{self.net.device}")
        _check(lib().fav_stream_get_state(self.h, _p(out), _stream()))
        return out

    def set_state(self, t):
        _chk_f32(t, "state")
        _check(lib().fav_stream_set_state(self.h, _p(t), _stream()))

    def set_host_ordered(self, on: bool):
        """look-ahead without events: the caller synchronises the inputs itself before prefetch_mask (see include/fav.h)"""
        _check(lib().fav_stream_set_host_ordered(self.h, 1 if on else 0))

    def png_buffers(self):
        """(out, nbytes) device buffers for encode_png_into: capacity fav_png_capacity(W, H), one int32"""
        torch = _torch()
        cap = lib().fav_png_capacity(self.Wo, self.Ho)
        dev = torch.device("cuda", self.net.device)
        return torch.empty((cap + 3) // 4 * 4, dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)

    def encode_png_into(self, out, nbytes):
        """fav_stream_encode_png: the current stylised frame as the bytes of a PNG file, enqueued on the current stream (no sync)"""
        _check(lib().fav_stream_encode_png(self.h, _p(out), C.c_size_t(out.numel()), _p(nbytes), _stream()))

    def encode_png_async_into(self, out, nbytes):
        """fav_stream_encode_png_async: the same bytes, produced on the stream's own encoder queue next to the following frame"""
        _check(lib().fav_stream_encode_png_async(self.h, _p(out), C.c_size_t(out.numel()), _p(nbytes), _stream()))

    def wait_png(self):
        """fav_stream_wait_png: the current stream waits for the asynchronous encodes issued so far"""
        _check(lib().fav_stream_wait_png(self.h, _stream()))

    def last_input(self):
        """[7][H][W] network input of the last frame (content | prior | certainty), un-padded copy of the fused kernel's output"""
        torch = _torch()
        out = torch.empty((7, self.H, self.W), dtype=torch.float32, device=f"cuda:{self.net.device}")
        _check(lib().fav_stream_get_input_f32(self.h, _p(out), _stream()))
        return out

    def last_mask(self):
        """copy of the u8 certainty mask used for the last frame (before the min filter)"""
        torch = _torch()
        ptr = lib().fav_stream_last_mask(self.h)
        out = torch.empty((self.H, self.W), dtype=torch.uint8, device=f"cuda:{self.net.device}")
        # device-to-device copy through torch's runtime (same HIP context)
        src = _from_ptr_u8(ptr, self.H * self.W, self.net.device)
        out.view(-1).copy_(src)
        return out


def png_encode(img, from_stream: "Stream" = None) -> bytes:
    """A9 on the GPU: the bytes of the PNG file (fav_png_encode_rgb8 for a u8 [H][W][3] tensor, fav_png_encode_f32 for a float
    [3][H][W] tensor, fav_stream_encode_png for a Stream's current frame)."""
    torch = _torch()
    if from_stream is not None:
        h, w, dev = from_stream.Ho, from_stream.Wo, torch.device("cuda", from_stream.net.device)
    elif img.dtype == torch.uint8:
        h, w, dev = img.shape[0], img.shape[1], img.device
    else:
        h, w, dev = img.shape[1], img.shape[2], img.device
    cap = lib().fav_png_capacity(w, h)
    out = torch.empty((cap + 3) // 4 * 4, dtype=torch.uint8, device=dev)
    nbytes = torch.zeros(1, dtype=torch.int32, device=dev)
    if from_stream is not None:
        _check(lib().fav_stream_encode_png(from_stream.h, _p(out), C.c_size_t(cap), _p(nbytes), _stream()))
    else:
        wsb = lib().fav_png_workspace_bytes(w, h)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        assert img.is_contiguous()
        fn = lib().fav_png_encode_rgb8 if img.dtype == torch.uint8 else lib().fav_png_encode_f32
        _check(fn(_p(img), w, h, _p(out), C.c_size_t(cap), _p(nbytes), _p(ws), C.c_size_t(wsb), _stream()))
    torch.cuda.synchronize()
    n = int(nbytes.item())
    return bytes(out[:n].cpu().numpy().tobytes())


def png_tables():
    """the encoder's Huffman codes (host-only): list of dicts {len[277], code[277], hdr (uint32 words), hdr_bits, btype, dist_len, dist_code}"""
    cnt, tb = C.c_int(), C.c_int()
    _check(lib().fav_png_tables_host(None, C.c_size_t(0), C.byref(cnt), C.byref(tb)))
    buf = np.zeros(cnt.value * tb.value // 4, np.uint32)
    _check(lib().fav_png_tables_host(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes), C.byref(cnt), C.byref(tb)))
    out = []
    w = tb.value // 4
    for k in range(cnt.value):
        t = buf[k * w:(k + 1) * w]
        out.append({"len": (t[:277] >> 16).astype(int), "code": (t[:277] & 0xFFFF).astype(int), "hdr": t[277:277 + 40].copy(),
                    "hdr_bits": int(t[317]), "btype": int(t[318]), "dist_len": int(t[319]), "dist_code": int(t[320])})
    return out


def sequential_sum(x) -> float:
    """fp32 left-to-right sum with per-step rounding (CMatrix::avg's arithmetic), evaluated by the exact parallel scan."""
    torch = _torch()
    _chk_f32(x, "x")
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    _check(lib().fav_sequential_sum_f32(_p(x), C.c_size_t(x.numel()), _p(out), _stream()))
    return out


def temporal_loss(prev_rgb, cur_rgb, backward_flo, cert_u8, border: int = BORDER_STN) -> float:
    """fast_artistic_video.lua:128-151: MSE between the flow-warped previous and the current stylised frame on reliable pixels."""
    _chk_f32(prev_rgb, "prev_rgb"); _chk_f32(cur_rgb, "cur_rgb")
    _, h, w = prev_rgb.shape
    out = C.c_double()
    _check(lib().fav_temporal_loss_host(_p(prev_rgb), _p(cur_rgb), _p(backward_flo), _p(cert_u8), h, w, border, C.byref(out), _stream()))
    return out.value


class _VROpts(C.Structure):
    _fields_ = [("overlap_w", C.c_int), ("overlap_h", C.c_int), ("occlusions_min_filter", C.c_int), ("median_filter", C.c_int),
                ("fill_random", C.c_int), ("seed", C.c_uint), ("create_inconsistent", C.c_int),
                ("create_inconsistent_border", C.c_int), ("out_equi_w", C.c_int), ("out_equi_h", C.c_int), ("border_mode", C.c_int)]


def vr_map_host(kind: int, hplus: int, wplus: int, overlap: int, median: int = 3, out_w: int = 0, out_h: int = 0) -> np.ndarray:
    """Static maps of the cube-map orchestration (vr_helper.lua), computed on the host: no device needed."""
    shape = (2, out_h, out_w) if kind == 4 else (2, hplus, wplus)
    out = np.empty(shape, np.float32)
    _check(lib().fav_vr_map_host(kind, hplus, wplus, overlap, median, out_w, out_h, out.ctypes.data_as(C.c_void_p)))
    return out


class VR:
    """360-degree cube-map orchestration (fast_artistic_video_vr.lua): six faces per frame, see include/fav.h."""

    def __init__(self, net: Net, hplus: int, wplus: int, overlap_w: int = 20, overlap_h: int = 20, min_filter_r: int = 7,
                 median: int = 3, out_equi_w: int = 0, out_equi_h: int = 0, fill_random: bool = False, seed: int = 1,
                 image_net: Optional[Net] = None, create_inconsistent: bool = False, create_inconsistent_border: bool = False,
                 border: int = BORDER_STN):
        self.net, self.img, self.H, self.W = net, image_net, hplus, wplus
        o = _VROpts(overlap_w, overlap_h, min_filter_r, median, int(fill_random), seed, int(create_inconsistent),
                    int(create_inconsistent_border), out_equi_w, out_equi_h, border)
        hd = C.c_void_p()
        _check(lib().fav_vr_create(net.h, image_net.h if image_net is not None else None, hplus, wplus, C.byref(o), C.byref(hd)))
        self.h = hd
        sz = [C.c_int() for _ in range(6)]
        _check(lib().fav_vr_output_sizes(self.h, *[C.byref(x) for x in sz]))
        self.equi_w, self.equi_h, self.cube_w, self.cube_h, self.filt_w, self.filt_h = [x.value for x in sz]

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            try:
                lib().fav_vr_destroy(self.h)
            except Exception:      # interpreter shutdown
                pass
            self.h = None

    __del__ = close

    def face(self, i: int, frame_u8_hwc, backward_flo=None, cert_u8=None):
        torch = _torch()
        out = torch.empty((3, self.H, self.W), dtype=torch.float32, device=frame_u8_hwc.device)
        _check(lib().fav_vr_face(self.h, i, _p(frame_u8_hwc), _p(backward_flo), _p(cert_u8), _p(out), _stream()))
        return out

    def finish_frame(self, want_equi: bool = True, want_cube: bool = True):
        torch = _torch()
        dev = torch.device("cuda", torch.cuda.current_device())
        e = torch.empty((self.equi_h, self.equi_w, 3), dtype=torch.uint8, device=dev) if (want_equi and self.equi_w) else None
        c = torch.empty((self.cube_h, self.cube_w, 3), dtype=torch.uint8, device=dev) if (want_cube and self.cube_w) else None
        _check(lib().fav_vr_finish_frame(self.h, _p(e), _p(c), _stream()))
        return e, c

    def get(self, which: int, k: int = 0):
        torch = _torch()
        shape = {0: (3, self.H, self.W), 1: (3, self.H, self.W), 2: (3, self.filt_h, self.filt_w),
                 3: (3, self.equi_h, self.equi_w), 4: (3, self.cube_h, self.cube_w)}[which]
        out = torch.empty(shape, dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device()))
        _check(lib().fav_vr_get_f32(self.h, which, k, _p(out), _stream()))
        return out


def _from_ptr_u8(ptr: int, n: int, device: int):
    """wrap a raw device pointer as a torch uint8 tensor (no ownership) via __cuda_array_interface__"""
    torch = _torch()

    class _W:
        __cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(_W(), device=f"cuda:{device}")


# ------------------------------------------------------------------------------------------------ host formats
def read_flo(path: str) -> np.ndarray:
    d = C.POINTER(C.c_float)(); w, h = C.c_int(), C.c_int()
    _check(lib().fav_read_flo_host(path.encode(), C.byref(d), C.byref(w), C.byref(h)))
    try:
        return np.ctypeslib.as_array(d, shape=(h.value, w.value, 2)).copy()
    finally:
        lib().fav_free_host(d)


def read_pnm(path: str) -> np.ndarray:
    d = C.POINTER(C.c_uint8)(); w, h, ch = C.c_int(), C.c_int(), C.c_int()
    _check(lib().fav_read_pnm_host(path.encode(), C.byref(d), C.byref(w), C.byref(h), C.byref(ch)))
    try:
        a = np.ctypeslib.as_array(d, shape=(h.value, w.value, ch.value)).copy()
        return a if ch.value == 3 else a[..., 0]
    finally:
        lib().fav_free_host(d)


def write_pgm(path: str, a: np.ndarray) -> None:
    a = np.ascontiguousarray(a, np.uint8)
    _check(lib().fav_write_pgm_host(path.encode(), a.ctypes.data_as(C.c_void_p), a.shape[1], a.shape[0]))


def write_png(path: str, rgb_hwc: np.ndarray, level: int = 1) -> None:
    a = np.ascontiguousarray(rgb_hwc, np.uint8)
    _check(lib().fav_write_png_rgb8_host(path.encode(), a.ctypes.data_as(C.c_void_p), a.shape[1], a.shape[0], level))
