"""CPU restatement of the 360-degree cube-map orchestration -- TEST INFRASTRUCTURE ONLY (see oracle.py's header).

Follows fast_artistic_video_vr.lua (callbacks of run_fast_neural_video, fast_artistic_video_core.lua:189-229) and
fast_artistic_video/vr_helper.lua, with dtype = torch.CudaTensor (fp32 tensors; the maps are built in Lua doubles and
cast).  PARITY UNPINNED: Lua/Torch7 cannot run here and the reference ships no fixtures for this path.  Assumptions
that are [recalled] rather than read from the tree:
  * a non-integer Lua number used as a tensor index is truncated toward zero (luaL_checklong) -- matters for the
    perspective maps, whose loop variable is fractional (vr_helper.lua:15-22);
  * numeric `for` loops advance by repeated addition of the step;
  * torch.median over 9 elements returns the 5th smallest.
`-fill_occlusions uniform-random` is unseeded torch.rand in the reference (core.lua:108-117); here it is the counter
RNG `fill_uniform` below (documented in DESIGN.md), identical in the HIP path.
"""
import math
from typing import List, Optional

import numpy as np

import oracle as O

F = np.float32
PROC_ORDER = [6, 1, 2, 5, 3, 4]          # fast_artistic_video_vr.lua:103 (file ids of modes 0..5)


# ----------------------------------------------------------------------------------------------- vr_helper.lua
def _trunc_index(v: float) -> int:
    return int(v)                         # luaL_checklong: C cast, toward zero [recalled]


def _lua_range(start: float, stop: float):
    x = start
    while x <= stop:
        yield x
        x = x + 1


def _map_width(height, crop):
    oversize = crop / 2                   # vr_helper.lua:4-5 (both oversizes default to crop/2)
    width = height / 2 / ((2 * oversize + height) / height)          # :6
    max_resize_factor = (width + oversize) / width                  # :7
    width = width - (max_resize_factor - 1) / max_resize_factor * oversize   # :8
    return width, oversize


def warp_map_left(height: int, crop_w: int, orig_width: int) -> np.ndarray:
    """vr_helper.lua:3-25.  Returns [2][height][orig_width] fp32 (dy, dx), 99999 where undefined."""
    width, ov = _map_width(height, crop_w)
    m = np.full((2, height, orig_width), 99999.0, np.float64)
    mid_y = height / 2
    for x in _lua_range(width - crop_w + 1, width):
        rh = (x + ov) / width
        rw = (x + ov) / width
        col = _trunc_index(x - (width - crop_w) + orig_width - crop_w) - 1
        for y in range(1, height + 1):
            m[0, y - 1, col] = (mid_y - y) * (-1 / rh + 1)
            m[1, y - 1, col] = (width - x - ov) * (rw - 1) / rw - orig_width + crop_w
    return m.astype(F)


def warp_map_right(height: int, crop_w: int, org_width: int) -> np.ndarray:
    """vr_helper.lua:27-48."""
    width, ov = _map_width(height, crop_w)
    m = np.full((2, height, org_width), 99999.0, np.float64)
    mid_y = height / 2
    for x in range(1, crop_w + 1):
        rh = (width - x + ov) / width
        rw = (width - x + ov) / width
        for y in range(1, height + 1):
            m[0, y - 1, x - 1] = (mid_y - y) * (-1 / rh + 1)
            m[1, y - 1, x - 1] = -(x - ov) * (rw - 1) / rw + org_width - crop_w
    return m.astype(F)


def warp_map_top(width: int, crop_h: int, orig_height: int) -> np.ndarray:
    """vr_helper.lua:50-72."""
    height, ov = _map_width(width, crop_h)
    m = np.full((2, orig_height, width), 99999.0, np.float64)
    mid_x = width / 2
    for y in _lua_range(height - crop_h + 1, height):
        rw = (y + ov) / height
        rh = (y + ov) / height
        row = _trunc_index(y - (height - crop_h) + orig_height - crop_h) - 1
        for x in range(1, width + 1):
            m[0, row, x - 1] = (height - y - ov) * (rh - 1) / rh - orig_height + crop_h
            m[1, row, x - 1] = (mid_x - x) * (-1 / rw + 1)
    return m.astype(F)


def warp_map_bottom(width: int, crop_h: int, orig_height: int) -> np.ndarray:
    """vr_helper.lua:75-96."""
    height, ov = _map_width(width, crop_h)
    m = np.full((2, orig_height, width), 99999.0, np.float64)
    mid_x = width / 2
    for y in range(1, crop_h + 1):
        rw = (height - y + ov) / height
        rh = (height - y + ov) / height
        for x in range(1, width + 1):
            m[0, y - 1, x - 1] = -(y - ov) * (rh - 1) / rh + orig_height - crop_h
            m[1, y - 1, x - 1] = (mid_x - x) * (-1 / rw + 1)
    return m.astype(F)


def equirect_map(w_plus: int, h_plus: int, overlap_w: float, overlap_h: float, out_w: int, out_h: int) -> np.ndarray:
    """vr_helper.lua:99-184 (cube layout f, l, r, b, u, d side by side).  [2][out_h][out_w] fp32 offsets."""
    m = np.empty((2, out_h, out_w), np.float64)
    cw = w_plus - overlap_w
    ch = h_plus - overlap_h
    for j in range(out_h):
        v = 1 - (j / out_h)
        theta = v * math.pi
        for i in range(out_w):
            u = i / out_w
            phi = u * 2 * math.pi
            x = math.sin(phi) * math.sin(theta) * -1
            y = math.cos(theta)
            z = math.cos(phi) * math.sin(theta) * -1
            a = max(abs(x), abs(y), abs(z))
            xa, ya, za = x / a, y / a, z / a
            if xa == 1:
                xp = (((za + 1) / 2) - 1) * cw; xo = 2 * w_plus; yp = ((ya + 1) / 2) * ch
            elif xa == -1:
                xp = ((za + 1) / 2) * cw; xo = 1 * w_plus; yp = ((ya + 1) / 2) * ch
            elif ya == 1:
                xp = ((xa + 1) / 2) * cw; xo = 5 * w_plus; yp = (((za + 1) / 2) - 1) * ch
            elif ya == -1:
                xp = ((xa + 1) / 2) * cw; xo = 4 * w_plus; yp = ((za + 1) / 2) * ch
            elif za == 1:
                xp = ((xa + 1) / 2) * cw; xo = 0 * w_plus; yp = ((ya + 1) / 2) * ch
            elif za == -1:
                xp = (((xa + 1) / 2) - 1) * cw; xo = 3 * w_plus; yp = ((ya + 1) / 2) * ch
            else:
                xp = 0; yp = 0; xo = 0
            xp = abs(xp); yp = abs(yp)
            xp = xp + xo + overlap_w / 2
            yp = yp + 0 + overlap_h / 2
            m[0, j, i] = yp - j
            m[1, j, i] = xp - i
    return m.astype(F)


# ----------------------------------------------------------------------------------------------- utils.lua
def grad_w_inc(h, w):   # utils.lua make_gradient_mask_w_inc: i/(w+1), i = 1..w, along x
    return np.broadcast_to((np.arange(1, w + 1, dtype=np.float64) / (w + 1))[None, :], (h, w)).copy()


def grad_w_dec(h, w):
    return np.broadcast_to((np.arange(w, 0, -1, dtype=np.float64) / (w + 1))[None, :], (h, w)).copy()


def grad_h_inc(h, w):
    return np.broadcast_to((np.arange(1, h + 1, dtype=np.float64) / (h + 1))[:, None], (h, w)).copy()


def grad_h_dec(h, w):
    return np.broadcast_to((np.arange(h, 0, -1, dtype=np.float64) / (h + 1))[:, None], (h, w)).copy()


def median_filter(img: np.ndarray, r: int) -> np.ndarray:
    """utils.lua:151-159: r x r windows without padding -> [3][H-r+1][W-r+1]; lower median."""
    c, h, w = img.shape
    hh, ww = h - r + 1, w - r + 1
    win = np.empty((c, hh, ww, r * r), F)
    for dy in range(r):
        for dx in range(r):
            win[..., dy * r + dx] = img[:, dy:dy + hh, dx:dx + ww]
    win.sort(axis=-1)
    return np.ascontiguousarray(win[..., (r * r - 1) // 2])


def rotate90(t):        # fast_artistic_video_vr.lua:134-136: reverse(transpose(2,3), dim 2)
    return np.ascontiguousarray(np.transpose(t, (0, 2, 1))[:, ::-1, :])


def rotate_minus90(t):  # :138-140: reverse(transpose(2,3), dim 3)
    return np.ascontiguousarray(np.transpose(t, (0, 2, 1))[:, :, ::-1])


def rotate180(t):       # :142-144
    return np.ascontiguousarray(t[:, ::-1, ::-1])


def fill_uniform(seed: int, index: int, h: int, w: int) -> np.ndarray:
    """Documented stand-in for the reference's unseeded torch.rand(1,3,h,w) (core.lua:109): a counter RNG,
    u = (hash(seed, index, c, y, x) >> 8) * 2^-24 in [0,1), c in RGB order."""
    c, y, x = np.meshgrid(np.arange(3, dtype=np.uint64), np.arange(h, dtype=np.uint64), np.arange(w, dtype=np.uint64), indexing="ij")
    k = (np.uint64(seed) * np.uint64(0x9E3779B1) + np.uint64(index) * np.uint64(0x85EBCA77)
         + c * np.uint64(0xC2B2AE3D) + y * np.uint64(0x27D4EB2F) + x * np.uint64(0x165667B1)) & np.uint64(0xFFFFFFFF)
    k ^= k >> np.uint64(15); k = (k * np.uint64(0x2C1B3C6D)) & np.uint64(0xFFFFFFFF)
    k ^= k >> np.uint64(12); k = (k * np.uint64(0x297A2D39)) & np.uint64(0xFFFFFFFF)
    k ^= k >> np.uint64(15)
    return ((k >> np.uint64(8)).astype(np.float64) * (1.0 / 16777216.0)).astype(F)


# ----------------------------------------------------------------------------------------------- the pipeline
class VRStylizer:
    """State machine of fast_artistic_video_vr.lua: call face(i, ...) for i = 1, 2, ... (6 faces per frame, processing
    order PROC_ORDER); after every 6th face `blended`, `filtered`, `equi`, `cubemap` hold the frame's outputs."""

    def __init__(self, layers, hplus: int, wplus: int, overlap_w: int = 20, overlap_h: int = 20, min_filter_r: int = 7,
                 median: int = 3, out_equi_w: int = 0, out_equi_h: int = 0, fill_random: bool = False, seed: int = 1,
                 image_layers=None, create_inconsistent: bool = False, create_inconsistent_border: bool = False):
        self.layers, self.image_layers = layers, image_layers
        self.hp, self.wp, self.ow, self.oh = hplus, wplus, overlap_w, overlap_h
        self.r, self.median = min_filter_r, median
        self.fill_random, self.seed = fill_random, seed
        self.inconsistent, self.inconsistent_border = create_inconsistent, create_inconsistent_border
        hp, wp = hplus, wplus
        ones = np.ones((1, hp, wp), F)
        # fast_artistic_video_vr.lua:164-198
        self.map_left = warp_map_left(hp, overlap_w, wp);   self.mask_left = O.warp(ones, self.map_left)
        self.map_top = warp_map_top(wp, overlap_h, hp);     self.mask_top = O.warp(ones, self.map_top)
        self.map_bottom = warp_map_bottom(wp, overlap_h, hp); self.mask_bottom = O.warp(ones, self.map_bottom)
        self.map_right = warp_map_right(hp, overlap_w, wp); self.mask_right = O.warp(ones, self.map_right)
        s = ((self.mask_left + self.mask_right) + self.mask_top) + self.mask_bottom
        self.mask_all_div = np.maximum(s, F(1)); self.mask_all = np.minimum(s, F(1))
        gh, gw = overlap_h - 10, overlap_w - 10
        z = np.zeros
        self.g_left = np.concatenate([grad_w_dec(hp, gw), z((hp, wp - gw))], 1)[None]
        self.g_right = np.concatenate([z((hp, wp - gw)), grad_w_inc(hp, gw)], 1)[None]
        self.g_top = np.concatenate([grad_h_dec(gh, wp), z((hp - gh, wp))], 0)[None]
        self.g_bottom = np.concatenate([z((hp - gh, wp)), grad_h_inc(gh, wp)], 0)[None]
        self.g_all = np.maximum(np.maximum(self.g_left, self.g_right), np.maximum(self.g_top, self.g_bottom))
        self.g_lr = np.maximum(self.g_left, self.g_right)
        self.equi_map = None
        if out_equi_w > 0:
            rr = median // 2
            self.equi_map = equirect_map(hp - 2 * rr, wp - 2 * rr, overlap_w - rr, overlap_h - rr, out_equi_w, out_equi_h)
        self.last = [None] * 6          # last_segments
        self.prev = [None] * 6          # prev_last_segments (blended faces of the previous frame)
        self.blended = self.filtered = self.equi = self.cubemap = None

    # fast_artistic_video_vr.lua:204-237 (+ the min filter of core.lua:207)
    def _cert(self, i, mode, cert_frame01):
        b = np.zeros((1, self.hp, self.wp), F)
        if not self.inconsistent_border:
            if mode in (1, 3, 4, 5): b = np.maximum(b, self.mask_left)
            if mode in (2, 3, 4, 5): b = np.maximum(b, self.mask_right)
            if mode in (4, 5): b = np.maximum(b, self.mask_top)
            if mode in (4, 5): b = np.maximum(b, self.mask_bottom)
        if i >= 7 and not self.inconsistent:
            c = np.maximum(np.asarray(cert_frame01, F)[None], b)
        else:
            c = b
        return O.min_filter(c[0], self.r)[None]

    # :239-302
    def _prior(self, i, mode, cert, flow_uv):
        L, w = self.last, O.warp
        d = self.mask_all_div
        border = np.zeros((3, self.hp, self.wp), F)
        if not self.inconsistent_border:
            if mode == 1:
                border = w(L[0], self.map_left)
            elif mode == 2:
                border = w(L[0], self.map_right)
            elif mode == 3:
                border = w(L[1], self.map_left) + w(L[2], self.map_right)
            elif mode == 4:
                border = w(rotate90(L[1]), self.map_left) / d
                border = border + w(rotate_minus90(L[2]), self.map_right) / d
                border = border + w(L[3], self.map_top) / d
                border = border + w(rotate180(L[0]), self.map_bottom) / d
            elif mode == 5:
                border = w(rotate_minus90(L[1]), self.map_left) / d
                border = border + w(rotate90(L[2]), self.map_right) / d
                border = border + w(rotate180(L[0]), self.map_top) / d
                border = border + w(L[3], self.map_bottom) / d
        if i >= 7 and not self.inconsistent:
            lfw = w(self.prev[mode], O.flo_to_lua(flow_uv))
            if mode == 0:
                return lfw
            cert_inv = F(1) - cert
            g = [self.g_right, self.g_left, self.g_lr, self.g_all, self.g_all][mode - 1].astype(F)
            mk = [self.mask_left, self.mask_right, self.mask_left + self.mask_right, self.mask_all, self.mask_all][mode - 1]
            mask = np.maximum(g, np.ceil(g) * cert_inv) * mk
            anti = F(1) - mask
            return lfw * anti + border * mask
        return border

    def _fill(self, i, cert):
        """generate_fill (core.lua:108-117)."""
        if not self.fill_random:
            return np.zeros((3, self.hp, self.wp), F)
        rnd = O.preprocess(fill_uniform(self.seed, i, self.hp, self.wp))
        cert_inv = (cert + F(-1)) * F(-1)
        return rnd * cert_inv

    def face(self, i: int, frame_rgb01: np.ndarray, flow_uv: Optional[np.ndarray] = None,
             cert_frame01: Optional[np.ndarray] = None) -> np.ndarray:
        mode = (i - 1) % 6
        single = (i % 6 == 1) if self.inconsistent else (i == 1)            # :304-310
        pre = O.preprocess(frame_rgb01)
        n = self.hp * self.wp
        if single:
            if self.image_layers is not None:                                # core.lua:146
                out = O.deprocess(O.net_forward(self.image_layers, pre))
            else:                                                            # core.lua:133-138
                x = np.concatenate([pre, self._fill(i, np.zeros((1, self.hp, self.wp), F)),
                                    np.zeros((1, self.hp, self.wp), F)], 0)
                out = O.deprocess(O.net_forward(self.layers, x))
        else:
            cert = self._cert(i, mode, cert_frame01)
            prior = self._prior(i, mode, cert, flow_uv)
            pm = O.preprocess(prior) * cert                                  # core.lua:165-166
            x = np.concatenate([pre, self._fill(i, cert) + pm, cert], 0)     # :168-169
            out = O.deprocess(O.net_forward(self.layers, np.ascontiguousarray(x, F)))
        self.last[mode] = out                                                # func_save_image :525
        if mode == 5:
            self._finish()
        return out

    # blend_other_sides :454-509
    def _blend(self):
        L, w, d = self.last, O.warp, self.mask_all_div
        g = self.g_all.astype(F); anti = (1.0 - self.g_all).astype(F)      # csub in double, then :type(dtype)

        def comb(a, b, c, e):
            r = a / d; r = r + b / d; r = r + c / d; r = r + e / d
            return r
        ml, mr, mt, mb = self.map_left, self.map_right, self.map_top, self.map_bottom
        B = [
            comb(w(L[1], mr), w(L[2], ml), w(rotate180(L[4]), mb), w(rotate180(L[5]), mt)),
            comb(w(L[0], ml), w(L[3], mr), w(rotate_minus90(L[4]), mb), w(rotate90(L[5]), mt)),
            comb(w(L[0], mr), w(L[3], ml), w(rotate90(L[4]), mb), w(rotate_minus90(L[5]), mt)),
            comb(w(L[1], ml), w(L[2], mr), w(L[4], mb), w(L[5], mt)),
            comb(w(rotate180(L[0]), mb), w(rotate90(L[1]), ml), w(rotate_minus90(L[2]), mr), w(L[3], mt)),
            comb(w(rotate180(L[0]), mt), w(rotate_minus90(L[1]), ml), w(rotate90(L[2]), mr), w(L[3], mb)),
        ]
        return [L[k] * anti + B[k] * g for k in range(6)]

    # func_save_image :511-559
    def _finish(self):
        self.blended = self._blend()
        self.prev = self.blended
        m = self.median
        self.filtered = [median_filter(f, m) if m > 0 else f for f in self.blended]
        fs = self.filtered
        rr = m // 2
        ovw = self.ow // 2 - rr; ovh = self.oh // 2 - rr
        if self.equi_map is not None:
            cat = np.concatenate([fs[0], fs[1], fs[2], fs[3], rotate180(fs[4]), rotate180(fs[5])], 2)
            self.equi = O.warp(np.ascontiguousarray(cat), self.equi_map)
        crop = lambda t: t[:, ovh:self.hp - ovh, ovw:self.wp - ovw]
        self.cubemap = np.ascontiguousarray(np.concatenate(
            [crop(fs[3]), crop(fs[0]), rotate90(crop(fs[4])), rotate_minus90(crop(fs[5])), crop(fs[2]), crop(fs[1])], 2))
