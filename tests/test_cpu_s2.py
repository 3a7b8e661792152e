"""The 3x3 stride-2 layers (csrc/kernels_s2.hip, conv3s2w_kernel; d64 / d128 of models_video.lua:88-92): the host-side weight packing
(csrc/s2_pack.h, compiled here with g++) and a lane-level numpy restatement of the kernel's data movement -- halo pieces of the staging
threads (even / odd column planes, 16-channel chunks at a pitch of 20 floats), fragment addresses of the eight waves, byte offsets into
the packed weights, the 32x32x2 MFMA operand / result layout, store addresses and the per-tile statistics -- checked against a direct
strided correlation.  No GPU: this pins the index arithmetic the HIP kernel is written from; the kernel itself is compared with the
oracle in tests/test_gpu_parity.py."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fast-artistic-videos_amd", "csrc")

SW_P, SW_EW, SW_OW = 20, 33, 32


@pytest.fixture(scope="module")
def packer(tmp_path_factory):
    d = tmp_path_factory.mktemp("s2w")
    src = d / "pack.cpp"
    src.write_text('#include "s2_pack.h"\n#include <cstring>\n'
                   'extern "C" long pack(const float* w, int cin, int cout, float* out) {\n'
                   '  std::vector<float> v; fav::conv_s2w_pack(w, cin, cout, v); if (out) memcpy(out, v.data(), v.size() * 4); return (long)v.size(); }\n')
    so = d / "libs2w.so"
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-I", CSRC, "-o", str(so), str(src)])
    lib = ctypes.CDLL(str(so))
    lib.pack.restype = ctypes.c_long
    lib.pack.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]

    def pack(w):
        w = np.ascontiguousarray(w, np.float32)
        n = lib.pack(w.ctypes.data, w.shape[1], w.shape[0], None)
        out = np.empty(n, np.float32)
        lib.pack(w.ctypes.data, w.shape[1], w.shape[0], out.ctypes.data)
        return out
    return pack


def geo(TR, NTC=None):
    """SwGeo<NTC, TR> of the kernel: one wave per (tile of 32 output channels, output row)"""
    NTC = NTC if NTC is not None else 8 // TR
    NTH = 64 * NTC * TR
    HR = 2 * TR + 1
    EP = HR * SW_EW
    HP = EP + HR * SW_OW
    PS = NTH // 4
    NPC = (HP + PS - 1) // PS
    return HR, EP, HP, NPC, NPC * PS * SW_P, NTH, PS


def a_off(tap, EP, row, m, h):
    """float offset of the A fragment of lane (m, h) for tap = ky * 3 + kx, output row `row` of the tile (kernel: SW_READ_A_DYN)"""
    ky, kx = divmod(tap, 3)
    aE = ((2 * row) * SW_EW + m) * SW_P + 4 * h
    aO = (EP + (2 * row) * SW_OW + m) * SW_P + 4 * h
    return aO + ky * SW_OW * SW_P if kx == 1 else aE + (ky * SW_EW + (1 if kx == 2 else 0)) * SW_P


def emulate_tile(x, wpk, bias, scale, shift, relu, pad, NTC, TR, ty, tx, OH, OW, f=np.float32):
    IH, IW, CIN = x.shape
    COUT = NTC * 32
    HR, EP, HP, NPC, HB, NTH, PS = geo(TR, NTC)
    NW = NTC * TR
    nch = CIN // 16
    lanes = np.arange(64)
    m, h, n = lanes & 31, lanes >> 5, lanes & 31
    acc = np.zeros((NW, 16, 64), np.float64)
    for chunk in range(nch):
        Hs = np.full(HB, np.nan, f)
        for t in range(NTH):
            c4, p0 = t & 3, t >> 2
            for i in range(NPC):
                pe = p0 + PS * i
                if pe < EP:
                    hy = pe // SW_EW; hx = 2 * (pe - hy * SW_EW)
                elif pe < HP:
                    q = pe - EP; hy = q // SW_OW; hx = 2 * (q - hy * SW_OW) + 1
                else:
                    hy, hx = 0x7000, 0
                iy, ix = 2 * ty * TR - pad + hy, 2 * tx * 32 - pad + hx
                v = np.zeros(4, f)
                if 0 <= iy < IH and 0 <= ix < IW:
                    ch = chunk * 16 + c4 * 4
                    v = x[iy, ix, ch:ch + 4].astype(f)
                    if scale is not None:
                        v = (v * scale[ch:ch + 4] + shift[ch:ch + 4]).astype(f)
                        if relu:
                            v = np.maximum(v, 0)
                dst = (p0 + PS * i) * SW_P + c4 * 4
                assert dst + 4 <= HB
                Hs[dst:dst + 4] = v
        for wave in range(NW):
            nt, row = wave % NTC, wave // NTC
            for kg in range(2):
                for tap in range(9):
                    off = np.array([a_off(tap, EP, row, int(mm), int(hh)) for mm, hh in zip(m, h)]) + kg * 8
                    A = Hs[off[:, None] + np.arange(4)[None, :]]                                          # [lane][4]
                    assert not np.isnan(A).any()
                    woff = lanes * 16 + ((chunk * 2 + kg) * 9 + tap) * NTC * 1024 + nt * 1024
                    assert (woff + 16 <= nch * 18 * NTC * 1024).all()
                    B = wpk[(woff // 4)[:, None] + np.arange(4)[None, :]]
                    for st in range(4):
                        a2 = A[:, st].reshape(2, 32).astype(np.float64)
                        b2 = B[:, st].reshape(2, 32).astype(np.float64)
                        D = a2.T @ b2
                        for r in range(16):
                            acc[wave, r] += D[(r & 3) + 8 * (r >> 2) + 4 * h, n]
    acc = acc.astype(f)
    Y = np.full((TR, 32, COUT), np.nan, f)
    stt = np.zeros((TR, COUT, 2)); wn = np.zeros(TR, np.int64)
    oy0, ox0 = ty * TR, tx * 32
    for wave in range(NW):
        nt, row = wave % NTC, wave // NTC
        co = nt * 32 + n
        oy = oy0 + row
        vals = {int(c): [] for c in set(co)}
        for r in range(16):
            mi = (r & 3) + 8 * (r >> 2)
            for l in range(64):
                ox = ox0 + 4 * h[l] + mi
                if oy < OH and ox < OW:
                    v = acc[wave, r, l] + bias[co[l]]
                    Y[row, ox - ox0, co[l]] = v
                    vals[int(co[l])].append(v)
        for c, v in vals.items():
            v = np.array(v, np.float64)
            mu = v.mean() if v.size else 0.0
            stt[row, c] = (mu, ((v - mu) ** 2).sum())
        if nt == 0:
            wn[row] = len(vals[0])
    cnt = int(wn.sum())
    mean = (wn[:, None] * stt[:, :, 0]).sum(0) / max(cnt, 1)
    m2 = (stt[:, :, 1] + wn[:, None] * (stt[:, :, 0] - mean) ** 2).sum(0)
    return Y, mean, m2, cnt


@pytest.mark.parametrize("cin,cout,tr,pad,ih,iw", [(32, 64, 4, 1, 18, 70), (64, 128, 3, 1, 15, 68), (64, 128, 2, 1, 11, 68), (32, 64, 4, 0, 19, 67)],
                         ids=["d64", "d128-3rows", "d128-2rows", "d64-pad0"])
def test_s2w_lane_level_restatement_matches_direct_convolution(packer, cin, cout, tr, pad, ih, iw):
    rng = np.random.default_rng(17 + cin)
    NTC = cout // 32
    TR = tr
    x = rng.standard_normal((ih, iw, cin)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, cout).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cin).astype(np.float32)
    shift = rng.uniform(-0.5, 0.5, cin).astype(np.float32)
    xin = np.maximum(x * scale + shift, 0)
    xp = np.pad(xin, ((pad, pad), (pad, pad), (0, 0))).astype(np.float64)
    OH, OW = (ih + 2 * pad - 3) // 2 + 1, (iw + 2 * pad - 3) // 2 + 1
    ref = np.zeros((OH, OW, cout))
    for ky in range(3):
        for kx in range(3):
            ref += np.einsum("hwc,oc->hwo", xp[ky:ky + 2 * OH - 1:2, kx:kx + 2 * OW - 1:2], w[:, :, ky, kx].astype(np.float64))
    ref += b
    wpk = packer(w)
    assert wpk.size == (cin // 16) * 18 * NTC * 256
    out = np.full((OH, OW, cout), np.nan, np.float32)
    for ty in range((OH + TR - 1) // TR):
        for tx in range((OW + 31) // 32):
            Y, mean, m2, cnt = emulate_tile(x, wpk, b, scale, shift, True, pad, NTC, TR, ty, tx, OH, OW)
            hh, ww = min(TR, OH - ty * TR), min(32, OW - tx * 32)
            assert np.isnan(Y[hh:]).all() and np.isnan(Y[:, ww:]).all()
            out[ty * TR: ty * TR + hh, tx * 32: tx * 32 + ww] = Y[:hh, :ww]
            blk = ref[ty * TR: ty * TR + hh, tx * 32: tx * 32 + ww].reshape(-1, cout)
            assert cnt == blk.shape[0]
            assert np.abs(mean - blk.mean(0)).max() < 1e-5 and np.abs(m2 - ((blk - blk.mean(0)) ** 2).sum(0)).max() < 1e-3 * max(1.0, m2.max())
    err = np.abs(out - ref).max()
    assert err < 2e-5, err


def test_s2w_fragment_reads_are_bank_conflict_free():
    """ds_read_b128 is served in groups of 16 lanes; with a pixel pitch of 5 sixteen-byte slots the 16 pixels of a group hit 16 slots"""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    lanes = np.arange(64)
    for TR, NTC in ((4, 2), (2, 4), (3, 4)):
        _, EP, _, _, _, _, _ = geo(TR, NTC)
        for row in range(TR):
            for tap in range(9):
                for kg in range(2):
                    addr = np.array([a_off(tap, EP, row, int(l & 31), int(l >> 5)) for l in lanes]) * 4 + kg * 32
                    assert (addr % 16 == 0).all()
                    for g in groups:
                        assert len(set((addr[g] // 16) % 16)) == 16
