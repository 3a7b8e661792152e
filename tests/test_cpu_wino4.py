"""Winograd F(4x4,3x3) path of the residual convolutions (csrc/kernels_wino4.hip, round 4): the host-side weight transform / packing
(csrc/wino4_pack.h, compiled here with g++) and a lane-level numpy restatement of the kernel's data movement -- the two transform
passes through LDS (items, addresses, the padded tile-row pitch), the fragment address of every lane, the 16x16x4 MFMA operand /
result layout, the in-lane output transform -- checked against a direct 3x3 correlation.  No GPU: this pins the index arithmetic the
HIP kernel is written from; the kernel itself is compared with the oracle in tests/test_gpu_parity.py."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fast-artistic-videos_amd", "csrc")

VPOS, VBUF, LLINE, LTY, LKQ, LBUF = 256, 36 * 256, 72, 452, 1856, 4 * 1856       # words: kernels_wino4.hip, W4_*


@pytest.fixture(scope="module")
def packer(tmp_path_factory):
    d = tmp_path_factory.mktemp("wino4")
    src = d / "pack.cpp"
    src.write_text('#include "wino4_pack.h"\n#include <cstring>\n'
                   'extern "C" long pack(const float* w, int cin, int cout, float* out) {\n'
                   '  std::vector<float> v; fav::conv_wino4_pack(w, cin, cout, v); if (out) memcpy(out, v.data(), v.size() * 4); return (long)v.size(); }\n'
                   'extern "C" void bt(const double* d, double* v) { fav::Wino4::bt<double>(d, v); }\n'
                   'extern "C" void at(const double* m, double* y) { fav::Wino4::at<double>(m, y); }\n')
    so = d / "libpack4.so"
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-I", CSRC, "-o", str(so), str(src)])
    lib = ctypes.CDLL(str(so))
    lib.pack.restype = ctypes.c_long
    lib.pack.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.bt.argtypes = [ctypes.c_void_p, ctypes.c_void_p]; lib.at.argtypes = [ctypes.c_void_p, ctypes.c_void_p]

    def pack(w):
        cout, cin = w.shape[:2]
        w = np.ascontiguousarray(w, np.float32)
        n = lib.pack(w.ctypes.data, cin, cout, None)
        out = np.empty(n, np.float32)
        lib.pack(w.ctypes.data, cin, cout, out.ctypes.data)
        return out

    def bt(d):
        d = np.ascontiguousarray(d, np.float64); v = np.empty(6)
        lib.bt(d.ctypes.data, v.ctypes.data)
        return v

    def at(m):
        m = np.ascontiguousarray(m, np.float64); y = np.empty(4)
        lib.at(m.ctypes.data, y.ctypes.data)
        return y
    pack.bt, pack.at = bt, at
    return pack


BT = np.array([[81 / 64, 0, -45 / 16, 0, 1, 0], [0, -27 / 16, -9 / 4, 3 / 4, 1, 0], [0, 27 / 16, -9 / 4, -3 / 4, 1, 0],
               [0, -27 / 32, -9 / 16, 3 / 2, 1, 0], [0, 27 / 32, -9 / 16, -3 / 2, 1, 0], [0, 81 / 64, 0, -45 / 16, 0, 1]])
G = np.array([[64 / 81, 0, 0], [-128 / 243, -32 / 81, -8 / 27], [-128 / 243, 32 / 81, -8 / 27], [32 / 243, 16 / 81, 8 / 27], [32 / 243, -16 / 81, 8 / 27], [0, 0, 1]])
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 3 / 4, -3 / 4, 3 / 2, -3 / 2, 0], [0, 9 / 16, 9 / 16, 9 / 4, 9 / 4, 0], [0, 27 / 64, -27 / 64, 27 / 8, -27 / 8, 1]])


def test_transform_matrices_are_a_minimal_filtering_algorithm(packer):
    """Y = A^T [(G g G^T) (.) (B^T d B)] A is the 3x3 correlation of a 6x6 patch for ANY d, g (exactly, in rational arithmetic up to
    double rounding), and the device forms of B^T / A^T (wino4_pack.h Wino4::bt / at) are those matrices"""
    rng = np.random.default_rng(1)
    for _ in range(5):
        d, g = rng.standard_normal((6, 6)), rng.standard_normal((3, 3))
        y = AT @ ((G @ g @ G.T) * (BT @ d @ BT.T)) @ AT.T
        ref = np.array([[(d[a:a + 3, b:b + 3] * g).sum() for b in range(4)] for a in range(4)])
        assert np.abs(y - ref).max() < 1e-12
        v = rng.standard_normal(6)
        assert np.abs(packer.bt(v) - BT @ v).max() < 1e-13 and np.abs(packer.at(v) - AT @ v).max() < 1e-13


def direct_conv(x, w, b):
    IH, IW, _ = x.shape
    OH, OW = IH - 2, IW - 2
    y = np.zeros((OH, OW, w.shape[0]))
    for ky in range(3):
        for kx in range(3):
            y += np.einsum("hwc,oc->hwo", x[ky:ky + OH, kx:kx + OW].astype(np.float64), w[:, :, ky, kx].astype(np.float64))
    return y + b.astype(np.float64)


def bt_rows(d):       # B^T applied to six [..]-shaped operands the way w4_bt does (operation order included)
    e1, o1 = d[4] - 2.25 * d[2], 0.75 * d[3] - 1.6875 * d[1]
    e2, o2 = d[4] - 0.5625 * d[2], 1.5 * d[3] - 0.84375 * d[1]
    return [1.265625 * d[0] + (d[4] - 2.8125 * d[2]), e1 + o1, e1 - o1, e2 + o2, e2 - o2, 1.265625 * d[1] + (d[5] - 2.8125 * d[3])]


def emulate_unit(x, wpk, bias, scale, shift, relu, oy0, ox0, f=np.float32):
    """One work unit (16 x 16 output pixels x 128 channels) the way conv3_wino4_kernel computes it.  Returns Y[16][16][128]."""
    IH, IW, CIN = x.shape
    ns = CIN // 16
    wpk = wpk.reshape(ns, 36, 4, 2, 64, 4)
    lanes = np.arange(64)
    acc = np.zeros((4, 36, 2, 64, 4), f)                      # [wave][position][nt][lane][register r]: the MFMA result registers
    for s in range(ns):
        # stage 1: items (pixel = 18 ty + x, cq) -> six row-transformed lines in L[cq][ty][i][x] (one 16-byte slot each)
        L = np.full(LBUF, np.nan, f)
        seen = set()
        for e in range(288):
            t = e if e < 256 else e - 256
            cq = (t >> 3) & 3
            if e < 256:
                pix = (t >> 5) * 8 + (t & 7); ty = (pix * 3641) >> 16; xx = pix - ty * 18
                assert ty == pix // 18
            else:
                assert t < 32
                ty, xx = 3, 10 + (t & 7)
            assert (ty, xx, cq) not in seen; seen.add((ty, xx, cq))
            rows = []
            for a in range(6):
                iy, ix = min(oy0 + 4 * ty + a, IH - 1), min(ox0 + xx, IW - 1)
                v = x[iy, ix, s * 16 + cq * 4: s * 16 + cq * 4 + 4].astype(f)
                if scale is not None:
                    v = v * scale[s * 16 + cq * 4: s * 16 + cq * 4 + 4] + shift[s * 16 + cq * 4: s * 16 + cq * 4 + 4]
                    if relu:
                        v = np.maximum(v, 0)
                rows.append(v.astype(f))
            dst = cq * LKQ + ty * LTY + xx * 4
            for i, l in enumerate(bt_rows(rows)):
                L[dst + i * LLINE: dst + i * LLINE + 4] = l.astype(f)
        assert len(seen) == 288
        # stage 2: items (tile m, line i, kq) -> six positions in V[6 i + j][kq][m] (one slot each)
        V = np.full(VBUF, np.nan, f)
        items = [(t & 15, (t >> 4) & 3, t >> 6) for t in range(256)] + [(t & 15, (t >> 4) & 3, 2 + (t >> 6)) for t in range(128, 256)]
        assert sorted(items) == sorted((m, kq, i) for m in range(16) for kq in range(4) for i in range(6))
        for m2, kq, i in items:
            src = kq * LKQ + (m2 >> 2) * LTY + 4 * (m2 & 3) * 4 + i * LLINE
            c = [L[src + k * 4: src + k * 4 + 4] for k in range(6)]
            assert not any(np.isnan(v).any() for v in c)
            for j, o in enumerate(bt_rows(c)):
                dst = (6 * i + j) * VPOS + (kq * 16 + m2) * 4
                V[dst: dst + 4] = o.astype(f)
        # matrix instructions: v_mfma_f32_16x16x4_f32: D[i][j] += sum_k A[i][k] B[k][j]; lane l holds A[l & 15][l >> 4], B[l >> 4][l & 15]
        # and, in register r, D[4 (l >> 4) + r][l & 15].  The kernel passes the TRANSFORMED INPUT as A (i = tile) and the weights as B
        # (j = channel inside the 16-group; round 5 -- the other way round before): register r of lane l = tile 4 (l >> 4) + r, channel l & 15
        aA = lanes * 4                                                       # V[p][kq = lane >> 4][m = lane & 15]
        for w in range(4):
            for p in range(36):
                Vf = V[(p * VPOS + aA)[:, None] + np.arange(4)[None, :]]         # [lane][step]
                assert not np.isnan(Vf).any()
                for nt in range(2):
                    Wf = wpk[s, p, w, nt]                                         # [lane][step]
                    for j in range(4):
                        a2 = Wf[:, j].reshape(4, 16)         # [k][channel i]   (lane = k * 16 + i)
                        b2 = Vf[:, j].reshape(4, 16)         # [k][tile j]
                        D = (b2.T.astype(np.float64) @ a2.astype(np.float64)).astype(f)      # [tile][channel]
                        for r in range(4):
                            acc[w, p, nt, :, r] += D[4 * (lanes >> 4) + r, lanes & 15]
    # output transform in the lane: register r of lane l = tile (row g = l >> 4, column r), channel 32 w + 16 nt + (l & 15): the 16
    # adjacent lanes of a group hold 16 consecutive channels of the same pixels -- 64 contiguous bytes per store request
    Y = np.zeros((16, 16, 128), f)
    for w in range(4):
        for nt in range(2):
            for lane in range(64):
                cn, g = lane & 15, lane >> 4
                for r in range(4):
                    M = acc[w, :, nt, lane, r].reshape(6, 6).astype(np.float64)
                    co = 32 * w + 16 * nt + cn
                    Y[4 * g: 4 * g + 4, 4 * r: 4 * r + 4, co] = (AT @ M @ AT.T + bias[co]).astype(f)
    return Y


@pytest.mark.parametrize("cin,affine", [(32, False), (128, True)])
def test_wino4_lane_level_restatement_matches_direct_convolution(packer, cin, affine):
    rng = np.random.default_rng(11 + cin)
    IH, IW = 21, 35                                   # 19 x 33 outputs: units (0, 0), (0, 1), (1, 2) incl. the clamped fringe
    x = rng.standard_normal((IH, IW, cin)).astype(np.float32)
    w = (rng.standard_normal((128, cin, 3, 3)) * np.sqrt(2 / (cin * 9))).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, 128).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cin).astype(np.float32) if affine else None
    shift = rng.uniform(-0.5, 0.5, cin).astype(np.float32) if affine else None
    xin = np.maximum(x * scale + shift, 0) if affine else x
    ref = direct_conv(xin, w, b)
    wpk = packer(w)
    assert wpk.size == (cin // 16) * 36 * 4 * 2 * 64 * 4
    for uy, ux in ((0, 0), (0, 1), (1, 2)):
        Y = emulate_unit(x, wpk, b, scale, shift, affine, uy * 16, ux * 16)
        oh, ow = min(16, 19 - uy * 16), min(16, 33 - ux * 16)
        r = ref[uy * 16: uy * 16 + oh, ux * 16: ux * 16 + ow]
        err = np.abs(Y[:oh, :ow] - r).max()
        assert err < 2e-4, (uy, ux, err)


RGROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
RGROUPS += [[l + 32 for l in g] for g in RGROUPS]


def test_wino4_lds_accesses_are_bank_conflict_free():
    """MI355X LDS (MI355X_MICROARCH.md): a ds_read_b128 is served in four NON-contiguous groups of 16 lanes over 16 sixteen-byte slots
    (64 banks), a ds_write_b128 in eight contiguous groups of 8 lanes over 8 slots (32 banks).  Within a group all slots must differ:
    the A-fragment read of every position, both sides of the column pass, the read-back of the landing area, and the row pass's
    writes (where a group of eight consecutive pixels straddles two tile rows one 2-way conflict remains: at most 3 of the 36 groups)"""
    lanes = np.arange(64)
    rslot = lambda words: (np.asarray(words) // 4) % 16
    wslot = lambda words: (np.asarray(words) // 4) % 8
    aA = lanes * 4
    for g in RGROUPS:
        assert len(set(rslot(aA[g]))) == 16
    # landing area: 16 bytes per thread, consecutive
    for g in RGROUPS:
        assert len(set(rslot((lanes * 4)[g]))) == 16
    for wave in range(4):
        t = wave * 64 + lanes
        m2, kq = t & 15, (t >> 4) & 3
        l2 = kq * LKQ + (m2 >> 2) * LTY + 4 * (m2 & 3) * 4
        for i in range(6):
            for k in range(6):
                for g in RGROUPS:
                    assert len(set(rslot((l2 + i * LLINE + k * 4)[g]))) == 16, ("column pass read", wave, i, k)
        v2 = lanes * 4
        for g0 in range(0, 64, 8):
            assert len(set(wslot(v2[g0: g0 + 8]))) == 8, ("column pass write", g0)
    # row pass writes: item A of all 256 threads, line 0 (the other lines add the same constant to every lane)
    t = np.arange(256)
    cq = (t >> 3) & 3
    pix = (t >> 5) * 8 + (t & 7)
    ty, xx = pix // 18, pix % 18
    l1 = cq * LKQ + ty * LTY + xx * 4
    bad = sum(1 for g0 in range(0, 256, 8) if len(set(wslot(l1[g0: g0 + 8]))) != 8)
    assert bad <= 3 * 4, bad                                  # (pixel groups 16..23, 32..39, 48..55 straddle a tile row, once per chunk)
    assert (LTY // 4) % 16 == 1 and (LKQ // 4) % 16 == 0 and LTY >= 6 * LLINE and LKQ >= 4 * LTY


def test_wino4_row_requests_and_their_read_back():
    """The raw-row requests (buffer_load ... lds, kernels_wino4.hip `hd_`): request lane d < 288 asks for pixel d >> 2, chunk d & 3 -- four
    ADJACENT lanes = the 64 contiguous bytes of one pixel's 16-channel slice -- and lands it in slot d; the row pass's thread
    t = (pix >> 3) * 32 + cq * 8 + (pix & 7) reads slot 4 pix + cq.  Every (pixel, chunk) of the 4 x 18 item grid is requested once, read
    by the thread that owns it, and the read-back is bank-conflict-free for the true ds_read_b128 lane groups."""
    d = np.arange(288)
    dpix, dchunk = d >> 2, d & 3
    assert dpix.max() == 71 and len(set(zip(dpix.tolist(), dchunk.tolist()))) == 288
    # four adjacent request lanes: one pixel, chunks 0..3 = byte offsets 0, 16, 32, 48 of its slice
    for q in range(0, 288, 4):
        assert len(set(dpix[q:q + 4])) == 1 and list(dchunk[q:q + 4]) == [0, 1, 2, 3]
    t = np.arange(288)
    cq = (t >> 3) & 3
    pix = np.minimum((t >> 5) * 8 + (t & 7), 71)
    slot = pix * 4 + cq                                            # rread = Rs + (pix1 * 4 + cq) * 4 words
    assert sorted(slot.tolist()) == list(range(288))               # every landed piece has exactly one reader
    assert np.array_equal(dpix[slot], pix) and np.array_equal(dchunk[slot], cq)       # ... the thread whose item it is
    for w0 in range(0, 256, 64):                                   # whole waves of the row pass (the fifth is half empty)
        s = slot[w0:w0 + 64]
        for g in RGROUPS:
            assert len(set((s[g] % 16).tolist())) == 16, ("read-back", w0)


def _shares(units, nslices, shares):
    """restatement of the stream-K dealing in conv3_wino4_kernel (`p.stream`): share sh = slices [tot sh / shares, tot (sh + 1) / shares) of the
    launch's units x nslices; returns per share its segments (unit, s0, s1, meeting place or -1, part)"""
    tot = units * nslices
    out = []
    for sh in range(shares):
        g, g1 = tot * sh // shares, tot * (sh + 1) // shares
        segs = []
        while g < g1:
            u = g // nslices; s0 = g - u * nslices; s1 = min(nslices, s0 + (g1 - g))
            whole = s0 == 0 and s1 == nslices
            segs.append((u, s0, s1, -1 if whole else (sh if s0 == 0 else sh - 1), 0 if s0 == 0 else 1))
            g += s1 - s0
        out.append(segs)
    return out


@pytest.mark.parametrize("units,shares", [(286, 252), (273, 252), (625, 252), (576, 252), (9, 4), (9, 7), (84, 37), (253, 252), (504, 252), (1000, 252)])
def test_stream_k_shares_cover_every_slice_once_and_cut_a_unit_in_two(units, shares):
    """launch_wino4_t deals a launch with a thin last round out as `shares` equal runs of 16-channel slices (shares = CUs - 4: a function
    of the layer and the device only).  Every (unit, slice) is computed exactly once; a share is at least a unit long, so a unit has at
    most two parts; the two parts of a cut unit name the same meeting place (the boundary's index) as part 0 (the slices before the cut,
    closing a share) and part 1 (those behind it, opening the next), and no meeting place is used by two units."""
    nslices = 8
    sh = _shares(units, nslices, shares)
    seen = np.zeros((units, nslices), np.int32)
    meet = {}
    for k, segs in enumerate(sh):
        assert sum(s1 - s0 for _, s0, s1, _, _ in segs) >= nslices                  # a share is at least a unit long ...
        for i, (u, s0, s1, m, part) in enumerate(segs):
            seen[u, s0:s1] += 1
            if m >= 0:
                assert 0 <= m < shares - 1 + 1 and part == (0 if s0 == 0 else 1)
                assert (part == 0 and i == len(segs) - 1 and m == k) or (part == 1 and i == 0 and m == k - 1)     # ... cut only at its ends
                meet.setdefault(m, []).append((u, part, s0, s1))
    assert (seen == 1).all()
    for m, parts in meet.items():
        assert len(parts) == 2 and parts[0][0] == parts[1][0], (m, parts)            # one unit per meeting place, both of its parts
        (u, p0, a0, a1), (_, p1, b0, b1) = sorted(parts, key=lambda x: x[1])
        assert (p0, p1) == (0, 1) and a0 == 0 and a1 == b0 and b1 == nslices
    cut_units = {parts[0][0] for parts in meet.values()}
    assert len(cut_units) == len(meet)                                                # no unit in three parts
