"""Winograd F(2x2,3x3) path of the residual convolutions (csrc/kernels_wino.hip): the host-side weight transform / packing
(csrc/wino_pack.h, compiled here with g++) and a lane-level numpy restatement of the kernel's data movement -- the LDS layout of the
row-transformed halo, the fragment addresses of every lane, the 32x32x2 MFMA operand / result layout, the cross-wave output
transform -- checked against a direct 3x3 correlation.  No GPU: this pins the index arithmetic the HIP kernel is written from;
the kernel itself is compared with the oracle in tests/test_gpu_parity.py."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fast-artistic-videos_amd", "csrc")

LDSS, TROW, TTYP, TBUF = 36, 18 * 36, 4 * 18 * 36, 16 * 18 * 36


@pytest.fixture(scope="module")
def packer(tmp_path_factory):
    d = tmp_path_factory.mktemp("wino")
    src = d / "pack.cpp"
    src.write_text('#include "wino_pack.h"\n#include <cstring>\n'
                   'extern "C" long pack(const float* w, int cin, int cout, float* out) {\n'
                   '  std::vector<float> v; fav::conv_wino_pack(w, cin, cout, v); if (out) memcpy(out, v.data(), v.size() * 4); return (long)v.size(); }\n')
    so = d / "libpack.so"
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-I", CSRC, "-o", str(so), str(src)])
    lib = ctypes.CDLL(str(so))
    lib.pack.restype = ctypes.c_long
    lib.pack.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]

    def pack(w):
        cout, cin = w.shape[:2]
        w = np.ascontiguousarray(w, np.float32)
        n = lib.pack(w.ctypes.data, cin, cout, None)
        out = np.empty(n, np.float32)
        lib.pack(w.ctypes.data, cin, cout, out.ctypes.data)
        return out
    return pack


def direct_conv(x, w, b):
    """x [IH][IW][CIN], w [COUT][CIN][3][3] -> [OH][OW][COUT], correlation, no padding, float64"""
    IH, IW, _ = x.shape
    OH, OW = IH - 2, IW - 2
    y = np.zeros((OH, OW, w.shape[0]))
    for ky in range(3):
        for kx in range(3):
            y += np.einsum("hwc,oc->hwo", x[ky:ky + OH, kx:kx + OW].astype(np.float64), w[:, :, ky, kx].astype(np.float64))
    return y + b.astype(np.float64)


def emulate_unit(x, wpk, bias, scale, shift, relu, oy0, ox0, f=np.float32):
    """One work unit (8 x 16 output pixels x 128 channels) the way conv3_wino_kernel computes it.  Returns Y[8][16][128]."""
    IH, IW, CIN = x.shape
    nslices, nkg = CIN // 32, CIN // 8
    wpk = wpk.reshape(nkg, 8, 2, 4, 64, 4)
    lanes = np.arange(64)
    m, h = lanes & 31, lanes >> 5
    acc = np.zeros((8, 2, 4, 32, 32), f)                      # [wave][q][nt][tile mi][n]
    for s in range(nslices):
        # staging: items (ty, x, c4) -> four row-transformed lines in T[ty][i][pixel][36]
        T = np.full(TBUF, np.nan, f)
        for e in range(576):
            c4, pix = e & 7, e >> 3
            ty = (pix * 3641) >> 16
            xx = pix - ty * 18
            assert ty == pix // 18
            r = []
            for a in range(4):
                iy, ix = min(oy0 + 2 * ty + a, IH - 1), min(ox0 + xx, IW - 1)
                v = x[iy, ix, s * 32 + c4 * 4: s * 32 + c4 * 4 + 4].astype(f)
                if scale is not None:
                    v = v * scale[s * 32 + c4 * 4: s * 32 + c4 * 4 + 4] + shift[s * 32 + c4 * 4: s * 32 + c4 * 4 + 4]
                    if relu:
                        v = np.maximum(v, 0)
                r.append(v.astype(f))
            dst = ty * TTYP + ((xx & 1) * 9 + (xx >> 1)) * LDSS + c4 * 4
            for i, l in enumerate((r[0] - r[2], r[1] + r[2], r[2] - r[1], r[1] - r[3])):
                T[dst + i * TROW: dst + i * TROW + 4] = l
        for w in range(8):
            wodd = w & 1
            ab = (m >> 3) * TTYP + (w >> 1) * TROW + (m & 7) * LDSS + 4 * h
            ap0, ap1, ap2 = ab + (1 if wodd else 0) * LDSS, ab + (10 if wodd else 9) * LDSS, ab + (9 if wodd else 1) * LDSS
            sg = f(-1.0 if wodd else 1.0)
            for kg in range(4):
                idx = lambda base: T[(base + kg * 8)[:, None] + np.arange(4)[None, :]]     # [lane][4]
                R0, R1, R2 = idx(ap0), idx(ap1), idx(ap2)
                assert not np.isnan(R0).any() and not np.isnan(R1).any() and not np.isnan(R2).any()
                A = (R0 - R2, R1 + sg * R2)
                for q in range(2):
                    for nt in range(4):
                        B = wpk[s * 4 + kg, w, q, nt]                                        # [lane][4]
                        for st in range(4):
                            # v_mfma_f32_32x32x2_f32: D[i][j] += sum_k A[i][k] B[k][j]; lane l holds A[l & 31][l >> 5] and B[l >> 5][l & 31]
                            a2 = A[q][:, st].reshape(2, 32)        # [k][i]
                            b2 = B[:, st].reshape(2, 32)           # [k][j]
                            acc[w, q, nt] += (a2.T.astype(np.float64) @ b2.astype(np.float64)).astype(f)
    # output transform: column fold per wave, row fold across waves through Ps[wave][cout][tile]
    Y = np.zeros((8, 16, 128), f)
    for b in range(2):
        Ps = np.zeros((8, 128, LDSS), f)
        for w in range(8):
            m0, m1 = acc[w, 0], acc[w, 1]                                                    # [nt][mi][n]
            P = (m0 if b == 0 else -(m0 + m1)) if (w & 1) else (m0 + m1 if b == 0 else m1)
            for nt in range(4):
                for lane in range(64):
                    n, hh = lane & 31, lane >> 5
                    for g in range(4):
                        for e in range(4):
                            rr = 4 * g + e
                            mi = (rr & 3) + 8 * (rr >> 2) + 4 * hh                           # accumulator register -> tile (MFMA D layout)
                            Ps[w, nt * 32 + n, 8 * g + 4 * hh + e] = P[nt, mi, n]
        for t in range(512):
            c, qq = t & 127, t >> 7
            z = [Ps[2 * i, c, 8 * qq: 8 * qq + 8] + Ps[2 * i + 1, c, 8 * qq: 8 * qq + 8] for i in range(4)]
            y0 = (z[0] + z[1]) + z[2] + bias[c]
            y1 = (z[1] - z[2]) - z[3] + bias[c]
            for k in range(8):
                Y[2 * qq + 0, 2 * k + b, c] = y0[k]
                Y[2 * qq + 1, 2 * k + b, c] = y1[k]
    return Y


@pytest.mark.parametrize("cin,aff", [(32, False), (64, True)])
def test_lane_level_restatement_matches_direct_convolution(packer, cin, aff):
    rng = np.random.default_rng(5 + cin)
    IH, IW, COUT = 13, 21, 128                         # 11 x 19 outputs: 2 x 2 units, ragged in both directions
    x = rng.standard_normal((IH, IW, cin)).astype(np.float32)
    w = (rng.standard_normal((COUT, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, COUT).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cin).astype(np.float32) if aff else None
    shift = rng.uniform(-0.5, 0.5, cin).astype(np.float32) if aff else None
    xin = np.maximum(x * scale + shift, 0) if aff else x
    ref = direct_conv(xin, w, b)
    wpk = packer(w)
    assert wpk.size == cin * 2048
    OH, OW = IH - 2, IW - 2
    out = np.zeros((OH, OW, COUT), np.float32)
    for uy in range((OH + 7) // 8):
        for ux in range((OW + 15) // 16):
            Y = emulate_unit(x, wpk, b, scale, shift, aff, uy * 8, ux * 16)
            hh, ww = min(8, OH - uy * 8), min(16, OW - ux * 16)
            out[uy * 8: uy * 8 + hh, ux * 16: ux * 16 + ww] = Y[:hh, :ww]
    err = np.abs(out - ref).max()
    assert err < 2e-5, err


def test_pending_join_is_written_exactly_once_per_input_pixel():
    """conv3_wino_kernel MODE 2 (the residual join formed while the halo is staged): every input pixel of the convolution is written to the
    joined tensor by exactly one staging item -- rows 2 ty, 2 ty + 1 of columns 0..15 of each unit plus the halo fringe of the last units"""
    for OH, OW in ((11, 19), (16, 32), (8, 16), (23, 50), (180, 320)):
        IH, IW = OH + 2, OW + 2
        units_y, units_x = (OH + 7) // 8, (OW + 15) // 16
        hits = np.zeros((IH, IW), np.int32)
        for uy in range(units_y):
            for ux in range(units_x):
                oy0, ox0 = uy * 8, ux * 16
                lastx, lasty = ux == units_x - 1, uy == units_y - 1
                items = [(t >> 3) for t in range(512) if (t & 7) == 0] + [64 + (l >> 3) for l in range(64) if (l & 7) == 0]      # one channel chunk per pixel
                for pix in items:
                    ty, x = pix // 18, pix % 18
                    assert (ty, x) == ((3, 10 + pix - 64) if pix >= 64 else (pix // 18, pix % 18))
                    col_ok = ox0 + x < IW and (x < 16 or lastx)
                    for a in range(4):
                        row_ok = oy0 + 2 * ty + a < IH and (a < 2 or (ty == 3 and lasty))
                        if col_ok and row_ok:
                            hits[oy0 + 2 * ty + a, ox0 + x] += 1
        assert (hits == 1).all(), (OH, OW, np.argwhere(hits != 1)[:5])


def test_fragment_reads_are_bank_conflict_free():
    """ds_read_b128 is served in groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, + 32 for the upper half-wave); within a
    group the 16-byte slots (address / 16 mod 16) must all differ (MI355X LDS: 64 banks x 4 B)."""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    lanes = np.arange(64)
    m, h = lanes & 31, lanes >> 5
    for w in range(8):
        ab = (m >> 3) * TTYP + (w >> 1) * TROW + (m & 7) * LDSS + 4 * h
        for off in (0, 1, 9, 10):
            for kg in range(4):
                addr = (ab + off * LDSS + kg * 8) * 4
                assert (addr % 16 == 0).all()
                for g in groups:
                    assert len(set((addr[g] // 16) % 16)) == 16
    # epilogue exchange: reads by (c = t & 127) at pitch 36 floats, writes by n = lane & 31 in groups of 8 consecutive lanes
    for t0 in range(0, 512, 64):
        t = t0 + lanes
        addr = ((t & 127) * LDSS + 8 * (t >> 7)) * 4
        for g in groups:
            assert len(set((addr[g] // 16) % 16)) == 16
    addr = ((lanes & 31) * LDSS + 4 * h) * 4
    for g0 in range(0, 64, 8):
        assert len(set((addr[g0:g0 + 8] // 16) % 8)) == 8


# ---------------------------------------------------------------------------------------------- first layer: F(2,3) along x
@pytest.fixture(scope="module")
def first_packer(tmp_path_factory):
    d = tmp_path_factory.mktemp("first")
    src = d / "pack.cpp"
    src.write_text('#include "first_pack.h"\n#include <cstring>\n'
                   'extern "C" long pack(const float* w, int cin, int cout, float* out) {\n'
                   '  std::vector<float> v; fav::conv_first_pack(w, cin, cout, v); if (out) memcpy(out, v.data(), v.size() * 4); return (long)v.size(); }\n'
                   'extern "C" void combo(int cr, int j, int h, int* c) { fav::conv_first_combo(cr, j, h, c, c + 1, c + 2); }\n'
                   'extern "C" int pairs(int cr) { return fav::conv_first_pairs(cr); }\n')
    so = d / "libfirst.so"
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-I", CSRC, "-o", str(so), str(src)])
    lib = ctypes.CDLL(str(so))
    lib.pack.restype = ctypes.c_long
    lib.pack.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return lib


@pytest.mark.parametrize("cin", [7, 3])
def test_first_layer_pairing_and_transform_match_direct_convolution(first_packer, cin):
    """csrc/first_pack.h: every (channel, filter row, block) appears in exactly one pair half, and the four GEMMs over the packed
    weights with the operands the kernel builds (d0-d2, d1+d2, d2-d1, d1-d3 of halo columns 2t+3b+j) give the 9x9 correlation."""
    lib = first_packer
    NJ = lib.pairs(cin)
    assert NJ == (cin // 2) * 27 + 14
    seen = {}
    combos = np.zeros((NJ, 2, 3), np.int32)
    for j in range(NJ):
        for h in range(2):
            c = (ctypes.c_int * 3)()
            lib.combo(cin, j, h, c)
            combos[j, h] = list(c)
            if c[0] >= 0:
                assert (c[0], c[1], c[2]) not in seen
                seen[(c[0], c[1], c[2])] = (j, h)
    assert len(seen) == cin * 27
    rng = np.random.default_rng(cin)
    cout, H, W = 32, 3, 14                                  # 3 output rows, 14 output columns = 7 tiles
    w = (rng.standard_normal((cout, cin, 9, 9)) / 20).astype(np.float32)
    x = rng.uniform(-120, 150, (cin, H + 8, W + 8)).astype(np.float32)          # the halo (pad already applied)
    n = lib.pack(w.ctypes.data, cin, cout, None)
    wpk = np.empty(n, np.float32)
    lib.pack(w.ctypes.data, cin, cout, wpk.ctypes.data)
    wpk = wpk.reshape(4, NJ, 2, 32)
    ref = np.zeros((H, W, cout))
    for ky in range(9):
        for kx in range(9):
            ref += np.einsum("chw,oc->hwo", x[:, ky:ky + H, kx:kx + W].astype(np.float64), w[:, :, ky, kx].astype(np.float64))
    out = np.zeros((H, W, cout), np.float32)
    for oy in range(H):
        for t in range(W // 2):
            M = np.zeros((4, cout), np.float32)
            for j in range(NJ):
                for h in range(2):
                    c, ky, b = combos[j, h]
                    if c < 0:
                        continue
                    d = x[c, oy + ky, 2 * t + 3 * b: 2 * t + 3 * b + 4]
                    V = np.array([d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]], np.float32)
                    M += V[:, None] * wpk[:, j, h, :]
            out[oy, 2 * t] = (M[0] + M[1]) + M[2]
            out[oy, 2 * t + 1] = (M[1] - M[2]) - M[3]
    err = np.abs(out - ref).max()
    assert err < 2e-3 * max(1.0, np.abs(ref).max() / 100), err


# ---------------------------------------------------------------------------------------------- first layer: F(2x2,3x3) over the nine 3x3 blocks
@pytest.fixture(scope="module")
def first2d_packer(tmp_path_factory):
    d = tmp_path_factory.mktemp("first2d")
    src = d / "pack.cpp"
    src.write_text('#include "first2d_pack.h"\n#include <cstring>\n'
                   'extern "C" long pack(const float* w, int cin, int cout, float* out) {\n'
                   '  std::vector<float> v; fav::conv_first2d_pack(w, cin, cout, v); if (out) memcpy(out, v.data(), v.size() * 4); return (long)v.size(); }\n'
                   'extern "C" void combo(int cr, int q, int g, int* c) { fav::conv_first2d_combo(cr, q, g, c, c + 1, c + 2); }\n'
                   'extern "C" int quads(int cr) { return fav::conv_first2d_quads(cr); }\n')
    so = d / "libfirst2d.so"
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-I", CSRC, "-o", str(so), str(src)])
    lib = ctypes.CDLL(str(so))
    lib.pack.restype = ctypes.c_long
    lib.pack.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return lib


@pytest.mark.parametrize("cin", [7, 3])
def test_first_layer_2d_lane_level_restatement_matches_direct_convolution(first2d_packer, cin):
    """csrc/first2d_pack.h + conv_first2d_kernel: every (channel, a, b) sits in exactly one (quad, lane group); one wave = 16 tiles of
    2x2 outputs, lane = (tile column, lane group), v_mfma_f32_16x16x4_f32 operand / result layout (A[i = l % 16][k = l / 16],
    B[k = l / 16][j = l % 16], D register r of lane l = D[4 (l / 16) + r][l % 16]), patch addresses in the plain-row halo planes, the
    output transform inside the lane -- against the 9x9 correlation."""
    lib = first2d_packer
    NQ = lib.quads(cin)
    assert NQ == (9 * cin + 3) // 4
    seen = set()
    combos = np.zeros((NQ, 4, 3), np.int32)
    for q in range(NQ):
        for g in range(4):
            c = (ctypes.c_int * 3)()
            lib.combo(cin, q, g, c)
            combos[q, g] = list(c)
            if c[0] >= 0:
                assert tuple(c) not in seen
                seen.add(tuple(c))
    assert len(seen) == 9 * cin
    rng = np.random.default_rng(40 + cin)
    cout, HR, HC = 32, 24, 40                              # one block tile: 16 x 32 outputs from a 24 x 40 halo
    CPL = HR * HC + 16
    w = (rng.standard_normal((cout, cin, 9, 9)) / 20).astype(np.float32)
    halo = rng.uniform(-120, 150, (cin, HR, HC)).astype(np.float32)
    n = lib.pack(w.ctypes.data, cin, cout, None)
    wpk = np.empty(n, np.float32)
    lib.pack(w.ctypes.data, cin, cout, wpk.ctypes.data)
    assert n == 16 * NQ * 128
    Hs = np.full(cin * CPL, np.nan, np.float32)
    for c in range(cin):
        Hs[c * CPL: c * CPL + HR * HC] = halo[c].ravel()
    ref = np.zeros((16, 32, cout))
    for ky in range(9):
        for kx in range(9):
            ref += np.einsum("chw,oc->hwo", halo[:, ky:ky + 16, kx:kx + 32].astype(np.float64), w[:, :, ky, kx].astype(np.float64))
    lanes = np.arange(64)
    txl, g = lanes & 15, lanes >> 4
    out = np.zeros((16, 32, cout), np.float32)
    for wave in range(8):
        acc = np.zeros((16, 2, 64, 4), np.float32)            # [position][nt][lane][register]
        a_lane = 2 * wave * HC + 2 * txl
        for q in range(NQ):
            idx = 4 * q + g
            c = idx // 9; a = (idx - 9 * c) // 3; b = idx - 9 * c - 3 * a
            qoff = np.where(idx < 9 * cin, c * CPL + 3 * a * HC + 3 * b, 0)
            assert all((combos[q, gg] == [c[gg * 16], a[gg * 16], b[gg * 16]]).all() for gg in range(4) if idx[gg * 16] < 9 * cin)
            R = np.stack([[Hs[a_lane + qoff + r * HC + d] for d in range(4)] for r in range(4)])      # [r][d][lane]
            assert not np.isnan(R).any()
            for i in range(4):
                L = (R[0] - R[2], R[1] + R[2], R[2] - R[1], R[1] - R[3])[i]                             # [d][lane]
                V = (L[0] - L[2], L[1] + L[2], L[2] - L[1], L[1] - L[3])
                for j in range(4):
                    for nt in range(2):
                        Bv = wpk[((((4 * i + j) * NQ + q) * 2 + nt) * 64) + lanes]
                        A2 = V[j].reshape(4, 16)                # [k][i = tile]
                        B2 = Bv.reshape(4, 16)                  # [k][j = channel]
                        D = (A2.T.astype(np.float64) @ B2.astype(np.float64)).astype(np.float32)       # [tile][channel]
                        for l in range(64):
                            for r in range(4):
                                acc[4 * i + j, nt, l, r] += D[4 * (l >> 4) + r, l & 15]
        for l in range(64):
            for nt in range(2):
                ch = nt * 16 + (l & 15)
                for r in range(4):
                    M = acc[:, nt, l, r].reshape(4, 4)
                    Q = np.stack([(M[:, 0] + M[:, 1]) + M[:, 2], (M[:, 1] - M[:, 2]) - M[:, 3]], axis=1)   # [i][b]
                    for bcol in range(2):
                        out[2 * wave + 0, 2 * (4 * (l >> 4) + r) + bcol, ch] = (Q[0, bcol] + Q[1, bcol]) + Q[2, bcol]
                        out[2 * wave + 1, 2 * (4 * (l >> 4) + r) + bcol, ch] = (Q[1, bcol] - Q[2, bcol]) - Q[3, bcol]
    err = np.abs(out - ref).max()
    assert err < 3e-3 * max(1.0, np.abs(ref).max() / 100), err


# ---------------------------------------------------------------------------------------------- U2 + c3s1-64: nine-position form
UW_HW, UW_HP, UW_HPP = 34, 6 * 34, 288
UW_HB = UW_HPP * LDSS


@pytest.fixture(scope="module")
def up2w_packer(tmp_path_factory):
    d = tmp_path_factory.mktemp("up2w")
    src = d / "pack.cpp"
    src.write_text('#include "up2_pack.h"\n#include <cstring>\n'
                   'extern "C" long pack(const float* w, int cin, float* out) {\n'
                   '  std::vector<float> v; fav::conv_up2w_pack(w, cin, v); if (out) memcpy(out, v.data(), v.size() * 4); return (long)v.size(); }\n')
    so = d / "libup2w.so"
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-I", CSRC, "-o", str(so), str(src)])
    lib = ctypes.CDLL(str(so))
    lib.pack.restype = ctypes.c_long
    lib.pack.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]

    def pack(w):
        w = np.ascontiguousarray(w, np.float32)
        n = lib.pack(w.ctypes.data, w.shape[1], None)
        out = np.empty(n, np.float32)
        lib.pack(w.ctypes.data, w.shape[1], out.ctypes.data)
        return out
    return pack


def emulate_up2w_tile(x, wpk_bytes, bias, scale, shift, relu, sy0, sx0, f=np.float32):
    """One tile (4 x 32 physical pixels -> 8 x 64 outputs x 64 channels) the way conv3_up2w_kernel computes it: byte offsets into the
    packed weights, the staging threads' halo pieces, the fragment addresses of the 12 waves, the 32x32x2 MFMA operand / result
    layout and the LDS exchange of the output transform are the kernel's formulas.  Returns Y[8][64][64] (rows/columns past the
    image are left NaN) and the per-channel (mean, M2, count) the tile publishes."""
    PH, PW, CIN = x.shape
    nslices, nkg = CIN // 32, CIN // 8
    lanes = np.arange(64)
    m, h, n = lanes & 31, lanes >> 5, lanes & 31
    wfl = wpk_bytes.view(np.float32)
    acc = np.zeros((12, 3, 2, 16, 64), np.float64)              # [wave][jj][nt][r][lane]
    for s in range(nslices):
        Hs = np.full(UW_HB, np.nan, f)
        for t in range(768):                                    # staging: thread t, pieces i = 0..2
            c4 = t & 7
            for i in range(3):
                pix = (t >> 3) + 96 * i
                hy = (pix * 1928) >> 16
                assert hy == pix // 34
                hx = pix - hy * UW_HW
                sy, sx = sy0 - 1 + hy, sx0 - 1 + hx
                v = pix < UW_HP and 0 <= sy < PH and 0 <= sx < PW
                q = np.zeros(4, f)
                if v:
                    ch = s * 32 + c4 * 4
                    q = x[sy, sx, ch:ch + 4].astype(f)
                    if scale is not None:
                        q = (q * scale[ch:ch + 4] + shift[ch:ch + 4]).astype(f)
                        if relu:
                            q = np.maximum(q, 0)
                dst = ((t >> 3) + i * 96) * LDSS + c4 * 4
                Hs[dst:dst + 4] = q
        for wave in range(12):
            kind, row = wave >> 2, wave & 3
            kap = f(0.0 if kind == 1 else 1.0)
            ra = ((row + (0 if kind == 0 else 1)) * UW_HW + m) * LDSS + 4 * h
            rb = ((row + (1 if kind <= 1 else 2)) * UW_HW + m) * LDSS + 4 * h
            wlo, wso = lanes * 16, kind * 3 * 2048
            for kg in range(4):
                kgg = s * 4 + kg
                rd = lambda base, dx: Hs[(base + dx * LDSS + kg * 8)[:, None] + np.arange(4)[None, :]]      # [lane][4]
                L = []
                for dx in range(3):
                    Ra, Rb = rd(ra, dx), rd(rb, dx)
                    assert not np.isnan(Ra).any() and not np.isnan(Rb).any()
                    L.append((Ra - kap * Rb).astype(f))
                V = (L[0] - L[1], L[1], L[1] - L[2])
                for jj in range(3):
                    for nt in range(2):
                        off = wlo + nt * 1024 + wso + kgg * 18432 + jj * 2048
                        assert (off + 16 <= nkg * 18 * 1024).all()
                        B = wfl[(off // 4)[:, None] + np.arange(4)[None, :]]                                  # [lane][4]
                        for st in range(4):
                            a2 = V[jj][:, st].reshape(2, 32).astype(np.float64)       # [k][i]: lane l holds A[l & 31][l >> 5]
                            b2 = B[:, st].reshape(2, 32).astype(np.float64)           # [k][j]: lane l holds B[l >> 5][l & 31]
                            D = a2.T @ b2                                             # [i][j]
                            for r in range(16):
                                i_ = (r & 3) + 8 * (r >> 2) + 4 * h                   # D layout: register r of lane l is D[i_][l & 31]
                                acc[wave, jj, nt, r] += D[i_, n]
    acc = acc.astype(f)
    # output transform through the exchange area
    Z = np.full(8 * 64 * 2 * LDSS, np.nan, f)
    for wave in range(12):
        kind, row = wave >> 2, wave & 3
        if kind == 1:
            continue
        zw = (((4 if kind else 0) + row) * 64 + n) * 2 * LDSS + 4 * h
        for nt in range(2):
            for g in range(4):
                for e in range(4):
                    r = 4 * g + e
                    Z[zw + nt * 32 * 2 * LDSS + 8 * g + e] = acc[wave, 0, nt, r] + acc[wave, 1, nt, r]
                    Z[zw + nt * 32 * 2 * LDSS + LDSS + 8 * g + e] = acc[wave, 1, nt, r] - acc[wave, 2, nt, r]
    Y = np.full((8, 64, 64), np.nan, f)
    stt = np.zeros((4, 64, 2), f)
    wn = np.zeros(4, np.int64)
    for row in range(4):
        wave = 4 + row
        z0r = ((0 + row) * 64 + n) * 2 * LDSS + 4 * h
        z3r = ((4 + row) * 64 + n) * 2 * LDSS + 4 * h
        sy = sy0 + row
        vals = [[[] for _ in range(64)] for _ in range(2)]
        for nt in range(2):
            co = nt * 32 + n
            for g in range(4):
                for e in range(4):
                    r = 4 * g + e
                    a0, a1 = Z[z0r + nt * 32 * 2 * LDSS + 8 * g + e], Z[z0r + nt * 32 * 2 * LDSS + LDSS + 8 * g + e]
                    c0, c1 = Z[z3r + nt * 32 * 2 * LDSS + 8 * g + e], Z[z3r + nt * 32 * 2 * LDSS + LDSS + 8 * g + e]
                    assert not (np.isnan(a0).any() or np.isnan(a1).any() or np.isnan(c0).any() or np.isnan(c1).any())
                    b0, b1 = acc[wave, 0, nt, r] + acc[wave, 1, nt, r], acc[wave, 1, nt, r] - acc[wave, 2, nt, r]
                    bv = bias[co]
                    y = (a0 + b0 + bv, a1 + b1 + bv, b0 - c0 + bv, b1 - c1 + bv)
                    sx = sx0 + (r & 3) + 8 * (r >> 2) + 4 * h
                    for l in range(64):
                        if sy < PH and sx[l] < PW:
                            my, mx = row, sx[l] - sx0
                            Y[2 * my, 2 * mx, co[l]], Y[2 * my, 2 * mx + 1, co[l]] = y[0][l], y[1][l]
                            Y[2 * my + 1, 2 * mx, co[l]], Y[2 * my + 1, 2 * mx + 1, co[l]] = y[2][l], y[3][l]
                            vals[nt][l] += [y[0][l], y[1][l], y[2][l], y[3][l]]
        for nt in range(2):
            for l in range(32):
                v = np.array(vals[nt][l] + vals[nt][l + 32], np.float64)
                mu = v.mean() if v.size else 0.0
                stt[row, nt * 32 + l] = (mu, ((v - mu) ** 2).sum())
        wn[row] = len(vals[0][0]) + len(vals[0][32])
    cnt = int(wn.sum())
    mean = (wn[:, None] * stt[:, :, 0]).sum(0) / max(cnt, 1)
    m2 = (stt[:, :, 1] + wn[:, None] * (stt[:, :, 0] - mean) ** 2).sum(0)
    return Y, mean, m2, cnt


@pytest.mark.parametrize("cin,relu", [(32, True), (64, False)])
def test_up2w_lane_level_restatement_matches_direct_convolution(up2w_packer, cin, relu):
    """conv3_up2w_kernel (csrc/kernels_up2.hip) + conv_up2w_pack (csrc/up2_pack.h): nearest x2 upsampling followed by the zero-padded
    3x3 convolution (models_video.lua:123-128), on the physical pixels with nine transform positions."""
    rng = np.random.default_rng(11 + cin)
    PH, PW, COUT = 6, 37, 64                                   # 2 x 2 tiles, ragged in both directions
    x = rng.standard_normal((PH, PW, cin)).astype(np.float32)
    w = (rng.standard_normal((COUT, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, COUT).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cin).astype(np.float32)
    shift = rng.uniform(-0.5, 0.5, cin).astype(np.float32)
    xin = x * scale + shift
    if relu:
        xin = np.maximum(xin, 0)
    up = np.repeat(np.repeat(xin, 2, axis=0), 2, axis=1)
    ref = direct_conv(np.pad(up, ((1, 1), (1, 1), (0, 0))), w, b)          # [2 PH][2 PW][64]
    wpk = up2w_packer(w)
    assert wpk.size == (cin // 8) * 9 * 2 * 64 * 4
    out = np.full((2 * PH, 2 * PW, COUT), np.nan, np.float32)
    for ty in range((PH + 3) // 4):
        for tx in range((PW + 31) // 32):
            Y, mean, m2, cnt = emulate_up2w_tile(x, wpk.view(np.uint8), b, scale, shift, relu, ty * 4, tx * 32)
            hh, ww = min(8, 2 * PH - ty * 8), min(64, 2 * PW - tx * 64)
            assert np.isnan(Y[hh:]).all() and np.isnan(Y[:, ww:]).all()
            out[ty * 8: ty * 8 + hh, tx * 64: tx * 64 + ww] = Y[:hh, :ww]
            blk = ref[ty * 8: ty * 8 + hh, tx * 64: tx * 64 + ww].reshape(-1, COUT)
            assert cnt == blk.shape[0]
            assert np.abs(mean - blk.mean(0)).max() < 1e-5 and np.abs(m2 - ((blk - blk.mean(0)) ** 2).sum(0)).max() < 1e-3 * max(1.0, m2.max())
    err = np.abs(out - ref).max()
    assert err < 2e-5, err


def test_up2w_fragment_reads_are_bank_conflict_free():
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    lanes = np.arange(64)
    m, h = lanes & 31, lanes >> 5
    for hrow in range(6):
        for dx in range(3):
            for kg in range(4):
                addr = (((hrow * UW_HW + m) * LDSS + 4 * h) + dx * LDSS + kg * 8) * 4
                assert (addr % 16 == 0).all()
                for g in groups:
                    assert len(set((addr[g] // 16) % 16)) == 16
