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
