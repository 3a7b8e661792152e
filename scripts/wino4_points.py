#!/usr/bin/env python3
"""fp32 error of one 3x3 128 -> 128 layer computed by Winograd minimal filtering, against the same layer in fp64 -- the comparison behind the
choice of interpolation points in csrc/wino4_pack.h (0, +-3/4, +-3/2, inf instead of the textbook 0, +-1, +-2, inf; Lavin & Gray 2016).
Transforms in fp32 (matrix products in the order B^T (d B), G g G^T in fp64 rounded once, A^T (M A)), channel sum in fp32 in channel order.
CPU only:  python scripts/wino4_points.py > profiles/r04_wino4_points.txt"""
import numpy as np

f32 = np.float32
def mats(name):
    if name == "F(4x4,3x3) points 0, +-3/4, +-3/2, inf (wino4_pack.h)":
        BT = [[81/64, 0, -45/16, 0, 1, 0], [0, -27/16, -9/4, 3/4, 1, 0], [0, 27/16, -9/4, -3/4, 1, 0],
              [0, -27/32, -9/16, 3/2, 1, 0], [0, 27/32, -9/16, -3/2, 1, 0], [0, 81/64, 0, -45/16, 0, 1]]
        G = [[64/81, 0, 0], [-128/243, -32/81, -8/27], [-128/243, 32/81, -8/27], [32/243, 16/81, 8/27], [32/243, -16/81, 8/27], [0, 0, 1]]
        AT = [[1, 1, 1, 1, 1, 0], [0, 3/4, -3/4, 3/2, -3/2, 0], [0, 9/16, 9/16, 9/4, 9/4, 0], [0, 27/64, -27/64, 27/8, -27/8, 1]]
    elif name == "F(4x4,3x3) points 0, +-1, +-2, inf (textbook)":
        BT = [[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]]
        G = [[1/4, 0, 0], [-1/6, -1/6, -1/6], [-1/6, 1/6, -1/6], [1/24, 1/12, 1/6], [1/24, -1/12, 1/6], [0, 0, 1]]
        AT = [[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]
    else:
        BT = [[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]]
        G = [[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]]
        AT = [[1, 1, 1, 0], [0, 1, -1, -1]]
    return np.array(BT, np.float64), np.array(G, np.float64), np.array(AT, np.float64)

def wino(x, w, name):
    BT, G, AT = mats(name)
    n, m = BT.shape[0], AT.shape[0]                 # patch size, outputs per tile side
    C, H, W = x.shape; K = w.shape[0]
    th, tw = (H - 2) // m, (W - 2) // m
    U = np.einsum("ia,kcab,jb->kcij", G, w.astype(np.float64), G).astype(f32)          # fp64, rounded once (as the pack does)
    BTf, ATf = BT.astype(f32), AT.astype(f32)
    out = np.zeros((K, th * m, tw * m), f32)
    for ty in range(th):
        for tx in range(tw):
            d = x[:, ty * m:ty * m + n, tx * m:tx * m + n]
            V = np.einsum("ia,cab->cib", BTf, d).astype(f32)
            V = np.einsum("cib,jb->cij", V, BTf).astype(f32)
            M = np.zeros((K, n, n), f32)
            for c in range(C):                       # fp32 accumulation in channel order
                M = (M + U[:, c] * V[c][None]).astype(f32)
            Y = np.einsum("ia,kab->kib", ATf, M).astype(f32)
            Y = np.einsum("kib,jb->kij", Y, ATf).astype(f32)
            out[:, ty * m:(ty + 1) * m, tx * m:(tx + 1) * m] = Y
    return out

def direct(x, w, dt):
    C, H, W = x.shape; K = w.shape[0]
    out = np.zeros((K, H - 2, W - 2), dt)
    for c in range(C):                               # channel-major accumulation, taps inside
        for a in range(3):
            for b in range(3):
                out = (out + w[:, c, a, b].astype(dt)[:, None, None] * x[c, a:a + H - 2, b:b + W - 2].astype(dt)[None]).astype(dt)
    return out

rng = np.random.default_rng(4)
C = K = 128
x = np.maximum(rng.standard_normal((C, 26, 26)), 0).astype(f32)          # what the residual layers see: normalised, rectified
w = (rng.standard_normal((K, C, 3, 3)) * 0.03).astype(f32)
ref = direct(x, w, np.float64)
scale = np.sqrt((ref ** 2).mean())
print("one 3x3 128 -> 128 layer, 24x24 outputs, input max(0, N(0,1)), weights N(0, 0.03^2); errors relative to the output's rms (%.3f)" % scale)
for name in ["F(4x4,3x3) points 0, +-3/4, +-3/2, inf (wino4_pack.h)", "F(4x4,3x3) points 0, +-1, +-2, inf (textbook)", "F(2x2,3x3) (wino_pack.h)"]:
    y = wino(x, w, name)
    e = (y.astype(np.float64) - ref[:, :y.shape[1], :y.shape[2]]) / scale
    print("%-62s rms %.2e   max %.2e" % (name, np.sqrt((e ** 2).mean()), np.abs(e).max()))
e = (direct(x, w, f32).astype(np.float64) - ref) / scale
print("%-62s rms %.2e   max %.2e" % ("direct form, fp32 accumulation chain (channel-major)", np.sqrt((e ** 2).mean()), np.abs(e).max()))
