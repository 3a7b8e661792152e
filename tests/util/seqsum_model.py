"""Numpy restatement of the MULTI-BLOCK sequential fp32 sum of csrc/kernels_consistency.hip (avg_chunk_sum_kernel,
avg_chunk_class_kernel, avg_chunk_scan_kernel + the one-block finisher): CMatrix::avg's running sum (consistencyChecker's
CMatrix.h:1245-1251) evaluated chunk by chunk from predicted binades.  TEST INFRASTRUCTURE: it pins the SCHEME on the CPU -- the
exponent prediction with its 3 % margins, the per-chunk parity transducers, the conservative binade-crossing rule, the native
chunks and the hand-over to the finisher -- against the plain sequential loop; the GPU suite (test_sequential_sum_bit_exact) then
pins the kernels themselves."""
import numpy as np
XCAP = 1 << 28
ACH = 256
def sat(a): return np.minimum(a, XCAP)
def elem_class(x, e):
    """vectorised elem_class: x float32 array, e int array (same shape) -> f, g, tie"""
    b = x.view(np.uint32).astype(np.int64)
    ex = (b >> 23) & 255
    m = b & 0x7FFFFF
    m = np.where(ex != 0, m | 0x800000, m); ex = np.where(ex != 0, ex, 1)
    k = e - ex
    bad = ((b >> 31) != 0) | (ex == 255) | ((k < 0) & (m != 0))
    ks = np.clip(k, 0, 25)
    r = m & ((1 << ks) - 1)
    half = np.where(ks > 0, 1 << np.maximum(ks - 1, 0), 0xFFFFFFFF)
    f = np.where(bad, XCAP, m >> ks)
    g = (r > half).astype(np.int64); tie = ((r == half) & (k <= 24)).astype(np.int64)
    g = np.where(k > 24, 0, g)
    return f, g, tie
def seq_sum(x): return np.add.accumulate(x.astype(np.float32), dtype=np.float32)[-1]
def f32_exp(p):
    f = np.float32(p)
    e = (f.view(np.uint32) >> 23) & 255
    return int(e) if (f > 0 and 1 <= e <= 254) else 0
def model(x):
    n = len(x); nc = (n + ACH - 1) // ACH
    xp = np.zeros(nc * ACH, np.float32); xp[:n] = x
    X = xp.reshape(nc, ACH)
    b = X.view(np.uint32)
    bad = (((b >> 31) != 0) | (((b >> 23) & 255) == 255)).any(axis=1)
    T = X.astype(np.float64).sum(axis=1)
    P = np.concatenate([[0.0], np.cumsum(T)[:-1]])
    e_lo = np.array([f32_exp(p * 0.96875) for p in P]); e_hi = np.array([f32_exp((p + t) * 1.03125) for p, t in zip(P, T)])
    nslots = np.where((e_lo == 0) | bad, 0, np.where(e_hi > e_lo, 2, 1))
    D0 = np.zeros((nc, 2), np.int64); D1 = np.zeros((nc, 2), np.int64); PP = np.full((nc, 2), 2, np.int64)
    for sl in range(2):
        e = (e_lo + sl)[:, None] * np.ones((1, ACH), np.int64)
        f, g, tie = elem_class(X, e)
        d0 = np.zeros(nc, np.int64); d1 = np.zeros(nc, np.int64); p0 = np.zeros(nc, np.int64); p1 = np.ones(nc, np.int64)
        for j in range(ACH):
            fo = f[:, j] & 1
            dl0 = f[:, j] + g[:, j] + (tie[:, j] & (p0 ^ fo)); dl1 = f[:, j] + g[:, j] + (tie[:, j] & (p1 ^ fo))
            d0 = sat(d0 + dl0); p0 = (p0 + dl0) & 1
            d1 = sat(d1 + dl1); p1 = (p1 + dl1) & 1
        D0[:, sl] = d0; D1[:, sl] = d1; PP[:, sl] = p0 | (p1 << 1)
    # level 2
    s = np.float32(0); ci = 0; miss = 0; rounds = 0; natives = 0; halted = False
    while ci < nc:
        rounds += 1
        bits = int(np.float32(s).view(np.uint32)); e = (bits >> 23) & 255
        ok = s > 0 and 1 <= e <= 254
        M = (bits & 0x7FFFFF) | 0x800000; limit = (1 << 24) - M
        cnt = min(1024, nc - ci)
        par = M & 1; D = 0; first = cnt
        for t in range(cnt):
            c = ci + t; sl = e - e_lo[c]
            usable = ok and not bad[c] and 0 <= sl < 2 and sl < nslots[c]
            if usable:
                d = D1[c, sl] if par else D0[c, sl]
            if (not usable) or min(min(D + d, XCAP) + 1, XCAP) >= limit:
                first = t; break
            D = min(D + d, XCAP); par = (PP[c, sl] >> par) & 1
        sn = np.float32(np.ldexp(np.float32(M + D), e - 150)) if (ok and D > 0) else s
        if first < cnt:
            c = ci + first; sl = e - e_lo[c]
            usable = ok and not bad[c] and 0 <= sl < 2 and sl < nslots[c]
            miss = miss + 1 if (not usable and not bad[c]) else 0
            if bad[c] or miss > 8:
                halted = True; s = sn; ci = c; break
            natives += 1
            for v in X[c][: min(ACH, n - c * ACH)]: sn = np.float32(sn + v)
            ci = c + 1
        else:
            ci += cnt
        s = sn
    i = min(ci * ACH, n)
    # finisher: plain chain from (i, s)
    for v in x[i:]: s = np.float32(s + v)
    return s, dict(rounds=rounds, natives=natives, halted_at=(i if halted else None))
