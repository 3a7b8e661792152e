"""Layers with more filters than the canonical network's (round 5; the reference's published VR checkpoints "have more filters",
README.md:141, and models_video.lua:55-140 builds any architecture string) run on the same kernels in GROUPS of output channels:
the host packs one block per group with the layer's own packing function.  No GPU: this pins the block sizes and order the kernels'
address arithmetic relies on (group g's block starts g * <block floats> into the buffer and is the packing of filters
g * G .. g * G + G - 1 alone); the kernels themselves are compared with the oracle in tests/test_gpu_parity.py."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fast-artistic-videos_amd", "csrc")


@pytest.fixture(scope="module")
def packs(tmp_path_factory):
    d = tmp_path_factory.mktemp("widepack")
    src = d / "pack.cpp"
    src.write_text('''#include "wino4_pack.h"
#include "s2_pack.h"
#include "up2_pack.h"
#include "first2d_pack.h"
#include <cstring>
static long fin(const std::vector<float>& v, float* out) { if (out) memcpy(out, v.data(), v.size() * 4); return (long)v.size(); }
extern "C" long w4(const float* w, int cin, int cout, float* out) { std::vector<float> v; fav::conv_wino4_pack(w, cin, cout, v); return fin(v, out); }
extern "C" long w4g(const float* w, int cin, int cout, float* out) { std::vector<float> v; fav::conv_wino4_pack_groups(w, cin, cout, v); return fin(v, out); }
extern "C" long s2(const float* w, int cin, int cout, float* out) { std::vector<float> v; fav::conv_s2w_pack(w, cin, cout, v); return fin(v, out); }
extern "C" long s2g(const float* w, int cin, int cout, float* out) { std::vector<float> v; fav::conv_s2w_pack_groups(w, cin, cout, v); return fin(v, out); }
extern "C" long u2(const float* w, int cin, int cout, float* out) { std::vector<float> v; fav::conv_up2w_pack(w, cin, v); return fin(v, out); }
extern "C" long u2g(const float* w, int cin, int cout, float* out) { std::vector<float> v; fav::conv_up2w_pack_groups(w, cin, cout, v); return fin(v, out); }
extern "C" long f2(const float* w, int cin, int cout, float* out) { std::vector<float> v; fav::conv_first2d_pack(w, cin, cout, v); return fin(v, out); }
extern "C" long f2g(const float* w, int cin, int cout, int coutp, float* out) { std::vector<float> v; fav::conv_first2d_pack_groups(w, cin, cout, coutp, v); return fin(v, out); }
''')
    so = d / "libwidepack.so"
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-I", CSRC, "-o", str(so), str(src)])
    lib = ctypes.CDLL(str(so))

    def call(name, w, *extra):
        f = getattr(lib, name); f.restype = ctypes.c_long
        w = np.ascontiguousarray(w, np.float32)
        cout, cin = w.shape[:2]
        args = [ctypes.c_void_p(w.ctypes.data), ctypes.c_int(cin), ctypes.c_int(cout)] + [ctypes.c_int(e) for e in extra]
        n = f(*args, ctypes.c_void_p(0))
        out = np.empty(n, np.float32)
        f(*args, ctypes.c_void_p(out.ctypes.data))
        return out
    return call


@pytest.mark.parametrize("one,grp,G,k,cin,cout", [("w4", "w4g", 128, 3, 32, 256), ("w4", "w4g", 128, 3, 48, 384), ("s2", "s2g", 128, 3, 32, 256),
                                                 ("u2", "u2g", 64, 3, 64, 128), ("u2", "u2g", 64, 3, 32, 256)])
def test_group_blocks_are_the_packings_of_their_filters(packs, one, grp, G, k, cin, cout):
    w = np.random.default_rng(cout + cin).standard_normal((cout, cin, k, k)).astype(np.float32)
    whole = packs(grp, w)
    blocks = [packs(one, w[g * G:(g + 1) * G]) for g in range(cout // G)]
    assert all(b.size == blocks[0].size for b in blocks) and whole.size == blocks[0].size * len(blocks)
    assert np.array_equal(whole, np.concatenate(blocks))
    # what the kernels assume a block to be: kernels_wino4.hip (CIN / 16 slices x 36 positions x 8192 B), kernels_s2.hip
    # (CIN / 16 chunks x 18 x NTC = 4 KiB), kernels_up2.hip (CIN / 8 groups x 18 KiB)
    want = {"w4": cin // 16 * 36 * 2048, "s2": cin // 16 * 18 * 4 * 256, "u2": cin // 8 * 18 * 256}[one]
    assert blocks[0].size == want


def test_a_single_group_is_the_plain_packing(packs):
    rng = np.random.default_rng(3)
    w = rng.standard_normal((128, 64, 3, 3)).astype(np.float32)
    assert np.array_equal(packs("w4g", w), packs("w4", w)) and np.array_equal(packs("s2g", w), packs("s2", w))
    w64 = rng.standard_normal((64, 64, 3, 3)).astype(np.float32)
    assert np.array_equal(packs("s2g", w64), packs("s2", w64)) and np.array_equal(packs("u2g", w64), packs("u2", w64))


@pytest.mark.parametrize("cin,cout,coutp", [(7, 64, 64), (7, 48, 64), (3, 96, 128)])
def test_first_layer_groups_of_32(packs, cin, cout, coutp):
    """the 2-D minimal-filtering first layer keeps ONE group's transformed weights in LDS: block g = the packing of filters 32 g .. 32 g + 31
    (fewer in the last group of a padded count: the missing filters' weights are zero)"""
    w = np.random.default_rng(cout).standard_normal((cout, cin, 9, 9)).astype(np.float32)
    whole = packs("f2g", w, coutp)
    nq = (9 * cin + 3) // 4
    assert whole.size == coutp // 32 * 16 * nq * 128
    for g in range(coutp // 32):
        sub = w[32 * g:32 * g + 32]
        blk = whole[g * 16 * nq * 128:(g + 1) * 16 * nq * 128]
        if sub.shape[0] == 0:
            assert not blk.any()
        else:
            assert np.array_equal(blk, packs("f2", sub))
    assert whole.any()


def test_fixed_point_words_of_the_accumulator_statistics():
    """kernels_wino4.hip / res_add_kernel<true> (round 5): a unit's statistics term v (a double) is added to an InstanceNorm's accumulators as
    2^-40 fixed point in TWO 64-bit words -- t = v * 2^40, hi = floor(t / 2^32), lo = trunc(t - hi * 2^32) in [0, 2^32) -- with integer atomics, and the
    consumer reassembles (sum hi * 2^32 + sum lo) * 2^-40 in double.  Restated with Python integers: the split is exact up to the truncation
    below 2^-40, the sums of the words do not depend on the order, negative terms work, and 4 000 terms of magnitude 1e10 stay far inside
    64 bits per word."""
    import math, random
    rnd = random.Random(5)
    def split(v):
        t = v * 1099511627776.0                       # exact: a power of two
        hi = math.floor(t * (1.0 / 4294967296.0))
        lo = int(t - hi * 4294967296.0)               # the subtraction is exact (both multiples of t's ulp); the conversion truncates
        return int(hi), lo
    terms = [rnd.uniform(-1e10, 1e10) for _ in range(2000)] + [rnd.uniform(-1e-3, 1e-3) for _ in range(2000)] + [0.0, -0.0, 1e-13, -1e-13, 256.0 * 300.0]
    his, los = zip(*(split(v) for v in terms))
    for v, h, l in zip(terms, his, los):
        assert 0 <= l < 2 ** 32
        exact = int(math.floor(v * 2 ** 40)) if (v * 2 ** 40) == math.floor(v * 2 ** 40) else None
        assert abs((h * 2 ** 32 + l) - v * 2 ** 40) < 1.0                 # truncation below one unit of 2^-40
        if exact is not None:
            assert h * 2 ** 32 + l == exact
    H, L = sum(his), sum(los)
    assert abs(H) < 2 ** 62 and 0 <= L < 2 ** 62
    order = list(range(len(terms))); rnd.shuffle(order)
    assert sum(his[i] for i in order) == H and sum(los[i] for i in order) == L      # integer sums: any order
    total = (float(H) * 4294967296.0 + float(L)) * (1.0 / 1099511627776.0)
    assert abs(total - math.fsum(terms)) <= len(terms) * 2.0 ** -40 + abs(math.fsum(terms)) * 2.0 ** -52
