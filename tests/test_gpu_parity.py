"""GPU parity suite (-m gpu, runs on a real MI355X): every libfav entry point is called through the
C ABI and compared with the CPU oracle on the same seeded inputs; the mask additionally against the
committed outputs of the reference's own consistencyChecker.

Tolerances (fp32 path, stated once):
  * occlusion mask: bit-exact (integer output);
  * warp / min-filter / assembly: 1e-5 relative to the operand scale (exact min/max);
  * single convolution (+InstanceNorm): 2e-4 * output scale;
  * network output in the reference's 150*tanh space: max-abs <= 5e-2, i.e.
    <= 2e-4 after de-processing to [0,1]; 8-bit PSNR >= 50 dB.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from fav_amd import synth, t7

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# kernel-selection switches (FAV_NO_*, FAV_WINO_F2, FAV_W4_*, ...) exist in the DIAGNOSTIC build of the library only (make diag:
# libfav_diag.so, csrc/fav_internal.h diag_env): the child processes of the cross-check tests load that one; the release library ignores them
DIAG_LIB = os.path.join(ROOT, "fast-artistic-videos_amd", "libfav_diag.so")
DIAG_ENV = dict(os.environ, FAV_AMD_LIB=DIAG_LIB)

pytestmark = pytest.mark.gpu


def T(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def psnr8(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


def test_native_library_is_loaded(favlib, cuda):
    assert favlib.device_count() >= 1
    maps = open("/proc/self/maps").read()
    assert "libfav.so" in maps, "the HIP extension must be the code that runs"


# ---------------------------------------------------------------------------------------------- A2 warp
@pytest.mark.parametrize("mode", ["stn", "cpu"])
def test_warp_matches_oracle(favlib, oracle, cuda, mode):
    rng = np.random.default_rng(1)
    for (c, h, w) in [(3, 37, 53), (1, 8, 300), (5, 64, 64)]:
        img = rng.standard_normal((c, h, w)).astype(np.float32)
        flow = (rng.standard_normal((2, h, w)) * 5).astype(np.float32)
        flow[:, 0, :4] = [[-1.5] * 4, [-0.5, 0.0, 0.25, -2.0]]       # fringe band where the two policies differ
        flow[:, -1, -3:] = 0.75
        ref = oracle.warp(img, flow, mode)
        got = favlib.warp(T(img, cuda), T(flow, cuda), favlib.BORDER_CPU if mode == "cpu" else favlib.BORDER_STN).cpu().numpy()
        assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(img).max())
    # batched 4-D input and the identity warp
    imgb = rng.standard_normal((2, 3, 16, 24)).astype(np.float32)
    z = np.zeros((2, 2, 16, 24), np.float32)
    assert np.array_equal(favlib.warp(T(imgb, cuda), T(z, cuda)).cpu().numpy(), imgb)


def test_warp_policies_differ_only_in_fringe(favlib, cuda):
    rng = np.random.default_rng(2)
    img = rng.random((3, 40, 60)).astype(np.float32) + 1
    flow = (rng.standard_normal((2, 40, 60)) * 3).astype(np.float32)
    a = favlib.warp(T(img, cuda), T(flow, cuda), favlib.BORDER_STN).cpu().numpy()
    b = favlib.warp(T(img, cuda), T(flow, cuda), favlib.BORDER_CPU).cpu().numpy()
    yy, xx = np.mgrid[0:40, 0:60]
    sy, sx = yy + flow[0], xx + flow[1]
    interior = (sy >= 0) & (sy <= 39) & (sx >= 0) & (sx <= 59)
    assert np.abs(a - b)[:, interior].max() < 1e-5


WARP_GOLDEN = ["warp_fringe_37x53", "warp_wide_8x600", "warp_batch_resize", "warp_extreme_16x32"]


@pytest.mark.parametrize("name", WARP_GOLDEN)
def test_warp_bit_exact_vs_reference_kernel(favlib, oracle, cuda, golden_dir, name):
    """A2 pinned: fav_warp_bdhw_f32(FAV_BORDER_STN) against the reference's OWN kernel (stnbdhw/BilinearSamplerBDHW.cu:1-109 compiled
    for gfx950 into oracle/_ref/ by oracle/Makefile) -- live on this GPU and through the committed fixtures (its outputs on the
    same inputs): fringe band, >512 columns, batched + resized output, NaN / +-inf / |flow| >= 2^31."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    img, flow = g["img"], g["flow"]
    got = favlib.warp(T(img, cuda), T(flow, cuda), favlib.BORDER_STN).cpu().numpy()
    scale = max(1.0, float(np.abs(img).max()))
    refs = [("committed fixture", g["out_nofma"], g["out"])]
    if oracle.warp_ref_available():          # oracle/_ref travels to the GPU box with the snapshot
        refs.append(("live reference kernel", oracle.warp_ref_gpu(T(img, cuda), T(flow, cuda), contract=False).cpu().numpy(),
                     oracle.warp_ref_gpu(T(img, cuda), T(flow, cuda), contract=True).cpu().numpy()))
    for what, exact, contracted in refs:
        # the expression as written (:103-106), no FMA contraction: every bit, NaNs in the same places
        assert np.array_equal(got, exact, equal_nan=True), f"{name} vs {what}: {(~((got == exact) | (np.isnan(got) & np.isnan(exact)))).sum()} elements differ"
        # the default-flags build may contract into FMAs: same NaN pattern, a few ulp apart
        assert np.array_equal(np.isnan(got), np.isnan(contracted))
        fin = np.isfinite(got) & np.isfinite(contracted)
        assert np.array_equal(np.isfinite(got), np.isfinite(contracted)) and np.abs(got - contracted)[fin].max() <= 1e-6 * scale
    # the C restatement against the same kernel: fp32 form bit-exact, double-rounded form within 1e-6 relative
    for b in range(img.shape[0]):
        o32 = oracle.warp(img[b], flow[b], "stn_f32")
        assert np.array_equal(o32, g["out_nofma"][b], equal_nan=True)


def test_fused_prior_bit_exact_vs_reference_kernel(favlib, oracle, cuda, golden_dir):
    """the prior channels the fused per-frame kernel writes (prep_input_kernel: warp + preprocess + mask + concat) against the
    reference's own warp kernel followed by the fp32 expressions of preprocess.lua:57-62 / core.lua:166-170"""
    if not oracle.warp_ref_available():
        pytest.skip("oracle/_ref/libwarp_ref*.so not built (needs /root/reference at build time)")
    h, w = 48, 64
    rng = np.random.default_rng(77)
    state = (rng.standard_normal((3, h, w)) * 0.4 + 0.5).astype(np.float32)          # float, unclamped previous output
    flo = (rng.standard_normal((h, w, 2)) * 4).astype(np.float32)
    flo[0, :6] = [[0.0, -1.5], [-0.5, -1.5], [0.25, -0.5], [-2.0, 0.0], [np.inf, 0.0], [np.nan, 1.0]]
    flo[-1, -3:] = 0.75
    flo[5, :4] = [[2.0 ** 31, 0.0], [-3e9, 0.5], [0.0, 1e20], [0.0, -np.inf]]
    frame = synth.smooth_frame(h, w, 5)
    cert = np.full((h, w), 255, np.uint8)
    net = favlib.Net(os.path.join(golden_dir, "tiny_model.t7"), 0)
    st = favlib.Stream(net, h, w)
    st.set_state(T(state, cuda))
    st.next_frame_cert(T(frame, cuda), T(flo, cuda), T(cert, cuda))
    in7 = st.last_input().cpu().numpy()
    flow_lua = oracle.flo_to_lua(flo)
    ref = oracle.warp_ref_gpu(T(state[None], cuda), T(flow_lua[None], cuda), contract=False).cpu().numpy()[0]
    mean = np.array([103.939, 116.779, 123.68], np.float32)
    one = np.float32(1.0)
    for c in range(3):               # BGR
        with np.errstate(invalid="ignore"):
            want = (ref[2 - c] * np.float32(255.0) - mean[c]) * one
        assert np.array_equal(in7[3 + c], want, equal_nan=True), f"prior channel {c}: {(~((in7[3 + c] == want) | (np.isnan(want) & np.isnan(in7[3 + c])))).sum()} differ"
    assert np.array_equal(in7[6], np.ones((h, w), np.float32))
    f01 = frame.astype(np.float32) / np.float32(255)
    for c in range(3):
        assert np.array_equal(in7[c], f01[..., 2 - c] * np.float32(255.0) - mean[c])


# ---------------------------------------------------------------------------------------------- A3/A4 mask
@pytest.mark.parametrize("name", ["mask_smooth_64x96.npz", "mask_rand_120x160.npz", "mask_smooth_180x320.npz"])
def test_mask_bit_exact_vs_reference_golden(favlib, cuda, golden_dir, name):
    g = np.load(os.path.join(golden_dir, name))
    m3 = favlib.consistency(T(g["bw"], cuda), T(g["fw"], cuda)).cpu().numpy()
    assert np.array_equal(m3, g["mask3"]), f"{(m3 != g['mask3']).sum()} bytes differ from the reference binary (3-arg)"
    m4 = favlib.consistency(T(g["bw"], cuda), T(g["fw"], cuda), T(g["img"], cuda)).cpu().numpy()
    assert np.array_equal(m4, g["mask4"]), f"{(m4 != g['mask4']).sum()} bytes differ from the reference binary (4-arg)"


@pytest.mark.parametrize("size", [(360, 640), (720, 1280), (5, 3), (130, 1029), (2, 2), (17, 18), (98, 33), (53, 71), (480, 854), (9, 300), (2160, 3840)])
def test_mask_bit_exact_vs_oracle(favlib, oracle, cuda, size):
    h, w = size
    bw = synth.backward_flow(h, w, 5) if h > 8 else synth.random_flow(h, w, 5, 0.5)
    fw = synth.forward_flow_from_backward(bw, 6) if h > 8 else synth.random_flow(h, w, 6, 0.5)
    img = synth.smooth_frame(h, w, 7) if h > 8 else synth.random_frame(h, w, 7)
    m3 = favlib.consistency(T(bw, cuda), T(fw, cuda)).cpu().numpy()
    assert np.array_equal(m3, oracle.consistency(bw, fw))
    m4 = favlib.consistency(T(bw, cuda), T(fw, cuda), T(img, cuda)).cpu().numpy()
    ref4 = oracle.consistency(bw, fw, img)
    assert np.array_equal(m4, ref4), f"{(m4 != ref4).sum()} of {m4.size} bytes differ (4-arg)"
    if h >= 100:
        assert 0.2 < (m3 == 255).mean() < 0.98          # the fixture exercises both outcomes


@pytest.mark.parametrize("kind", ["black", "constant", "half-black"])
def test_mask_four_argument_mode_on_flat_frames(favlib, oracle, cuda, kind):
    """a fade-in's black frame: the structure map is all zero, CMatrix::normalize divides by (max - min) = 0 -> 1, CMatrix::avg's sum
    stays 0 and 4 / avg is infinite (consistencyChecker.cpp:122-124) -- the multi-block sum hands over to the one-block kernel there"""
    h, w = 720, 1280
    bw = synth.backward_flow(h, w, 15); fw = synth.forward_flow_from_backward(bw, 16)
    img = np.zeros((h, w, 3), np.uint8)
    if kind == "constant": img[:] = 77
    if kind == "half-black": img[:, w // 2:] = synth.smooth_frame(h, w, 17)[:, w // 2:]
    m4 = favlib.consistency(T(bw, cuda), T(fw, cuda), T(img, cuda)).cpu().numpy()
    ref4 = oracle.consistency(bw, fw, img)
    assert np.array_equal(m4, ref4), f"{(m4 != ref4).sum()} of {m4.size} bytes differ"


# ---------------------------------------------------------------------------------------------- A5-A7
def test_min_filter_and_assemble(favlib, oracle, cuda):
    rng = np.random.default_rng(3)
    for (h, w) in [(33, 70), (7, 5), (64, 257)]:
        cert = (rng.random((h, w)) > 0.1).astype(np.float32)
        for r in (7, 3, 1):
            assert np.array_equal(favlib.min_filter(T(cert, cuda), r).cpu().numpy(), oracle.min_filter(cert, r))
        grey = rng.random((h, w)).astype(np.float32)
        assert np.array_equal(favlib.min_filter(T(grey, cuda), 7).cpu().numpy(), oracle.min_filter(grey, 7))
        fr = rng.random((3, h, w)).astype(np.float32); wp = (rng.random((3, h, w)) * 1.2 - 0.1).astype(np.float32)
        got = favlib.assemble(T(fr, cuda), T(wp, cuda), T(cert, cuda)).cpu().numpy()
        assert np.abs(got - oracle.assemble(fr, wp, cert)).max() <= 1e-5 * 255
        got0 = favlib.assemble(T(fr, cuda)).cpu().numpy()
        assert np.abs(got0 - oracle.assemble(fr, None, None)).max() <= 1e-5 * 255


# ---------------------------------------------------------------------------------------------- A8 layers
CONV_CASES = [
    # cin, cout, k, stride, pad, H, W        (the canonical net's layer geometries at small spatial size)
    (7, 32, 9, 1, 4, 40, 56),
    (32, 64, 3, 2, 1, 40, 56),
    (64, 128, 3, 2, 1, 21, 29),
    (128, 128, 3, 1, 0, 19, 23),
    (128, 64, 3, 1, 1, 16, 20),
    (64, 3, 9, 1, 4, 24, 40),
    (8, 16, 3, 1, 1, 11, 13),        # M < one tile
    (4, 4, 1, 1, 0, 9, 9),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_matches_oracle(favlib, oracle, cuda, case):
    cin, cout, k, s, p, h, w = case
    rng = np.random.default_rng(cin * 1000 + cout)
    x = rng.standard_normal((cin, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, k, k)) * np.sqrt(2.0 / (cin * k * k))).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, cout).astype(np.float32)
    ref = oracle.conv2d(x, wt, b, s, p)
    got = favlib.conv2d(T(x, cuda), T(wt, cuda), T(b, cuda), s, p).cpu().numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max())
    if cout % 4 == 0:
        g = rng.uniform(0.1, 1, cout).astype(np.float32); bt = rng.standard_normal(cout).astype(np.float32)
        refn = oracle.instnorm_(ref.copy(), g, bt, 1e-5, relu=True)
        gotn = favlib.conv2d(T(x, cuda), T(wt, cuda), T(b, cuda), s, p, T(g, cuda), T(bt, cuda), 1e-5, True).cpu().numpy()
        assert np.abs(gotn - refn).max() <= 5e-4 * max(1.0, np.abs(refn).max())


def test_instance_norm_with_large_mean(favlib, oracle, cuda):
    # |mean| >> std: the per-tile (mean, M2) merge must not lose the variance (one-pass sum of squares would)
    rng = np.random.default_rng(9)
    x = rng.standard_normal((4, 50, 70)).astype(np.float32)
    wt = np.zeros((4, 4, 1, 1), np.float32); wt[np.arange(4), np.arange(4)] = 0.01
    b = np.full(4, 300.0, np.float32)
    g = np.ones(4, np.float32); bt = np.zeros(4, np.float32)
    ref = oracle.instnorm_(oracle.conv2d(x, wt, b, 1, 0), g, bt, 1e-5)
    got = favlib.conv2d(T(x, cuda), T(wt, cuda), T(b, cuda), 1, 0, T(g, cuda), T(bt, cuda), 1e-5, False).cpu().numpy()
    assert np.abs(got - ref).max() < 5e-3            # the fp32 input itself carries ~3e-5/0.01 relative noise


# ---------------------------------------------------------------------------------------------- A8 network
def _layers(path):
    return t7.extract_layers(t7.load(path)["model"])


def test_tiny_network_vs_golden_and_oracle(favlib, oracle, cuda, golden_dir):
    path = os.path.join(golden_dir, "tiny_model.t7")
    g = np.load(os.path.join(golden_dir, "tiny_net_io.npz"))
    net = favlib.Net(path, 0)
    assert net.describe() == favlib.describe_layers(_layers(path))
    got = net.forward(T(g["x"], cuda)).cpu().numpy()
    assert got.shape == g["y"].shape
    assert np.abs(got - g["y"]).max() <= 5e-2
    assert np.abs(got - oracle.net_forward(_layers(path), g["x"])).max() <= 5e-2
    # a second size through the same net (arena re-allocation) and the blob path used for the broadcast
    x2 = (np.random.default_rng(3).standard_normal((7, 36, 44)) * 40).astype(np.float32)
    net2 = favlib.Net(blob=favlib.pack_checkpoint(path), device=0)
    a = net.forward(T(x2, cuda)).cpu().numpy(); b = net2.forward(T(x2, cuda)).cpu().numpy()
    assert np.array_equal(a, b)
    assert np.abs(a - oracle.net_forward(_layers(path), x2)).max() <= 5e-2


@pytest.fixture(scope="module")
def canonical(tmp_path_factory):
    p = str(tmp_path_factory.mktemp("m") / "canonical.t7")
    t7.make_synthetic_checkpoint(p, seed=1234)
    return p


def test_canonical_network_vs_oracle(favlib, oracle, cuda, canonical):
    net = favlib.Net(canonical, 0)
    assert net.param_count() == 1679235
    assert net.output_size(720, 1280) == (720, 1280) and net.output_size(438, 640) == (440, 640)
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((7, 64, 96)) * 60).astype(np.float32)
    ref = oracle.net_forward(_layers(canonical), x)
    got = net.forward(T(x, cuda)).cpu().numpy()
    err = np.abs(got - ref).max()
    assert err <= 1e-2, err                                       # (measured 2-3e-3 with F(4x4); BASELINE.md section 4 allows 5e-2)
    assert np.abs(ref).max() > 100 and np.abs(ref).std() > 20     # not saturated / not trivial


def test_direct_form_kernels_behind_the_switches_match_the_minimal_filtering_ones(favlib, oracle, cuda, canonical, tmp_path):
    """FAV_NO_WINO / FAV_NO_UP2 / FAV_NO_FOLD_UP2 / FAV_NO_FIRST / FAV_NO_S2W select the earlier kernels of the same layers, FAV_UP2_PHASES the
    phase-merged form of U2 + c3s1-64 (read once per process, so child processes run them): every build of the canonical network agrees
    with the oracle, and with the default build far inside the tolerance."""
    import subprocess, sys
    rng = np.random.default_rng(6)
    x = (rng.standard_normal((7, 88, 120)) * 60).astype(np.float32)
    np.save(tmp_path / "x.npy", x)
    child = ("import sys, numpy as np, torch; sys.path.insert(0, %r); import fav_amd\n"
             "x = np.load(%r); net = fav_amd.Net(%r, 0)\n"
             "np.save(%r, net.forward(torch.from_numpy(x).cuda()).cpu().numpy())\n"
             % (os.path.join(ROOT, "fast-artistic-videos_amd", "python"), str(tmp_path / "x.npy"), canonical, str(tmp_path / "direct.npy")))
    env = dict(DIAG_ENV, FAV_NO_WINO="1", FAV_NO_UP2="1", FAV_NO_FOLD_UP2="1", FAV_NO_FIRST="1", FAV_NO_S2W="1")
    subprocess.check_call([sys.executable, "-c", child], env=env, timeout=300)
    direct = np.load(tmp_path / "direct.npy")
    ref = oracle.net_forward(_layers(canonical), x)
    got = favlib.Net(canonical, 0).forward(T(x, cuda)).cpu().numpy()
    assert np.abs(direct - ref).max() <= 5e-2 and np.abs(got - ref).max() <= 5e-2
    assert np.abs(got - direct).max() <= 2e-2, np.abs(got - direct).max()
    # FAV_UP2_PHASES: U2 + c3s1-64 as four phase-wise 2x2 convolutions (conv3_up2_kernel) instead of the nine-position form
    env = dict(DIAG_ENV, FAV_UP2_PHASES="1")
    subprocess.check_call([sys.executable, "-c", child], env=env, timeout=300)
    phases = np.load(tmp_path / "direct.npy")
    assert np.abs(phases - ref).max() <= 5e-2 and np.abs(got - phases).max() <= 2e-2, (np.abs(phases - ref).max(), np.abs(got - phases).max())
    # FAV_WINO_F2: the residual convolutions as F(2x2,3x3) (conv3_wino_kernel, the reference-accuracy form) instead of F(4x4,3x3)
    env = dict(DIAG_ENV, FAV_WINO_F2="1")
    subprocess.check_call([sys.executable, "-c", child], env=env, timeout=300)
    f2 = np.load(tmp_path / "direct.npy")
    e2, e4 = np.abs(f2 - ref).max(), np.abs(got - ref).max()
    print("canonical 88x120: F(2x2) max-abs %.3e, F(4x4) max-abs %.3e (150*tanh space)" % (e2, e4))
    assert e2 <= 5e-2 and np.abs(got - f2).max() <= 2e-2 and not np.array_equal(got, f2)


def test_accumulator_statistics_match_the_partials_form(favlib, oracle, cuda, canonical, tmp_path):
    """Round 5: nine of the sixteen InstanceNorms of the canonical network (conv -> IN -> ReLU -> conv inside the residual branches, and the
    branch's last IN in front of a plain join) take their statistics through ACCUMULATORS -- every work unit of the producing F(4x4)
    launch adds its exact 2^-40 fixed-point (sum, sum of squares) with 64-bit integer atomics, the consumer forms scale / shift in its
    prologue, no in_finalize launch in between (kernels_wino4.hip, Affine::acc1).  Integer addition commutes: the result must not depend
    on the order the atomics arrive in -- the same bits from run to run -- and must agree with the partials + in_finalize form
    (FAV_NO_ACC_STATS=1, a child process: the switch is read once) to rounding.  Two frame sizes, so that the accumulators' zeroing
    between frames (the consumer zeroes the other parity) is exercised across a re-allocation of the activation arena as well."""
    import subprocess, sys
    rng = np.random.default_rng(21)
    xs = [(rng.standard_normal((7, 88, 120)) * 60).astype(np.float32), (rng.standard_normal((7, 150, 210)) * 60).astype(np.float32)]
    for k, x in enumerate(xs):
        np.save(tmp_path / ("x%d.npy" % k), x)
    child = ("import sys, numpy as np, torch; sys.path.insert(0, %r); import fav_amd\n"
             "net = fav_amd.Net(%r, 0)\n"
             "for k in (0, 1, 0):\n"
             "    x = torch.from_numpy(np.load(%r %% k)).cuda()\n"
             "    np.save(%r %% k, net.forward(x).cpu().numpy())\n"
             % (os.path.join(ROOT, "fast-artistic-videos_amd", "python"), canonical, str(tmp_path / "x%d.npy"), str(tmp_path / "p%d.npy")))
    subprocess.check_call([sys.executable, "-c", child], env=dict(DIAG_ENV, FAV_NO_ACC_STATS="1"), timeout=300)
    net = favlib.Net(canonical, 0)
    first = {}
    for rep in range(3):                                  # frames alternate sizes: parity flips every forward
        for k in (0, 1):
            got = net.forward(T(xs[k], cuda)).cpu().numpy()
            if rep == 0:
                first[k] = got
                ref = oracle.net_forward(_layers(canonical), xs[k])
                part = np.load(tmp_path / ("p%d.npy" % k))
                d = np.abs(got - part).max()
                print("accumulator vs partials statistics at %dx%d: max-abs %.3e (150*tanh space)" % (xs[k].shape[2], xs[k].shape[1], d))
                assert np.abs(got - ref).max() <= 1e-2 and np.abs(part - ref).max() <= 1e-2
                assert d <= 2e-3, d
            else:
                assert np.array_equal(got, first[k]), (rep, k)      # bit-deterministic, whatever order the atomics arrive in
    # forwards that do NOT use the accumulators in between (the bf16-operand mode has other kernels in the residual stage): one, then two
    for nb in (1, 2):
        net.set_precision(True)
        for _ in range(nb):
            net.forward(T(xs[0], cuda))
        net.set_precision(False)
        assert np.array_equal(net.forward(T(xs[0], cuda)).cpu().numpy(), first[0]), nb
        assert np.array_equal(net.forward(T(xs[1], cuda)).cpu().numpy(), first[1]), nb
    net.check()


@pytest.mark.parametrize("size,grid,lazy", [((88, 120), 4, 0), ((88, 120), 7, 1), ((90, 122), 5, 0), ((360, 640), 37, 1), ((720, 1280), 0, 0)])
def test_stream_k_shares_of_the_winograd_layers(favlib, cuda, canonical, tmp_path, size, grid, lazy):
    """conv3_wino4_kernel deals a launch with a thin last round out as one sequence of 16-channel slices, an equal share per block; a unit
    cut by a share boundary is computed in two parts by two blocks and whichever finishes second adds them (kernels_wino4.hip, `stream`).
    FAV_W4_GRID caps the number of blocks, so small images reach the many-shares case (pending joins included: their joined rows are
    written slice by slice by whichever part stages them); FAV_W4_NO_STREAM computes whole units only.  Not the same bits -- a cut unit's
    channels are summed in two pieces -- but the same numbers to rounding, and the same bits from run to run whoever arrives second."""
    import subprocess, sys
    H, W = size
    rng = np.random.default_rng(3 * H + W)
    x = (rng.standard_normal((7, H, W)) * 60).astype(np.float32)
    np.save(tmp_path / "x.npy", x)
    child = ("import sys, numpy as np, torch; sys.path.insert(0, %r); import fav_amd\n"
             "x = torch.from_numpy(np.load(%r)).cuda(); net = fav_amd.Net(%r, 0)\n"
             "a = net.forward(x).cpu().numpy(); b = net.forward(x).cpu().numpy(); c = net.forward(x).cpu().numpy()\n"
             "assert np.array_equal(a, b) and np.array_equal(a, c)\n"
             "np.save(sys.argv[1], a)\n"
             % (os.path.join(ROOT, "fast-artistic-videos_amd", "python"), str(tmp_path / "x.npy"), canonical))
    envg = {"FAV_W4_GRID": str(grid)} if grid else {}
    if lazy: envg["FAV_LAZY_JOIN"] = "1"           # residual joins pending in the next convolution (conv3_wino4_kernel<2>) instead of launched
    subprocess.check_call([sys.executable, "-c", child, str(tmp_path / "whole.npy")], env=dict(DIAG_ENV, FAV_W4_NO_STREAM="1", **envg), timeout=300)
    subprocess.check_call([sys.executable, "-c", child, str(tmp_path / "stream.npy")], env=dict(DIAG_ENV, **envg), timeout=300)
    whole, stream = np.load(tmp_path / "whole.npy"), np.load(tmp_path / "stream.npy")
    assert np.isfinite(stream).all()
    d = np.abs(stream - whole).max()
    print("stream-K vs whole units at %dx%d, grid cap %d: max-abs %.3e (150*tanh space)" % (H, W, grid, d))
    assert d <= 5e-3, d
    assert not np.array_equal(stream, whole)       # (the shares really cut units at these sizes)


@pytest.mark.parametrize("size", [(88, 120), (90, 122), (360, 640), (720, 1280)])
def test_pending_residual_joins_give_the_bits_of_the_launched_ones(favlib, cuda, canonical, tmp_path, size):
    """The joins of residual blocks 2-4 (models_video.lua:41-53) are not launched: the next block's first Winograd convolution forms
    z = skip + IN(branch) while it stages its halo (same operations in res_add_kernel's order) and writes it out as that block's skip;
    the branch's output is laid out under the skip tensor for that.  FAV_NO_LAZY_JOIN (read once per process: a child process runs it)
    launches every join -- same bits out of the whole network, at sizes with ragged last units in both directions."""
    import subprocess, sys
    H, W = size
    rng = np.random.default_rng(H + W)
    x = (rng.standard_normal((7, H, W)) * 60).astype(np.float32)
    np.save(tmp_path / "x.npy", x)
    child = ("import sys, numpy as np, torch; sys.path.insert(0, %r); import fav_amd\n"
             "x = np.load(%r); net = fav_amd.Net(%r, 0)\n"
             "np.save(%r, net.forward(torch.from_numpy(x).cuda()).cpu().numpy())\n"
             % (os.path.join(ROOT, "fast-artistic-videos_amd", "python"), str(tmp_path / "x.npy"), canonical, str(tmp_path / "eager.npy")))
    # (launched joins are the default again since round 4: FAV_LAZY_JOIN asks for the pending form -- in the F(4x4) kernel here, and in the
    #  F(2x2) kernel, whose default it still is, with FAV_WINO_F2)
    net = favlib.Net(canonical, 0)
    eager = net.forward(T(x, cuda)).cpu().numpy()
    assert np.isfinite(eager).all()
    assert np.array_equal(eager, net.forward(T(x, cuda)).cpu().numpy())
    for extra in ({"FAV_LAZY_JOIN": "1"}, {"FAV_WINO_F2": "1"}):
        ref_env = dict(DIAG_ENV, **extra)
        subprocess.check_call([sys.executable, "-c", child], env=ref_env, timeout=300)
        pending = np.load(tmp_path / "eager.npy")
        if "FAV_WINO_F2" in extra:      # the F(2x2) kernels: pending (their default) against launched, both in children
            subprocess.check_call([sys.executable, "-c", child], env=dict(ref_env, FAV_NO_LAZY_JOIN="1"), timeout=300)
            launched = np.load(tmp_path / "eager.npy")
        else:
            launched = eager
        assert np.array_equal(pending, launched), (extra, np.abs(pending - launched).max())


@pytest.mark.parametrize("inorm", [True, False])
def test_image_model_vs_oracle(favlib, oracle, cuda, tmp_path, golden_dir, inorm):
    """-model_img <file> -- 3-channel image model with nn.SpatialFullConvolution ('u' layers,
    models_video.lua:99-102) and InstanceNorm or evaluate-mode BatchNorm, used for frames without a prior (core.lua:59-66,146)"""
    pi = str(tmp_path / "img.t7")
    t7.make_synthetic_checkpoint(pi, arch="c9s1-8,d16,d32,R32,R32,u16,u8,c9s1-3", seed=5, in_channels=3, use_instance_norm=inorm)
    img_layers = _layers(pi)
    net_img = favlib.Net(pi, 0)
    assert net_img.describe() == favlib.describe_layers(img_layers)
    x = (np.random.default_rng(2).standard_normal((3, 40, 56)) * 50).astype(np.float32)
    got = net_img.forward(T(x, cuda)).cpu().numpy()
    ref = oracle.net_forward(img_layers, x)
    assert got.shape == ref.shape == (3, 40, 56)
    assert np.abs(got - ref).max() <= 5e-2
    # in the pipeline: first frame through the image model, the following ones through the video model
    vid = os.path.join(golden_dir, "tiny_model.t7")
    h, w = 48, 64
    frames, bws, fws = _clip(h, w, 2, 80)
    net = favlib.Net(vid, 0)
    st = favlib.Stream(net, h, w)
    st.set_image_net(net_img)
    o0, _ = st.first_frame(T(frames[0], cuda))
    o1, _ = st.next_frame_flow(T(frames[1], cuda), T(bws[1], cuda), T(fws[1], cuda))
    ref_s = oracle.Stylizer(_layers(vid))
    r0 = ref_s.first(_f01(frames[0]), image_layers=img_layers)
    assert np.abs(o0.cpu().numpy() - r0).max() <= 2e-4
    ref_s.last = o0.cpu().numpy()
    m = oracle.consistency(bws[1], fws[1])
    r1 = ref_s.next(_f01(frames[1]), bws[1], m.astype(np.float32) / np.float32(255))
    assert np.abs(o1.cpu().numpy() - r1).max() <= 2e-4


@pytest.mark.parametrize("cin", [16, 48, 80, 64])
def test_f4x4_kernel_on_an_odd_number_of_channel_slices(favlib, oracle, cuda, tmp_path, cin):
    """With -padding_type reflect a c3s1-128 layer is a padding module + a 3x3 convolution with padW = 0 (models_video.lua:70-75): it runs
    on the F(4x4) kernel with however many input channels the layer before it left -- 16, 48, 80 channels are 1, 3, 5 slices of 16.  The
    kernel requests its raw rows two slices at a time (round 6): with an odd count the last request's second half lies past the pixel's
    channels and is never taken.  Whole-unit launches and (FAV_W4_GRID, diagnostic library, in a child process) a stream-K dealing."""
    arch = "c3s1-%d,c3s1-128,c9s1-3" % cin
    p = str(tmp_path / "m.t7")
    t7.make_synthetic_checkpoint(p, arch=arch, seed=77, padding_type="reflect")
    layers = _layers(p)
    h, w = 37, 70
    x = (np.random.default_rng(31).standard_normal((7, h, w)) * 50).astype(np.float32)
    ref = oracle.net_forward(layers, x)
    net, got, kids = _wide_forward(favlib, cuda, p, x)
    assert kids[1] in (728, 729), kids                # the F(4x4) kernel (fav_internal.h: conv kernel ids)
    assert got.shape == ref.shape == (3, h, w)
    err = np.abs(got - ref).max()
    assert err <= 2e-2 and np.abs(ref).std() > 5, err
    np.save(tmp_path / "x.npy", x)
    child = ("import sys, numpy as np, torch; sys.path.insert(0, %r); import fav_amd\n"
             "net = fav_amd.Net(%r, 0); x = torch.from_numpy(np.load(%r)).cuda()\n"
             "np.save(sys.argv[1], net.forward(x).cpu().numpy()); net.check()\n" % (os.path.join(ROOT, "fast-artistic-videos_amd", "python"), p, str(tmp_path / "x.npy")))
    subprocess.check_call([sys.executable, "-c", child, str(tmp_path / "stream.npy")], env=dict(DIAG_ENV, FAV_W4_GRID="5"), timeout=300)
    stream = np.load(tmp_path / "stream.npy")
    assert np.abs(stream - ref).max() <= 2e-2, float(np.abs(stream - ref).max())


@pytest.mark.parametrize("ptype", ["reflect", "replicate", "zero", "none"])
@pytest.mark.parametrize("arch", ["c9s1-8,d16,d32,R32,R32,R32,U2,c3s1-16,U2,c9s1-3", "c9s1-32,d64,d128,R128,R128,R128,U2,c3s1-64,U2,c9s1-3"])
def test_padding_types_vs_oracle(favlib, oracle, cuda, tmp_path, ptype, arch):
    """train_video.lua:25 -padding_type reflect | replicate | zero | none (reflect-start is every other test): three residual blocks per type.
    reflect / replicate put a padding MODULE in front of every convolution (models_video.lua:12-16,27-31,70-75): the leading symmetric
    reflection is folded into the input assembly like reflect-start's, every other one is a gather launch (pad_nhwc_kernel) through which
    the pending InstanceNorm / ReLU passes unchanged -- also behind an upsampling; with the canonical filter counts the block convolutions
    stay on the F(4x4) kernel (they see a padded input with padW = 0).  Net forward and two recurrent frames against the oracle."""
    p = str(tmp_path / "m.t7")
    t7.make_synthetic_checkpoint(p, arch=arch, seed=21, padding_type=ptype)
    layers = _layers(p)
    net = favlib.Net(p, 0)
    assert net.describe() == favlib.describe_layers(layers)
    h, w = 56, 72
    x = (np.random.default_rng(4).standard_normal((7, h, w)) * 40).astype(np.float32)
    got = net.forward(T(x, cuda)).cpu().numpy()
    ref = oracle.net_forward(layers, x)
    assert got.shape == ref.shape
    assert ptype == "none" or got.shape == (3, h, w)
    assert np.abs(got - ref).max() <= 5e-2, float(np.abs(got - ref).max())
    if ptype == "none":
        return                       # (frames shrink: fast_artistic_video.lua cannot warp such outputs either)
    frames, bws, fws = _clip(h, w, 3, 90)
    st = favlib.Stream(net, h, w)
    ref_s = oracle.Stylizer(layers)
    o, _ = st.first_frame(T(frames[0], cuda))
    assert np.abs(o.cpu().numpy() - ref_s.first(_f01(frames[0]))).max() <= 2e-4
    for i in (1, 2):
        ref_s.last = o.cpu().numpy()                                   # teacher-forced
        o, _ = st.next_frame_flow(T(frames[i], cuda), T(bws[i], cuda), T(fws[i], cuda))
        m = oracle.consistency(bws[i], fws[i])
        assert np.array_equal(st.last_mask().cpu().numpy(), m)
        r = ref_s.next(_f01(frames[i]), bws[i], m.astype(np.float32) / np.float32(255))
        assert np.abs(o.cpu().numpy() - r).max() <= 2e-4, (i, float(np.abs(o.cpu().numpy() - r).max()))
    net.check()


def test_non_finite_activations_are_deterministic_and_leave_nothing_behind(favlib, cuda, canonical):
    """a NaN / Inf in the input (a diverged free-running clip, a damaged checkpoint) must not be undefined behaviour anywhere: the
    accumulator form of the InstanceNorm statistics (exact fixed point, round 5) cannot hold a non-finite sum -- its producer poisons the
    high words instead (STAT_NONFINITE, csrc/fav_internal.h) and the consumer forms NaN scale / shift, like in_finalize_kernel does for the
    partials form -- so the frame is the same bits every time it is computed, and the frames after it are clean again (the accumulators'
    halves alternate and are zeroed by their consumers).
    (Known difference, DESIGN.md section 7: the kernels apply ReLU as max(x, 0), which maps NaN to 0 -- nn.ReLU's `x <= 0 ? 0 : x`
    [THNN Threshold, recalled] keeps it -- so a non-finite frame comes out as finite garbage here and as NaN in the reference; neither is
    a picture.)"""
    net = favlib.Net(canonical, 0)
    rng = np.random.default_rng(31)
    x = (rng.standard_normal((7, 96, 128)) * 60).astype(np.float32)
    clean = net.forward(T(x, cuda)).cpu().numpy()
    assert np.isfinite(clean).all()
    for bad in (np.nan, np.inf, -np.inf):
        xb = x.copy(); xb[2, 40, 50] = bad
        outs = [net.forward(T(xb, cuda)).cpu().numpy() for _ in range(3)]
        assert all(np.array_equal(outs[0], o, equal_nan=True) for o in outs[1:])
        for _ in range(2):                                                   # both parities of the accumulators
            again = net.forward(T(x, cuda)).cpu().numpy()
            assert np.array_equal(again, clean)
    net.check()


def test_unsupported_models_fail_with_status(favlib, cuda, tmp_path):
    p = str(tmp_path / "odd.t7")
    t7.make_synthetic_checkpoint(p, arch="c9s1-8,d16,R16,U2,c9s1-3", seed=2, in_channels=5)
    with pytest.raises(favlib.FavError, match="5 input channels"):
        favlib.Net(p, 0)
    with pytest.raises(favlib.FavError):
        favlib.Net(str(tmp_path / "missing.t7"), 0)


@pytest.mark.parametrize("arch", ["c9s1-48,d96,R96,R96,U2,c3s1-24,U2,c9s1-3", "c9s1-12,d20,d40,R40,U2,c3s1-20,U2,c9s1-3"])
def test_filter_counts_that_are_not_powers_of_two(favlib, oracle, cuda, tmp_path, arch):
    """models_video.lua:55-140 builds any `c9s1-48,d96,...` and the VR checkpoints have "more filters" (README.md:141): such networks
    run with their channel counts padded to the next power of two by zero filters (exact: the padded channels are identically zero),
    describe() and the parameter count still show the checkpoint's own sizes"""
    p = str(tmp_path / "odd.t7")
    t7.make_synthetic_checkpoint(p, arch=arch, seed=13)
    layers = _layers(p)
    net = favlib.Net(p, 0)
    assert net.describe() == favlib.describe_layers(layers)
    rng = np.random.default_rng(4)
    x = (rng.standard_normal((7, 64, 88)) * 50).astype(np.float32)
    got = net.forward(T(x, cuda)).cpu().numpy()
    ref = oracle.net_forward(layers, x)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 5e-2, np.abs(got - ref).max()
    assert np.abs(ref).std() > 10


@pytest.mark.parametrize("size", [(438, 640), (480, 854), (50, 70), (53, 71)])
def test_frame_sizes_that_are_not_multiples_of_four(favlib, oracle, cuda, golden_dir, size):
    """stylizeVideo_deepflow.sh:72-78 lets the user pick any w:h (854x480 is not a multiple of 4): the network's output is then up to
    3 pixels larger than the frame (two stride-2 convolutions, two x2 upsamplings); the reference keeps and saves that larger image
    and warps it with the FLOW's size (BilinearSamplerBDHW.lua:71).  Three frames against the oracle, fused check and certainty path."""
    h, w = size
    path = os.path.join(golden_dir, "tiny_model.t7")
    layers = _layers(path)
    net = favlib.Net(path, 0)
    ho, wo = net.output_size(h, w)
    assert (ho, wo) != (h, w) and ho >= h and wo >= w and ho - h < 4 and wo - w < 4
    frames, bws, fws = _clip(h, w, 3, 40)
    st = favlib.Stream(net, h, w)
    assert (st.Ho, st.Wo) == (ho, wo)
    ref = oracle.Stylizer(layers)
    o0, u0 = st.first_frame(T(frames[0], cuda), want_u8=True)
    r0 = ref.first(_f01(frames[0]))
    assert tuple(o0.shape) == (3, ho, wo) == r0.shape and tuple(u0.shape) == (ho, wo, 3)
    assert np.abs(o0.cpu().numpy() - r0).max() <= 2e-4
    m1 = oracle.consistency(bws[1], fws[1])
    o1, _ = st.next_frame_flow(T(frames[1], cuda), T(bws[1], cuda), T(fws[1], cuda))
    assert np.array_equal(st.last_mask().cpu().numpy(), m1)
    tf = oracle.Stylizer(layers); tf.last = o0.cpu().numpy()
    r1 = tf.next(_f01(frames[1]), bws[1], m1.astype(np.float32) / np.float32(255))           # teacher-forced: the larger state warped on the flow's grid
    assert np.abs(o1.cpu().numpy() - r1).max() <= 2e-4
    m2 = oracle.consistency(bws[2], fws[2], frames[2])
    o2, u2 = st.next_frame_cert(T(frames[2], cuda), T(bws[2], cuda), T(m2, cuda), want_u8=True)
    tf.last = o1.cpu().numpy()
    r2 = tf.next(_f01(frames[2]), bws[2], m2.astype(np.float32) / np.float32(255))
    assert np.abs(o2.cpu().numpy() - r2).max() <= 2e-4
    assert np.abs(u2.cpu().numpy().astype(int) - oracle.to_u8_hwc(r2).astype(int)).max() <= 1
    # the PNG of the larger frame, and the state round trip (-continue_with)
    import io
    from PIL import Image
    png = favlib.png_encode(None, from_stream=st)
    assert np.array_equal(np.array(Image.open(io.BytesIO(png)).convert("RGB")), u2.cpu().numpy())
    assert tuple(st.state().shape) == (3, ho, wo)


# ---------------------------------------------------------------------------------------------- pipeline
def _clip(h, w, n, seed):
    frames = [synth.smooth_frame(h, w, seed + i) for i in range(n)]
    bws = [None] + [synth.backward_flow(h, w, seed + 100 + i) for i in range(1, n)]
    fws = [None] + [synth.forward_flow_from_backward(bws[i], seed + 200 + i) for i in range(1, n)]
    return frames, bws, fws


def _f01(u8):
    return np.transpose(u8, (2, 0, 1)).astype(np.float32) / np.float32(255)


@pytest.mark.parametrize("mode", ["cert", "flow3", "flow4"])
def test_stream_vs_oracle_recurrent(favlib, oracle, cuda, golden_dir, mode):
    path = os.path.join(golden_dir, "tiny_model.t7")
    h, w, n = 48, 72, 4
    frames, bws, fws = _clip(h, w, n, 10)
    layers = _layers(path)
    net = favlib.Net(path, 0)
    st = favlib.Stream(net, h, w)
    free = oracle.Stylizer(layers)            # free-running oracle
    outs = []
    for i in range(n):
        if i == 0:
            o, u8 = st.first_frame(T(frames[0], cuda), want_u8=True)
            r = free.first(_f01(frames[0]))
        else:
            img = frames[i] if mode == "flow4" else None
            mask = oracle.consistency(bws[i], fws[i], img)
            if mode == "cert":
                o, u8 = st.next_frame_cert(T(frames[i], cuda), T(bws[i], cuda), T(mask, cuda), want_u8=True)
            else:
                o, u8 = st.next_frame_flow(T(frames[i], cuda), T(bws[i], cuda), T(fws[i], cuda), use_structure=(mode == "flow4"), want_u8=True)
                assert np.array_equal(st.last_mask().cpu().numpy(), mask)
            # teacher-forced: the oracle continues from the GPU's previous output
            tf = oracle.Stylizer(layers); tf.last = outs[-1]
            rt = tf.next(_f01(frames[i]), bws[i], mask.astype(np.float32) / np.float32(255))
            assert np.abs(o.cpu().numpy() - rt).max() <= 2e-4
            r = free.next(_f01(frames[i]), bws[i], mask.astype(np.float32) / np.float32(255))
        got = o.cpu().numpy(); outs.append(got)
        assert np.abs(got - r).max() <= 1e-3, f"free-running drift at frame {i}"
        q = oracle.to_u8_hwc(got)
        assert np.abs(u8.cpu().numpy().astype(int) - q.astype(int)).max() == 0
        assert psnr8(u8.cpu().numpy(), oracle.to_u8_hwc(r)) >= 50.0
    assert np.array_equal(st.state().cpu().numpy(), outs[-1])


@pytest.mark.parametrize("size", [(48, 72), (130, 1029), (301, 258)])
@pytest.mark.parametrize("host_ordered", [False, True])
def test_stream_mask_lookahead(favlib, oracle, cuda, golden_dir, host_ordered, size):
    """fav_stream_prefetch_mask: the next frame's mask computed on the side stream gives identical frames -- event-ordered (default)
    and host-ordered (fav_stream_set_host_ordered: no event in any queue, the caller has seen the inputs complete, the consumer waits on
    the host for a sequence number in host-mapped memory).  In the 4-argument mode this is the PACKED form of the mask pipeline (round 6:
    eight blocks per long-lived kernel, the ones on the XCDs the network leaves a CU free on work -- xcd_share; four lines per lane, LDS
    ring): sizes with one and with several groups of 256 lines, line counts that are not multiples of 4 or 256, lines shorter and longer
    than the ring"""
    import torch
    path = os.path.join(golden_dir, "tiny_model.t7")
    (h, w), n = size, 4
    frames, bws, fws = _clip(h, w, n, 60)
    net = favlib.Net(path, 0)
    fr = [T(f, cuda) for f in frames]; bw = [None] + [T(b, cuda) for b in bws[1:]]; fw = [None] + [T(f, cuda) for f in fws[1:]]
    torch.cuda.synchronize()                 # (host-ordered mode: the inputs are complete before any look-ahead starts)
    for structure in (False, True):
        a = favlib.Stream(net, h, w); b = favlib.Stream(net, h, w)
        b.set_host_ordered(host_ordered)
        oa, _ = a.first_frame(fr[0]); ob, _ = b.first_frame(fr[0])
        b.prefetch_mask(fr[1], bw[1], fw[1], structure)
        for i in range(1, n):
            oa, _ = a.next_frame_flow(fr[i], bw[i], fw[i], use_structure=structure)
            if i + 1 < n:           # two masks in flight: i (not yet consumed) and i+1
                b.prefetch_mask(fr[i + 1], bw[i + 1], fw[i + 1], structure)
            ob, _ = b.next_frame_flow(fr[i], bw[i], fw[i], use_structure=structure)
            assert torch.equal(oa, ob)
            assert np.array_equal(b.last_mask().cpu().numpy(), oracle.consistency(bws[i], fws[i], frames[i] if structure else None))


def test_stream_options(favlib, oracle, cuda, golden_dir):
    path = os.path.join(golden_dir, "tiny_model.t7")
    h, w = 48, 64
    frames, bws, fws = _clip(h, w, 2, 30)
    layers = _layers(path)
    net = favlib.Net(path, 0)
    mask = oracle.consistency(bws[1], fws[1])
    big = bws[1] + np.float32(9.0)        # push part of the frame out of bounds so -fix_occlusions has something to fix
    for border, bname in ((favlib.BORDER_STN, "stn"), (favlib.BORDER_CPU, "cpu")):
        for fix in (False, True):
            st = favlib.Stream(net, h, w, border=border, min_filter_r=3, invert_occlusion=True, fix_occlusions=fix)
            o0, _ = st.first_frame(T(frames[0], cuda))
            o1, _ = st.next_frame_cert(T(frames[1], cuda), T(big, cuda), T(255 - mask, cuda))
            ref = oracle.Stylizer(layers, border=bname, min_filter_r=3, invert_occlusion=True, fix_occlusions=fix)
            ref.first(_f01(frames[0]))
            r1 = ref.next(_f01(frames[1]), big, (255 - mask).astype(np.float32) / np.float32(255))
            assert np.abs(o1.cpu().numpy() - r1).max() <= 2e-4, (bname, fix)
    # -fill_occlusions uniform-random (core.lua:108-117) with the documented counter RNG: first frame = noise prior, then
    # noise only where the certainty is low; a different seed must give a different frame
    st = favlib.Stream(net, h, w, fill_random=True, seed=5)
    ref = oracle.Stylizer(layers, fill_random=True, seed=5)
    o0, _ = st.first_frame(T(frames[0], cuda)); r0 = ref.first(_f01(frames[0]))
    assert np.abs(o0.cpu().numpy() - r0).max() <= 2e-4
    ref.last = o0.cpu().numpy()
    o1, _ = st.next_frame_cert(T(frames[1], cuda), T(bws[1], cuda), T(mask, cuda))
    r1 = ref.next(_f01(frames[1]), bws[1], mask.astype(np.float32) / np.float32(255))
    assert np.abs(o1.cpu().numpy() - r1).max() <= 2e-4
    st2 = favlib.Stream(net, h, w, fill_random=True, seed=6)
    assert np.abs(st2.first_frame(T(frames[0], cuda))[0].cpu().numpy() - o0.cpu().numpy()).max() > 1e-3
    st50 = favlib.Stream(net, 50, 64)                # not a multiple of 4: accepted since round 3, the stylised frames are 52 x 64
    assert (st50.Ho, st50.Wo) == (52, 64)
    with pytest.raises(favlib.FavError, match="smaller than the reflection padding"):
        favlib.Stream(net, 8, 64)
    st = favlib.Stream(net, h, w)
    with pytest.raises(favlib.FavError, match="previous"):
        st.next_frame_cert(T(frames[1], cuda), T(bws[1], cuda), T(mask, cuda))


# ---------------------------------------------------------------------------------------------- full size
@pytest.mark.parametrize("size,opts", [((60, 76), dict()), ((53, 71), dict(invert_occlusion=True, fix_occlusions=True)), ((96, 160), dict(fill_random=True, seed=5, min_filter_r=5))],
                         ids=["pad-overlap", "options-odd-size", "random-fill-r5"])
def test_fused_check_and_input_assembly_gives_the_bytes_of_the_two_launches(favlib, cuda, canonical, tmp_path, size, opts):
    """Round 5: fav_stream_next_frame_flow runs the forward-backward check, the certainty options, the erosion AND the input assembly
    (warp of the previous output, pre-processing, masking, fill, reflection padding) of a frame in ONE tile kernel (check_prep_kernel);
    FAV_NO_CHECK_PREP=1 (read once per process: a child runs it) keeps them as two launches (min_filter_kernel<2> + prep_input_kernel).
    Same mask bytes, same network input, same stylised frames bit for bit -- the stylised frame depends on every padded input pixel, so
    the reflections the tile kernel writes itself are covered; 60 rows with 40 of padding make a pixel appear three times per axis."""
    import subprocess, sys
    h, w = size
    frames, bws, fws = _clip(h, w, 3, 300 + h)
    np.savez(tmp_path / "clip.npz", f0=frames[0], f1=frames[1], f2=frames[2], b1=bws[1], b2=bws[2], w1=fws[1], w2=fws[2])
    child = ("import sys, numpy as np, torch; sys.path.insert(0, %r); import fav_amd\n"
             "d = np.load(%r); T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()\n"
             "net = fav_amd.Net(%r, 0); st = fav_amd.Stream(net, %d, %d, **%r)\n"
             "st.first_frame(T(d['f0']))\n"
             "o1, _ = st.next_frame_flow(T(d['f1']), T(d['b1']), T(d['w1'])); i1 = st.last_input().cpu().numpy(); m1 = st.last_mask().cpu().numpy()\n"
             "o2, _ = st.next_frame_flow(T(d['f2']), T(d['b2']), T(d['w2'])); m2 = st.last_mask().cpu().numpy()\n"
             "np.savez(sys.argv[1], o1=o1.cpu().numpy(), o2=o2.cpu().numpy(), i1=i1, m1=m1, m2=m2)\n"
             % (os.path.join(ROOT, "fast-artistic-videos_amd", "python"), str(tmp_path / "clip.npz"), canonical, h, w, opts))
    subprocess.check_call([sys.executable, "-c", child, str(tmp_path / "two.npz")], env=dict(DIAG_ENV, FAV_NO_CHECK_PREP="1"), timeout=300)
    subprocess.check_call([sys.executable, "-c", child, str(tmp_path / "one.npz")], env=os.environ.copy(), timeout=300)
    a, b = np.load(tmp_path / "two.npz"), np.load(tmp_path / "one.npz")
    for k in ("m1", "m2", "i1", "o1", "o2"):
        assert np.array_equal(a[k], b[k]), k
    assert a["m1"].min() == 0 and a["m1"].max() == 255 and np.abs(a["o2"]).std() > 0.01


def test_full_size_properties_1280x720(favlib, oracle, cuda, canonical):
    """BASELINE config 3 geometry: size-independent properties (the oracle needs ~20 s per frame here)."""
    import torch
    h, w = 720, 1280
    net = favlib.Net(canonical, 0)
    st = favlib.Stream(net, h, w)
    f0, f1 = synth.random_frame(h, w, 1), synth.random_frame(h, w, 2)
    bw = synth.backward_flow(h, w, 3); fw = synth.forward_flow_from_backward(bw, 4)
    o0, _ = st.first_frame(T(f0, cuda))
    o1, u1 = st.next_frame_flow(T(f1, cuda), T(bw, cuda), T(fw, cuda), want_u8=True)
    torch.cuda.synchronize()
    a0, a1 = o0.cpu().numpy(), o1.cpu().numpy()
    assert np.isfinite(a0).all() and np.isfinite(a1).all()
    lo, hi = (-150 + 103.939) / 255, (150 + 123.68) / 255          # 150*tanh + mean, /255
    assert a1.min() >= lo - 1e-4 and a1.max() <= hi + 1e-4
    assert np.array_equal(st.last_mask().cpu().numpy(), oracle.consistency(bw, fw))
    # determinism: a second stream replays bit-identically
    st2 = favlib.Stream(net, h, w)
    p0, _ = st2.first_frame(T(f0, cuda))
    p1, _ = st2.next_frame_flow(T(f1, cuda), T(bw, cuda), T(fw, cuda))
    assert torch.equal(p0, o0) and torch.equal(p1, o1)
    # the fused pipeline equals the operator-level composition (warp -> min-filter -> assemble -> net -> deprocess)
    cert = favlib.min_filter((st.last_mask().float() / 255).contiguous(), 7)
    flow_lua = torch.stack([T(bw, cuda)[..., 1], T(bw, cuda)[..., 0]]).contiguous()
    warped = favlib.warp(o0, flow_lua)
    in7 = favlib.assemble((T(f1, cuda).permute(2, 0, 1).float() / 255).contiguous(), warped, cert)
    raw = net.forward(in7)
    mean = torch.tensor([103.939, 116.779, 123.68], device=cuda).view(3, 1, 1)
    dep = ((raw + mean) / 255).flip(0)
    assert (dep - o1).abs().max().item() <= 2e-5
    # a centre crop agrees with the oracle run on the crop's receptive field?  No: InstanceNorm is global.
    # Instead: the first-frame result is independent of the (unused) prior -- zero prior, zero mask (core.lua:133-138)
    in7_first = favlib.assemble((T(f0, cuda).permute(2, 0, 1).float() / 255).contiguous())
    dep0 = ((net.forward(in7_first) + mean) / 255).flip(0)
    assert (dep0 - o0).abs().max().item() <= 2e-5
    assert u1.shape == (h, w, 3)


def test_canonical_640x360_frame_vs_oracle(favlib, oracle, cuda, canonical):
    """BASELINE config 2 geometry: one recurrent step against the oracle (~6 s of CPU)."""
    h, w = 360, 640
    layers = _layers(canonical)
    frames, bws, fws = _clip(h, w, 2, 50)
    net = favlib.Net(canonical, 0)
    st = favlib.Stream(net, h, w)
    oracle.set_threads(len(os.sched_getaffinity(0)))        # a real-size problem: use the whole host (conftest caps the team at 16)
    o0, _ = st.first_frame(T(frames[0], cuda))
    o1, u1 = st.next_frame_flow(T(frames[1], cuda), T(bws[1], cuda), T(fws[1], cuda), want_u8=True)
    ref = oracle.Stylizer(layers)
    r0 = ref.first(_f01(frames[0]))
    assert np.abs(o0.cpu().numpy() - r0).max() <= 2e-4
    ref.last = o0.cpu().numpy()
    mask = oracle.consistency(bws[1], fws[1])
    r1 = ref.next(_f01(frames[1]), bws[1], mask.astype(np.float32) / np.float32(255))
    assert np.abs(o1.cpu().numpy() - r1).max() <= 2e-4
    assert psnr8(u1.cpu().numpy(), oracle.to_u8_hwc(r1)) >= 50.0
    oracle.set_threads(min(16, len(os.sched_getaffinity(0))))


def test_config2_640x360_32_frames_reference_masks(favlib, oracle, cuda, canonical):
    """BASELINE config 2 AS WRITTEN: 640x360 x 32 frames, precomputed .flo + occlusion masks -- the masks are files written by the
    REFERENCE's own consistencyChecker binary (oracle/_ref, 4-argument call of makeOptFlow_deepflow.sh:59-60), consumed through the
    certainty path.  BASELINE.md section 4: teacher-forced per frame (gated on ALL 32 frames: <= 2e-4 de-processed, >= 50 dB) AND
    free-running over the full clip (REPORTED: gpurun_out/parity_c2.json / profiles/r03_parity_c2.json).  The free-running distance
    is not gateable over a whole clip with the synthetic random-init weights: the recurrent map frame -> frame amplifies ANY
    perturbation ~3.1x per frame (measured GPU-vs-oracle AND oracle-vs-perturbed-oracle, profiles/r03_parity_sensitivity_control_c2.json),
    so 6e-7 rms of rounding differences at frame 1 saturates around frame 13 on any two implementations, the oracle and itself
    included.  Gated here: the first four free-running frames (<= 1e-3, as test_stream_vs_oracle_recurrent) and that the growth is
    no faster than that sensitivity explains (<= 6x per frame)."""
    if not os.path.exists(oracle.REF_CHECKER):
        pytest.skip("oracle/_ref/consistencyChecker not built (needs /root/reference at build time)")
    import json, sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import parity_clip
    try:
        res = parity_clip.run_clip(favlib, oracle, canonical, 360, 640, 32, mode="cert", seed=2000, pool=8,
                                   threads=parity_clip.effective_cpus(), log=lambda s: None)
    finally:
        oracle.set_threads(min(16, len(os.sched_getaffinity(0))))
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "parity_c2.json"), "w") as f:
            json.dump(res, f, indent=1)
    print({k: v for k, v in res.items() if k != "per_frame"})
    assert res["frames_compared"] == 32
    assert res["teacher_max_abs_worst"] <= 2e-4, res["teacher_max_abs_worst"]
    assert res["teacher_psnr8_db_min"] >= 50.0
    rows = res["per_frame"]
    assert max(r["free_max_abs"] for r in rows[:4]) <= 1e-3, [r["free_max_abs"] for r in rows[:4]]
    growth = [rows[k + 1]["free_rms"] / rows[k]["free_rms"] for k in range(8)]
    assert max(growth) <= 6.0, growth
    assert 5.0 < np.mean([r["reliable_pct"] for r in res["per_frame"][1:]]) < 99.0      # the masks gate a real share of the prior


@pytest.fixture(scope="module")
def contractive(tmp_path_factory):
    """the canonical architecture with the first convolution's weights on the prior + certainty channels scaled by 0.05
    (t7.build_model, recurrent_gain): frame -> frame contracts (CPU control: tests/test_cpu_oracle.py), so the whole clip is gateable"""
    p = str(tmp_path_factory.mktemp("m") / "contractive.t7")
    t7.make_synthetic_checkpoint(p, seed=1234, recurrent_gain=parity_gain())
    return p


def parity_gain():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import parity_clip
    return parity_clip.CONTRACTIVE_GAIN


def test_config2_free_running_whole_clip_contractive_weights(favlib, oracle, cuda, contractive):
    """BASELINE.md section 4's SECOND gate -- free-running over the full clip -- as a gate: config 2 (640x360 x 32 frames, masks =
    files of the reference's own checker binary) on the contractive synthetic checkpoint.  The GPU chain and the oracle chain run
    independently from frame 1; every one of the 32 frames within 2e-4 (de-processed) and 50 dB.  Teacher-forced numbers of the same
    run are gated as well (they use the same oracle calls' inputs, so they cost a second oracle frame each)."""
    if not os.path.exists(oracle.REF_CHECKER):
        pytest.skip("oracle/_ref/consistencyChecker not built (needs /root/reference at build time)")
    import json
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import parity_clip
    try:
        res = parity_clip.run_clip(favlib, oracle, contractive, 360, 640, 32, mode="cert", seed=2000, pool=8, teacher=False,
                                   threads=parity_clip.effective_cpus(), log=lambda s: None)
    finally:
        oracle.set_threads(min(16, len(os.sched_getaffinity(0))))
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "parity_c2_freerun_contractive.json"), "w") as f:
            json.dump(res, f, indent=1)
    print({k: v for k, v in res.items() if k != "per_frame"})
    assert res["frames_compared"] == 32
    # gates at what DESIGN.md states for the F(4x4) build (measured 7.2e-6 / 86.6 dB), not at BASELINE.md's 2e-4 / 50 dB ceiling
    assert res["free_max_abs_worst"] <= 5e-5, [r["free_max_abs"] for r in res["per_frame"]]
    assert res["free_psnr8_db_min"] >= 75.0, res["free_psnr8_db_min"]
    # the recurrent path is live: the prior moves the output by far more than the gate (a dead prior would make the gate vacuous)
    assert res["prior_influence_max_abs"] > 50 * 2e-4, res["prior_influence_max_abs"]


def test_unit_gain_free_run_divergence_against_a_perturbed_oracle_control(favlib, oracle, cuda, canonical):
    """The whole-clip free-running gate above runs on a contractive checkpoint (recurrent gain 0.05): errors in the recurrent path --
    warp, prior, mask, the F(4x4) error fed back through the prior -- are damped 20x there.  On the UNIT-gain random-init checkpoint
    frame -> frame is an expanding map (any perturbation grows ~3x per frame), so an absolute gate is meaningless -- but a RELATIVE one is
    not: the oracle chain is run twice, clean and with its first output perturbed by Gaussian noise of the GPU's own first-frame rms
    error; the GPU chain must not diverge from the clean oracle chain faster than that control does (factor 10: the control's noise
    is white, the GPU's error is not).  A regression in the recurrent path (a wrong prior, a mask off by a pixel) shows as a
    divergence orders of magnitude above the control from the first recurrent frame on."""
    h, w, n = 256, 256, 6
    layers = _layers(canonical)
    frames, bws, fws = _clip(h, w, n, 95)
    masks = [None] + [oracle.consistency(bws[i], fws[i]) for i in range(1, n)]
    net = favlib.Net(canonical, 0)
    st = favlib.Stream(net, h, w)
    gpu = [st.first_frame(T(frames[0], cuda))[0].cpu().numpy()]
    for i in range(1, n):
        gpu.append(st.next_frame_cert(T(frames[i], cuda), T(bws[i], cuda), T(masks[i], cuda))[0].cpu().numpy())
    net.check()
    clean, ctrl = oracle.Stylizer(layers), oracle.Stylizer(layers)
    a0 = clean.first(_f01(frames[0])); ctrl.first(_f01(frames[0]))
    rms0 = float(np.sqrt(np.mean((gpu[0] - a0).astype(np.float64) ** 2)))
    assert 0 < rms0 < 1e-5, rms0
    ctrl.last = (a0 + np.random.default_rng(7).standard_normal(a0.shape).astype(np.float32) * np.float32(rms0)).astype(np.float32)
    rows = []
    for i in range(1, n):
        c01 = masks[i].astype(np.float32) / np.float32(255)
        a = clean.next(_f01(frames[i]), bws[i], c01); b = ctrl.next(_f01(frames[i]), bws[i], c01)
        dg = float(np.sqrt(np.mean((gpu[i] - a).astype(np.float64) ** 2))); dc = float(np.sqrt(np.mean((b - a).astype(np.float64) ** 2)))
        rows.append((i + 1, dg, dc))
    print("unit-gain free run, rms divergence from the clean oracle chain (frame, GPU, perturbed-oracle control): " +
          "  ".join("%d: %.2e / %.2e" % r for r in rows))
    for (fr, dg, dc) in rows:
        assert dg <= 10.0 * dc + 1e-6, (fr, dg, dc)
    assert rows[-1][2] > rows[0][2]                 # the map really expands: the control's divergence grows over the clip


def test_canonical_1280x720_recurrent_step_vs_oracle(favlib, oracle, cuda, canonical, poison):
    """BASELINE config 3 -- the configuration bench.py's headline number is quoted on: one recurrent step
    (fused consistency check + min filter + warp + assembly + network + de-process, core.lua:161-180) at 1280x720 against the
    oracle (~6 s of CPU).  Gates of BASELINE.md section 4: mask bit-exact, max-abs <= 2e-4 de-processed (= 5e-2 in the
    150*tanh space), 8-bit PSNR >= 50 dB."""
    h, w = 720, 1280
    layers = _layers(canonical)
    frames, bws, fws = _clip(h, w, 2, 70)
    net = favlib.Net(canonical, 0)
    st = favlib.Stream(net, h, w)
    oracle.set_threads(len(os.sched_getaffinity(0)))
    try:
        poison()                                    # (stale NaN patterns in LDS / HBM must not matter: see test_results_do_not_depend_...)
        o0, _ = st.first_frame(T(frames[0], cuda))
        import torch
        torch.cuda.synchronize(); poison()
        o1, u1 = st.next_frame_flow(T(frames[1], cuda), T(bws[1], cuda), T(fws[1], cuda), want_u8=True)
        net.check()
        mask = oracle.consistency(bws[1], fws[1])
        assert np.array_equal(st.last_mask().cpu().numpy(), mask)
        ref = oracle.Stylizer(layers)
        ref.last = o0.cpu().numpy()                 # teacher-forced: the step under test is the recurrent one
        r1 = ref.next(_f01(frames[1]), bws[1], mask.astype(np.float32) / np.float32(255))
        err = float(np.abs(o1.cpu().numpy() - r1).max())
        ps = psnr8(u1.cpu().numpy(), oracle.to_u8_hwc(r1))
        print("1280x720 recurrent step: max-abs %.3e (de-processed), 8-bit PSNR %.1f dB" % (err, ps))
        assert err <= 5e-5, err                     # (measured 9.0e-6 / 85.2 dB with F(4x4), 4.6e-6 / 87.8 dB with F(2x2); BASELINE.md's ceiling: 2e-4 / 50 dB)
        assert ps >= 75.0, ps
        assert np.abs(r1).std() > 0.05              # not a saturated / trivial frame
    finally:
        oracle.set_threads(min(16, len(os.sched_getaffinity(0))))


def test_config1_256x256_two_frames_vs_oracle(favlib, oracle, cuda, canonical):
    """BASELINE config 1 geometry (2-frame 256x256 clip, precomputed .flo + reliable_*.pgm, the reference's plumbing case):
    both frames free-running against the oracle, canonical architecture."""
    h, w = 256, 256
    layers = _layers(canonical)
    frames, bws, fws = _clip(h, w, 2, 90)
    mask = oracle.consistency(bws[1], fws[1])       # what consistencyChecker writes to reliable_2_1.pgm
    net = favlib.Net(canonical, 0)
    st = favlib.Stream(net, h, w)
    o0, u0 = st.first_frame(T(frames[0], cuda), want_u8=True)
    o1, u1 = st.next_frame_cert(T(frames[1], cuda), T(bws[1], cuda), T(mask, cuda), want_u8=True)
    net.check()
    ref = oracle.Stylizer(layers)
    r0 = ref.first(_f01(frames[0]))
    r1 = ref.next(_f01(frames[1]), bws[1], mask.astype(np.float32) / np.float32(255))     # free-running: the oracle's own frame 1 is the prior
    assert np.abs(o0.cpu().numpy() - r0).max() <= 2e-4
    assert np.abs(o1.cpu().numpy() - r1).max() <= 4e-4       # two frames of accumulated rounding
    assert psnr8(u0.cpu().numpy(), oracle.to_u8_hwc(r0)) >= 50.0 and psnr8(u1.cpu().numpy(), oracle.to_u8_hwc(r1)) >= 50.0


def test_consistency_extreme_flows(favlib, oracle, cuda):
    """Flows that leave the int range (the Middlebury 'unknown' marker 1e10, +-inf, NaN): the reference's (int)floor() yields
    INT_MIN there (cvttsd2si) and the pixel is written 0; the GPU's saturating conversion must not turn that into an
    out-of-bounds gather."""
    h, w = 40, 56
    bw = synth.backward_flow(h, w, 5); fw = synth.forward_flow_from_backward(bw, 6)
    bw = bw.copy(); fw = fw.copy()
    vals = [1e10, -1e10, np.inf, -np.inf, np.nan, 2147483648.0, 2147483520.0, -2147483904.0, 3e9]
    rng = np.random.default_rng(1)
    for k, v in enumerate(vals * 3):
        y, x, c = int(rng.integers(0, h)), int(rng.integers(0, w)), int(rng.integers(0, 2))
        bw[y, x, c] = np.float32(v)
    fw[3, 4, 0] = np.float32(1e10); fw[7, 9, 1] = np.float32(np.nan)      # poisoned samples of flow2: only the values propagate
    got = favlib.consistency(T(bw, cuda), T(fw, cuda)).cpu().numpy()
    with np.errstate(all="ignore"):
        want = oracle.consistency(bw, fw)
    assert np.array_equal(got, want)


def test_two_networks_on_two_hip_streams_are_serialised(favlib, cuda, canonical):
    """the persistent / stream-K grids must not overlap on a device: forwards enqueued on different HIP streams (two networks, one
    process) are ordered by the library itself (an event per forward, a wait on a stream switch) -- same results as on one stream,
    no hand-off time-out (fav.h, concurrency note)"""
    import torch
    h, w = 360, 640
    rng = np.random.default_rng(3)
    xa = T((rng.standard_normal((7, h, w)) * 60).astype(np.float32), cuda)
    xb = T((rng.standard_normal((7, h, w)) * 60).astype(np.float32), cuda)
    na, nb = favlib.Net(canonical, 0), favlib.Net(canonical, 0)
    ra, rb = na.forward(xa).clone(), nb.forward(xb).clone()                  # one stream: the reference results
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for it in range(6):                                                      # interleaved on two streams, nothing synchronised in between
        with torch.cuda.stream(sa):
            oa = na.forward(xa)
        with torch.cuda.stream(sb):
            ob = nb.forward(xb)
        outs.append((oa, ob))
    torch.cuda.synchronize()
    na.check(); nb.check()
    for oa, ob in outs:
        assert torch.equal(oa, ra) and torch.equal(ob, rb)


def test_bf16_fast_mode_with_structure_lookahead(favlib, oracle, cuda, canonical):
    """with look-ahead side queues active (4-argument masks), EVERY stream-K kernel -- the halo-resident 3x3 kernel of the
    bf16 fast mode included -- takes the data-parallel descriptor; the frames equal those computed without look-ahead"""
    import torch
    h, w = 360, 640
    frames, bws, fws = _clip(h, w, 3, 60)
    outs = []
    for look in (False, True):
        net = favlib.Net(canonical, 0); net.set_precision(True)
        st = favlib.Stream(net, h, w)
        st.first_frame(T(frames[0], cuda))
        d = [(T(frames[i], cuda), T(bws[i], cuda), T(fws[i], cuda)) for i in (1, 2)]
        if look:
            st.prefetch_mask(*d[0], use_structure=True); st.prefetch_mask(*d[1], use_structure=True)
        o1, _ = st.next_frame_flow(*d[0], use_structure=True)
        o2, _ = st.next_frame_flow(*d[1], use_structure=True)
        torch.cuda.synchronize(); net.check()
        outs.append((o1.cpu().numpy(), o2.cpu().numpy(), st.last_mask().cpu().numpy()))
    assert np.array_equal(outs[0][2], outs[1][2])
    # the two grid forms add in different orders; in THIS mode a 1e-7 difference of an activation can flip its bf16 rounding (2^-8
    # relative), so the two runs differ by about the mode's own error against fp32 (measured 1.4e-2 max-abs): gate as the mode is
    # gated (>= 45 dB on the 8-bit frames)
    for k in (0, 1):
        a8, b8 = oracle.to_u8_hwc(outs[0][k]), oracle.to_u8_hwc(outs[1][k])
        assert psnr8(a8, b8) >= 45.0, psnr8(a8, b8)


def test_shared_device_mode_matches_oracle(favlib, oracle, cuda, canonical):
    """fav_net_set_shared_device: data-parallel convolution grids (one block per tile, no stream-K hand-off, no co-residency
    assumption) -- the mode for a GPU this process does not own, and what fav_net_check falls back to after a timed-out
    hand-off.  Same arithmetic per tile up to the order in which split tiles are summed: same tolerance as the default mode."""
    layers = _layers(canonical)
    h, w = 96, 128
    frames, bws, fws = _clip(h, w, 2, 120)
    net = favlib.Net(canonical, 0)
    net.set_shared_device(True)
    st = favlib.Stream(net, h, w)
    o0, _ = st.first_frame(T(frames[0], cuda))
    o1, _ = st.next_frame_flow(T(frames[1], cuda), T(bws[1], cuda), T(fws[1], cuda))
    net.check()
    ref = oracle.Stylizer(layers)
    r0 = ref.first(_f01(frames[0]))
    assert np.abs(o0.cpu().numpy() - r0).max() <= 2e-4
    ref.last = o0.cpu().numpy()
    m = oracle.consistency(bws[1], fws[1])
    r1 = ref.next(_f01(frames[1]), bws[1], m.astype(np.float32) / np.float32(255))
    assert np.abs(o1.cpu().numpy() - r1).max() <= 2e-4
    # and at a size where the default mode splits tiles between blocks (more tiles than CUs): both modes agree closely
    net2 = favlib.Net(canonical, 0)
    x = (np.random.default_rng(7).standard_normal((7, 360, 640)) * 40).astype(np.float32)
    a = net2.forward(T(x, cuda)).cpu().numpy()
    net2.set_shared_device(True)
    b = net2.forward(T(x, cuda)).cpu().numpy()
    assert np.abs(a - b).max() <= 2e-2            # 150*tanh space; summation order of split tiles only


def test_results_do_not_depend_on_stale_lds_or_memory(favlib, oracle, cuda, golden_dir, canonical, poison):
    """A GPU handed over by another tenant holds arbitrary bit patterns in LDS and HBM.  NaN-poison every CU's LDS and a gigabyte
    of device memory (tests/util/lds_poison.hip, compiled here), then run the tiny and the canonical network: an operand the
    kernels never wrote, multiplied by a zero weight, would turn into NaN and -- through the InstanceNorm statistics and the ReLU's
    fmaxf -- into a finite but wrong frame (seen once on a fresh box with the dense-K first layer's unpaired tap)."""
    for path, (h, w) in ((os.path.join(golden_dir, "tiny_model.t7"), (48, 64)), (canonical, (64, 96))):
        layers = _layers(path)
        frames, bws, fws = _clip(h, w, 2, 140)
        poison(3)
        net = favlib.Net(path, 0)
        st = favlib.Stream(net, h, w)
        poison(3)
        o0, _ = st.first_frame(T(frames[0], cuda))
        import torch
        torch.cuda.synchronize()
        poison(3)
        o1, _ = st.next_frame_flow(T(frames[1], cuda), T(bws[1], cuda), T(fws[1], cuda))
        ref = oracle.Stylizer(layers)
        r0 = ref.first(_f01(frames[0]))
        assert np.abs(o0.cpu().numpy() - r0).max() <= 2e-4
        ref.last = o0.cpu().numpy()
        m = oracle.consistency(bws[1], fws[1])
        r1 = ref.next(_f01(frames[1]), bws[1], m.astype(np.float32) / np.float32(255))
        assert np.abs(o1.cpu().numpy() - r1).max() <= 2e-4
        # the structure-aware (4-argument) mask: order-preserving scans through LDS
        poison(2)
        got = favlib.consistency(T(bws[1], cuda), T(fws[1], cuda), T(frames[1], cuda)).cpu().numpy()
        assert np.array_equal(got, oracle.consistency(bws[1], fws[1], frames[1]))


def test_temporal_loss_vs_oracle(favlib, oracle, cuda):
    """the temporal-consistency number of -evaluate (fast_artistic_video.lua:128-151)."""
    h, w = 90, 130
    rng = np.random.default_rng(3)
    prev = rng.random((3, h, w)).astype(np.float32) * 1.2 - 0.1
    bw = synth.backward_flow(h, w, 5)
    cert = ((rng.random((h, w)) > 0.3) * 255).astype(np.uint8)
    cur = oracle.warp(prev, oracle.flo_to_lua(bw)) + rng.normal(0, 0.02, (3, h, w)).astype(np.float32)
    want = oracle.temporal_loss(prev, cur, bw, cert.astype(np.float32) / np.float32(255))
    got = favlib.temporal_loss(T(prev, cuda), T(cur, cuda), T(bw, cuda), T(cert, cuda))
    assert want > 1e-5 and abs(got - want) <= 1e-5 * want
    assert favlib.temporal_loss(T(prev, cuda), T(oracle.warp(prev, oracle.flo_to_lua(bw)), cuda), T(bw, cuda), T(cert, cuda)) <= 1e-12


@pytest.mark.parametrize("arch,size", [
    ("c3s1-32,U2,c3s1-64,c3s1-128,c3s1-128,c9s1-3", (37, 45)),      # halo<64> with TWO pending stages + x2 upsample; halo<128> with 64 and 128 input channels
    ("c3s1-64,c3s1-64,U2,c3s1-128,c9s1-3", (50, 26)),               # ragged tiles (rows % 8, columns % 32 != 0), fewer work units than CUs
    ("c9s1-32,d64,R64,R64,U2,c3s1-32,c9s1-3", (64, 72)),            # residual blocks at 64 channels (halo<64>, shave), reflection pad 8
], ids=["two-stage-ups", "ragged", "res64"])
def test_halo_kernel_variants_in_networks(favlib, oracle, cuda, tmp_path, arch, size):
    """Networks built to reach every instantiation / edge of the halo-resident 3x3 kernel (conv3_halo_kernel<BN,S2>)."""
    p = str(tmp_path / "m.t7")
    t7.make_synthetic_checkpoint(p, arch=arch, seed=77)
    layers = _layers(p)
    net = favlib.Net(p, 0)
    h, w = size
    x = (np.random.default_rng(11).standard_normal((7, h, w)) * 60).astype(np.float32)
    ref = oracle.net_forward(layers, x)
    got = net.forward(T(x, cuda)).cpu().numpy()
    assert got.shape == ref.shape
    err = np.abs(got - ref).max()
    assert err <= 5e-2, err
    assert np.abs(ref).std() > 5


@pytest.mark.parametrize("arch,size", [
    ("c3s1-128,R128,R128,c9s1-3", (45, 61)),                 # 2 blocks: inputs with and without a pending InstanceNorm; 6 x 4 / 5 x 4 units, ragged both ways
    ("c9s1-32,d64,d128,R128,U2,c3s1-64,U2,c9s1-3", (72, 136)),  # one unit row, fewer units than CUs
    ("c3s1-128,R128,R128,R128,c9s1-3", (150, 330)),           # more units than CUs (persistent blocks take two)
    ("c3s1-128,R128,c9s1-3", (138, 258)),                     # 272 and 256 units on 256 CUs: the thin second round runs as quarter units
], ids=["two-blocks", "canonical-small", "multi-round", "quarter-units"])
def test_winograd_residual_layers_in_networks(favlib, oracle, cuda, tmp_path, arch, size):
    """The residual 128 -> 128 convolutions run as Winograd F(2x2,3x3) (conv3_wino_kernel; lane-level restatement in
    tests/test_cpu_wino.py): against the oracle's direct convolution, same tolerance as the direct-form kernels."""
    p = str(tmp_path / "m.t7")
    t7.make_synthetic_checkpoint(p, arch=arch, seed=78)
    layers = _layers(p)
    net = favlib.Net(p, 0)
    h, w = size
    x = (np.random.default_rng(12).standard_normal((7, h, w)) * 60).astype(np.float32)
    ref = oracle.net_forward(layers, x)
    got = net.forward(T(x, cuda)).cpu().numpy()
    net.check()
    assert got.shape == ref.shape
    err = np.abs(got - ref).max()
    assert err <= 5e-2, err
    assert np.abs(ref).std() > 5


@pytest.mark.parametrize("arch,size", [
    ("c3s1-32,U2,c9s1-3", (37, 45)),           # 32 input channels, ragged tiles, one pending InstanceNorm
    ("c3s1-64,U2,c9s1-3", (41, 70)),           # 64 input channels, two column tiles (120 + 20), six row tiles
    ("c3s1-32,U2,c3s1-64,U2,c9s1-3", (24, 31)),  # the canonical tail: upsample, conv, upsample, last conv
], ids=["cin32", "cin64", "tail"])
def test_last_layer_on_upsampled_input(favlib, oracle, cuda, tmp_path, arch, size):
    """U2 + c9s1-3 (models_video.lua:129-136) runs on the PHYSICAL pixels with merged ky weight slices (conv_rowfold_up2_kernel)
    instead of the four-fold repeated upsampled image: same result as the oracle's upsample-then-convolve."""
    p = str(tmp_path / "m.t7")
    t7.make_synthetic_checkpoint(p, arch=arch, seed=79)
    layers = _layers(p)
    net = favlib.Net(p, 0)
    h, w = size
    x = (np.random.default_rng(13).standard_normal((7, h, w)) * 60).astype(np.float32)
    ref = oracle.net_forward(layers, x)
    got = net.forward(T(x, cuda)).cpu().numpy()
    assert got.shape == ref.shape
    err = np.abs(got - ref).max()
    assert err <= 5e-2, err
    assert np.abs(ref).std() > 5


@pytest.mark.parametrize("arch,size", [
    ("c3s1-64,R64,U2,c3s1-64,c9s1-3", (29, 43)),                    # 64 input channels, 25 x 39 physical pixels: ragged tiles both ways
    ("c9s1-32,d64,d128,R128,U2,c3s1-64,U2,c9s1-3", (88, 152)),      # the canonical tail behind one residual block (128 input channels)
], ids=["cin64-ragged", "canonical-tail"])
def test_conv_on_upsampled_join_in_networks(favlib, oracle, cuda, tmp_path, arch, size):
    """U2 + c3s1-64 behind a residual join runs as four 2x2 convolutions with merged weights on the physical pixels
    (conv3_up2_kernel, csrc/up2_pack.h): same result as the oracle's upsample-then-convolve."""
    p = str(tmp_path / "m.t7")
    t7.make_synthetic_checkpoint(p, arch=arch, seed=80)
    layers = _layers(p)
    net = favlib.Net(p, 0)
    h, w = size
    x = (np.random.default_rng(14).standard_normal((7, h, w)) * 60).astype(np.float32)
    ref = oracle.net_forward(layers, x)
    got = net.forward(T(x, cuda)).cpu().numpy()
    assert got.shape == ref.shape
    err = np.abs(got - ref).max()
    assert err <= 5e-2, err
    assert np.abs(ref).std() > 5


@pytest.mark.parametrize("arch,size", [
    ("c9s1-32,d64,d128,c9s1-3", (92, 140)),        # canonical head: 46 x 70 -> 23 x 35 (ragged 4-row tiles) -> 12 x 18 (3-row tiles)
    ("c3s1-64,d64,c9s1-3", (54, 66)),              # 64 input channels: four 16-channel chunks per tile; 27 x 33 outputs (one column past a tile)
    ("c3s1-128,d128,c9s1-3", (34, 134)),           # 128 input channels, 128 outputs: eight chunks, 17 x 67 outputs = 6 x 3 tiles with ragged edges
    ("c3s1-32,d128,c9s1-3", (22, 62)),             # 32 input channels into 128 outputs: two chunks per tile
], ids=["canonical-head", "cin64-d64", "cin128-d128", "cin32-d128"])
def test_stride2_layers_in_networks(favlib, oracle, cuda, tmp_path, arch, size):
    """d64 / d128 (3x3, stride 2, zero padding 1, models_video.lua:88-92) run on conv3s2w_kernel (csrc/kernels_s2.hip): border tiles,
    ragged right / bottom tiles and chunk counts other than the canonical network's, against the oracle's direct form."""
    p = str(tmp_path / "m.t7")
    t7.make_synthetic_checkpoint(p, arch=arch, seed=81)
    layers = _layers(p)
    net = favlib.Net(p, 0)
    h, w = size
    x = (np.random.default_rng(15).standard_normal((7, h, w)) * 60).astype(np.float32)
    ref = oracle.net_forward(layers, x)
    got = net.forward(T(x, cuda)).cpu().numpy()
    net.check()
    assert got.shape == ref.shape
    err = np.abs(got - ref).max()
    assert err <= 5e-2, err
    assert np.abs(ref).std() > 5


# ---- checkpoints with more filters than the canonical one (round 5) ----------------------------------------------------------------
# models_video.lua:55-140 builds the network from ANY architecture string and the published VR checkpoints "have more filters"
# (README.md:141, models/download_models_vr.sh:1-5).  Their layers run on the same minimal-filtering kernels as the canonical
# network's, in groups of output channels (F(4x4) residual 3x3: groups of 128; stride-2: 128; U2 + 3x3: 64; first layer: 32; the last
# layer: passes of 64 input channels) -- kernel ids as fav_net::timed_conv reports them through fav_net_profile_read_host
WIDE2 = "c9s1-64,d128,d256,R256,R256,R256,R256,R256,U2,c3s1-128,U2,c9s1-3"          # every filter count doubled
WIDE15 = "c9s1-48,d96,d192,R192,R192,R192,R192,R192,U2,c3s1-96,U2,c9s1-3"          # x1.5: runs zero-padded to 64 / 128 / 256
WIDE_KERNEL_IDS = [16, 700 + 128, 700 + 256] + [600 + 256] * 10 + [500 + 128, 1]


def _wide_forward(favlib, cuda, path, x):
    net = favlib.Net(path, 0)
    net.profile_enable(True)
    got = net.forward(T(x, cuda)).cpu().numpy()
    ids = [kid for (ms, n, macs, kid) in net.profile_read()]
    net.profile_enable(False)
    net.check()
    return net, got, ids


@pytest.mark.parametrize("arch,size", [(WIDE2, (88, 120)), (WIDE2, (150, 210)), (WIDE15, (90, 122)), (WIDE2, (720, 1280)), (WIDE15, (720, 1280))],
                         ids=["2x-88x120", "2x-150x210", "1.5x-90x122", "2x-720p", "1.5x-720p"])
def test_networks_with_more_filters_stay_on_the_fast_kernels(favlib, oracle, cuda, tmp_path, arch, size):
    """Doubled and x1.5 filter counts against the oracle's direct form, small ragged sizes and 1280x720; every convolution must have
    run on the kernel the canonical network uses for that layer (no generic implicit-GEMM fallback)."""
    p = str(tmp_path / "wide.t7")
    t7.make_synthetic_checkpoint(p, arch=arch, seed=91)
    layers = _layers(p)
    h, w = size
    x = (np.random.default_rng(16).standard_normal((7, h, w)) * 60).astype(np.float32)
    big = h * w > 500000
    if big: oracle.set_threads(len(os.sched_getaffinity(0)))
    try:
        ref = oracle.net_forward(layers, x)
    finally:
        if big: oracle.set_threads(min(16, len(os.sched_getaffinity(0))))
    net, got, ids = _wide_forward(favlib, cuda, p, x)
    assert ids == WIDE_KERNEL_IDS, ids
    assert net.describe() == favlib.describe_layers(layers)
    assert got.shape == ref.shape
    err = np.abs(got - ref).max()
    print("more filters %s at %dx%d: max-abs %.3e (150*tanh space)" % (arch.split(",")[0], h, w, err))
    assert err <= 2e-2, err
    assert np.abs(ref).std() > 10
    again = net.forward(T(x, cuda)).cpu().numpy()
    assert np.array_equal(got, again)


def test_stream_k_shares_with_more_filters(favlib, cuda, tmp_path):
    """The F(4x4) kernel's stream-K mode with two groups of 128 filters (group-major sequence, shares that cross from one group into the
    next): FAV_W4_GRID forces many shares at a small size; same numbers as whole units, same bits from run to run."""
    import subprocess, sys
    p = str(tmp_path / "wide.t7")
    t7.make_synthetic_checkpoint(p, arch="c3s1-128,d256,R256,R256,U2,c9s1-3", seed=92)
    x = (np.random.default_rng(17).standard_normal((7, 118, 150)) * 60).astype(np.float32)
    np.save(tmp_path / "x.npy", x)
    child = ("import sys, numpy as np, torch; sys.path.insert(0, %r); import fav_amd\n"
             "x = torch.from_numpy(np.load(%r)).cuda(); net = fav_amd.Net(%r, 0)\n"
             "a = net.forward(x).cpu().numpy(); b = net.forward(x).cpu().numpy()\n"
             "assert np.array_equal(a, b)\n"
             "np.save(sys.argv[1], a)\n"
             % (os.path.join(ROOT, "fast-artistic-videos_amd", "python"), str(tmp_path / "x.npy"), p))
    subprocess.check_call([sys.executable, "-c", child, str(tmp_path / "whole.npy")], env=dict(DIAG_ENV, FAV_W4_NO_STREAM="1", FAV_W4_GRID="7"), timeout=300)
    subprocess.check_call([sys.executable, "-c", child, str(tmp_path / "stream.npy")], env=dict(DIAG_ENV, FAV_W4_GRID="7"), timeout=300)
    whole, stream = np.load(tmp_path / "whole.npy"), np.load(tmp_path / "stream.npy")
    assert np.isfinite(stream).all()
    d = np.abs(stream - whole).max()
    print("more filters, stream-K vs whole units: max-abs %.3e" % d)
    assert d <= 5e-3 and not np.array_equal(stream, whole), d


@pytest.mark.parametrize("arch,size,ids", [
    ("c9s1-64,c9s1-3", (40, 70), [16, None]),                                   # first layer alone: two groups of 32, ragged tiles
    ("c9s1-96,d64,c9s1-3", (36, 52), [16, 764, None]),                          # 96 filters: padded to 128 = four groups
    ("c3s1-64,d256,c9s1-3", (38, 134), [None, 956, None]),                      # stride 2 into 256: 64 input channels, ragged tiles
    ("c3s1-128,R128,U2,c3s1-128,c9s1-3", (29, 43), [None, None, None, 628, None]),               # U2 + 3x3 into 128: two groups of 64 (behind a join, as the builder's strings have it: ONE pending normalisation)
    ("c3s1-128,R128,U2,c3s1-256,c3s1-16,c9s1-3", (24, 40), [None, None, None, 756, None, None]),   # ... into 256: four groups
    ("c3s1-128,U2,c9s1-3", (41, 70), [None, 1]),                                # last layer on 128 channels: two passes of 64
    ("c3s1-256,U2,c9s1-3", (26, 37), [None, 1]),                                # ... on 256: four passes
], ids=["first64", "first96", "d256", "up2-128", "up2-256", "last128", "last256"])
def test_layers_with_more_filters_one_at_a_time(favlib, oracle, cuda, tmp_path, arch, size, ids):
    p = str(tmp_path / "m.t7")
    t7.make_synthetic_checkpoint(p, arch=arch, seed=93)
    layers = _layers(p)
    h, w = size
    x = (np.random.default_rng(18).standard_normal((7, h, w)) * 60).astype(np.float32)
    ref = oracle.net_forward(layers, x)
    net, got, kids = _wide_forward(favlib, cuda, p, x)
    assert len(kids) == len(ids) and all(want is None or want == k for want, k in zip(ids, kids)), kids
    assert got.shape == ref.shape
    err = np.abs(got - ref).max()
    assert err <= 2e-2, err
    assert np.abs(ref).std() > 5


def _stress_cases():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import wino4_stress
    return [c[0] for c in wino4_stress.CASES]


@pytest.mark.parametrize("case", _stress_cases())
def test_f4x4_under_stress(favlib, oracle, cuda, tmp_path, case):
    """The F(4x4,3x3) layers outside the comfort zone of rounds 2-4 (N(0, sqrt(2 / fan-in)) weights, 128 input channels, N(0, 60)
    inputs): 1 % weight outliers at 20 sigma, InstanceNorm gammas up to 8, inputs at std 600, 64 and 256 input channels
    (scripts/wino4_stress.py builds the checkpoints; run as a script it reports F(4x4) / F(2x2) / direct side by side ->
    profiles/*wino4_stress*).  Every case stays inside BASELINE.md's 5e-2 (150*tanh space) with a margin of 5."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import wino4_stress
    p = str(tmp_path / "stress.t7")
    x = wino4_stress.write_case(p, case)
    layers = _layers(p)
    ref = oracle.net_forward(layers, x)
    net = favlib.Net(p, 0)
    net.profile_enable(True)
    got = net.forward(T(x, cuda)).cpu().numpy()
    ids = [kid for (ms, n, macs, kid) in net.profile_read()]
    net.check()
    assert any(k in (728, 729, 856) for k in ids), ids          # at least one layer ran on the F(4x4) kernel
    err = np.abs(got - ref).max()
    print("F(4x4) stress %-32s max-abs %.3e (150*tanh space)" % (case, err))
    assert err <= 1e-2, err
    assert np.abs(ref).std() > 20 and (np.abs(ref) > 149.0).mean() < 0.01


def _seq_sum(x):
    return np.add.accumulate(x.astype(np.float32), dtype=np.float32)[-1]     # sequential, one rounding per element


@pytest.mark.parametrize("case", ["uniform", "ties", "mixed", "negatives", "zeros", "all-zero", "tiny", "structure-like"])
def test_sequential_sum_bit_exact(favlib, cuda, case):
    """CMatrix::avg's arithmetic (CMatrix.h:1245-1251) evaluated by the parallel parity-transducer scan: bit-identical to the
    scalar loop, including exact round-to-even ties, binade crossings, and the inputs that force the native fallback."""
    rng = np.random.default_rng(sum(map(ord, case)))
    n = 921600
    if case == "uniform":
        x = rng.random(n, dtype=np.float32)
    elif case == "ties":                      # multiples of 2^-12: once the sum passes 2^12 every few additions is an exact tie
        x = (rng.integers(0, 4096, n) * 2.0 ** -12).astype(np.float32)
    elif case == "mixed":
        x = (10.0 ** rng.uniform(-9, 3, n)).astype(np.float32)
    elif case == "negatives":
        x = rng.random(n, dtype=np.float32); x[rng.random(n) < 0.01] *= -1
    elif case == "zeros":
        x = rng.random(n, dtype=np.float32); x[rng.random(n) < 0.7] = 0; x[:20000] = 0
    elif case == "all-zero":
        x = np.zeros(70000, np.float32)
    elif case == "tiny":
        x = rng.random(37, dtype=np.float32)
    else:                                     # what the checker feeds it: mostly small values, a few near 1
        x = (rng.random(n) ** 6).astype(np.float32)
    got = favlib.sequential_sum(T(x, cuda)).cpu().numpy()[0]
    want = _seq_sum(x)
    assert got.tobytes() == want.tobytes(), (case, got, want)
    if case == "uniform":                     # and it is NOT what a pairwise / fp64 sum gives: the test is sensitive
        assert np.float32(x.astype(np.float64).sum()) != want


def test_bf16_operand_fast_mode(favlib, oracle, cuda, canonical):
    """optional fast mode (bf16 operands in the halo-resident 3x3 convolutions, fp32 accumulation and
    activations).  Checked against an oracle that rounds the same operands, and gated by PSNR against the fp32 oracle."""
    layers = _layers(canonical)
    net = favlib.Net(canonical, 0)
    x = (np.random.default_rng(5).standard_normal((7, 64, 96)) * 60).astype(np.float32)
    ref32 = oracle.net_forward(layers, x)
    ref16 = oracle.net_forward(layers, x, bf16_ops=True)
    got32 = net.forward(T(x, cuda)).cpu().numpy()
    net.set_precision(True)
    got16 = net.forward(T(x, cuda)).cpu().numpy()
    net.set_precision(False)
    again32 = net.forward(T(x, cuda)).cpu().numpy()
    assert np.array_equal(got32, again32)                       # the switch is clean
    assert np.abs(got32 - ref32).max() <= 5e-2
    d = np.abs(got16 - got32).max()
    assert d > 1e-2, "the fast mode must actually change the arithmetic"
    # same rounding in the oracle: what remains is accumulation order and bf16 rounding flips of activations that sit near a
    # tie (the InstanceNorm statistics differ in the last fp32 bits), i.e. a fraction of the mode's own error
    rms = lambda a: float(np.sqrt(np.mean(a.astype(np.float64) ** 2)))
    assert rms(got16 - ref16) <= 0.75 * rms(ref16 - ref32) + 1e-3
    assert 0.5 <= rms(got16 - ref32) / rms(ref16 - ref32) <= 1.5          # and its error against fp32 is the oracle-predicted one
    # quality gate vs the fp32 reference, 8-bit output space (measured: 50 dB)
    to8 = lambda o: oracle.to_u8_hwc(oracle.deprocess(o))
    assert psnr8(to8(got16), to8(ref32)) >= 45.0
