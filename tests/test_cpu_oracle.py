"""CPU suite (-m "not gpu"): pins the oracle against the committed golden vectors (outputs of the
reference's own consistencyChecker), cross-checks the [recalled] Torch7 semantics against PyTorch-CPU,
and checks the host-side pieces of libfav (no compute calls: there is no GPU here)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from fav_amd import synth, t7

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------------------------- mask
@pytest.mark.parametrize("name", ["mask_smooth_64x96.npz", "mask_rand_120x160.npz", "mask_smooth_180x320.npz"])
def test_oracle_mask_matches_reference_golden(oracle, golden_dir, name):
    g = np.load(os.path.join(golden_dir, name))
    m3 = oracle.consistency(g["bw"], g["fw"])
    m4 = oracle.consistency(g["bw"], g["fw"], g["img"])
    assert np.array_equal(m3, g["mask3"]), "3-argument mode must be bit-exact with the reference binary"
    assert np.array_equal(m4, g["mask4"]), "4-argument mode must be bit-exact with the reference binary"
    assert set(np.unique(m3)) <= {0, 255}
    assert (g["mask3"] != g["mask4"]).sum() > 0     # the structure term is exercised


def test_oracle_mask_vs_live_reference_binary(oracle, tmp_path):
    if not os.path.exists(oracle.REF_CHECKER):
        pytest.skip("oracle/_ref/consistencyChecker not built (reference tree absent)")
    for seed, (h, w) in enumerate([(33, 47), (72, 128), (2, 9)]):
        bw = synth.random_flow(h, w, 50 + seed, 2.0); fw = synth.random_flow(h, w, 60 + seed, 2.0)
        img = synth.random_frame(h, w, 70 + seed)
        a, b, i, o = (str(tmp_path / n) for n in ("a.flo", "b.flo", "i.ppm", "o.pgm"))
        oracle.write_flo(a, bw); oracle.write_flo(b, fw); oracle.write_pnm(i, img)
        subprocess.check_call([oracle.REF_CHECKER, a, b, o], stdout=subprocess.DEVNULL)
        assert np.array_equal(oracle.read_pnm(o), oracle.consistency(bw, fw))
        subprocess.check_call([oracle.REF_CHECKER, a, b, o, i], stdout=subprocess.DEVNULL)
        assert np.array_equal(oracle.read_pnm(o), oracle.consistency(bw, fw, img))


def test_oracle_mask_extreme_flows_vs_live_reference_binary(oracle, tmp_path):
    """|flow| >= 2^31, +-inf and NaN (the Middlebury 'unknown flow' marker is 1e10): the reference's (int)floor() gives INT_MIN
    on x86 and the pixel is written 0.  Pins the oracle -- and through it the GPU kernel's float-domain range test
    (tests/test_gpu_parity.py::test_consistency_extreme_flows) -- on the compiled reference for those inputs."""
    if not os.path.exists(oracle.REF_CHECKER):
        pytest.skip("oracle/_ref/consistencyChecker not built (reference tree absent)")
    h, w = 24, 40
    bw = synth.random_flow(h, w, 7, 2.0); fw = synth.random_flow(h, w, 8, 2.0)
    vals = [1e10, -1e10, np.inf, -np.inf, np.nan, 2147483648.0, 2147483520.0, -2147483904.0, 3e9]
    rng = np.random.default_rng(2)
    hit = []
    for v in vals * 2:
        y, x, c = int(rng.integers(0, h)), int(rng.integers(0, w)), int(rng.integers(0, 2))
        bw[y, x, c] = np.float32(v); hit.append((y, x))
    a, b, o = (str(tmp_path / n) for n in ("a.flo", "b.flo", "o.pgm"))
    oracle.write_flo(a, bw); oracle.write_flo(b, fw)
    subprocess.check_call([oracle.REF_CHECKER, a, b, o], stdout=subprocess.DEVNULL)
    ref = oracle.read_pnm(o)
    with np.errstate(all="ignore"):
        assert np.array_equal(ref, oracle.consistency(bw, fw))
    assert all(ref[y, x] == 0 for y, x in hit)


def test_launcher_shards_streams_across_workers(favlib, tmp_path):
    """fav_stylize -streams a,b,c,d,e -gpus 2 -dry_run 1: one worker PROCESS per GPU, stream s -> worker s mod N, %S replaced
    in every path option, per-worker PNG-writer budget = host threads / N.  No device is touched."""
    import json
    exe = os.path.join(ROOT, "fast-artistic-videos_amd", "bin", "fav_stylize")
    names = ["a", "b", "c", "d", "e"]
    base = [exe, "-input_pattern", "in/%S/frame_%05d.ppm", "-flow_pattern", "in/%S/flow/backward_[%d]_{%d}.flo",
            "-forward_flow_pattern", "in/%S/flow/forward_{%d}_[%d].flo", "-output_prefix", "out/%S/out", "-model_vid", "m.t7", "-model_img", "self",
            "-streams", ",".join(names), "-dry_run", "1"]
    for world in (2, 3):
        r = subprocess.run(base + ["-gpus", str(world), "-gpu", "1"], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr
        lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
        ceil = [d["host_ceiling"] for d in lines if "host_ceiling" in d]
        assert len(ceil) == 1 and ceil[0]["host_bound_ceiling_fps"] == pytest.approx(ceil[0]["usable_cpus"] * 1e3 / ceil[0]["host_cpu_ms_per_frame"], rel=1e-3)
        recs = sorted((d for d in lines if "rank" in d), key=lambda d: d["rank"])
        assert [d["rank"] for d in recs] == list(range(world)) and all(d["world"] == world for d in recs)
        assert [d["device"] for d in recs] == [1 + k for k in range(world)]            # devices -gpu .. -gpu + n - 1
        aff = len(os.sched_getaffinity(0)); quota = 1 << 20
        try:        # the writer budget: the worker's share of the affinity mask (workers are pinned) and of the cgroup CPU quota
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max": quota = max(1, -(-int(q) // int(per)))
        except OSError:
            pass
        shares = [aff * (k + 1) // world - aff * k // world for k in range(world)] if aff >= world else [aff] * world
        assert [d["cpus"] for d in recs] == shares                                     # each worker pinned to its contiguous share
        assert [d["writers"] for d in recs] == [max(2, min(32, sh, max(1, quota // world))) for sh in shares]
        for d in recs:
            want = [n for k, n in enumerate(names) if k % world == d["rank"]]
            assert [s["name"] for s in d["streams"]] == want
            for s in d["streams"]:
                assert s["input_pattern"] == f"in/{s['name']}/frame_%05d.ppm" and s["output_prefix"] == f"out/{s['name']}/out"
                assert s["flow_pattern"] == f"in/{s['name']}/flow/backward_[%d]_{{%d}}.flo"
                assert s["forward_flow_pattern"] == f"in/{s['name']}/flow/forward_{{%d}}_[%d].flo"
    # single process (no -gpus): the streams run back to back in this process
    r = subprocess.run(base, capture_output=True, text=True, timeout=60)
    recs = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(recs) == 1 and [s["name"] for s in recs[0]["streams"]] == names and recs[0]["world"] == 1
    # streams that would overwrite each other are refused
    bad = [a if a != "out/%S/out" else "out/out" for a in base]
    r = subprocess.run(bad + ["-gpus", "2"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "%S" in r.stderr


def test_launcher_fails_fast_instead_of_hanging(favlib, golden_dir, tmp_path):
    """`-gpus 2 -model_vid <typo>` used to hang (rank 0 died parsing AFTER the communicator was up, its peers
    blocked in the broadcast).  Now (1) the launcher itself parses the checkpoint(s) before any worker exists, (2) a worker that
    fails takes its siblings down (waitpid(-1) + SIGTERM), (3) the exchange files live in a private mkdtemp directory that the
    launcher removes."""
    import time
    exe = os.path.join(ROOT, "fast-artistic-videos_amd", "bin", "fav_stylize")
    base = [exe, "-input_pattern", "in/%S/f_%05d.ppm", "-flow_pattern", "in/%S/b_[%d]_{%d}.flo", "-forward_flow_pattern", "in/%S/f_{%d}_[%d].flo",
            "-output_prefix", "out/%S/o", "-model_img", "self", "-streams", "a,b", "-gpus", "2"]
    env = dict(os.environ, TMPDIR=str(tmp_path))
    t0 = time.time()
    r = subprocess.run(base + ["-model_vid", str(tmp_path / "typo.t7")], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode != 0 and "typo.t7" in r.stderr and time.time() - t0 < 20
    damaged = tmp_path / "damaged.t7"
    damaged.write_bytes(open(os.path.join(golden_dir, "tiny_model.t7"), "rb").read()[:5000])
    r = subprocess.run(base + ["-model_vid", str(damaged)], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode != 0 and r.stderr.strip()
    # one worker fails while the other blocks: the launcher returns the failing status promptly and leaves nothing behind
    t0 = time.time()
    r = subprocess.run(base + ["-model_vid", "m.t7", "-dry_run", "1"], capture_output=True, text=True, timeout=60,
                       env=dict(env, FAV_TEST_WORKER_FAIL="1"))
    assert r.returncode == 3 and "stopping the other workers" in r.stderr and time.time() - t0 < 20, (r.returncode, r.stderr)
    assert not [n for n in os.listdir(tmp_path) if n.startswith("fav_launch_")]


def test_vr_launcher_shards_videos_across_workers(favlib):
    """fav_stylize_vr shares the launcher (host/fav_launcher.h): 360-degree videos are the unit of sharding (BASELINE config 5)"""
    import json
    exe = os.path.join(ROOT, "fast-artistic-videos_amd", "bin", "fav_stylize_vr")
    r = subprocess.run([exe, "-input_pattern", "v/%S/f_%05d-%d.ppm", "-flow_pattern", "v/%S/flow-%d/backward_[%d]_{%d}.flo",
                        "-occlusions_pattern", "v/%S/flow-%d/reliable_[%d]_{%d}.pgm", "-output_prefix", "o/%S/out", "-model_vid", "m.t7", "-gpu", "2",
                        "-streams", "x,y,z", "-gpus", "2", "-dry_run", "1"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    recs = sorted((json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")), key=lambda d: d["rank"])
    assert [d["device"] for d in recs] == [2, 3]
    assert [[s["name"] for s in d["streams"]] for d in recs] == [["x", "z"], ["y"]]
    assert recs[1]["streams"][0]["occlusions_pattern"] == "v/y/flow-%d/reliable_[%d]_{%d}.pgm" and recs[1]["streams"][0]["output_prefix"] == "o/y/out"


def test_mask_edge_cases(oracle):
    h, w = 8, 12
    z = np.zeros((h, w, 2), np.float32)
    m = oracle.consistency(z, z)
    assert (m[:-1, :-1] == 255).all() and (m[-1, :] == 0).all() and (m[:, -1] == 0).all()   # x2>=W / y2>=H -> 0
    far = np.full((h, w, 2), 100.0, np.float32)
    assert (oracle.consistency(far, z) == 0).all()


# ---------------------------------------------------------------------------------------------- warp & co
def _warp_numpy(img, flow, mode):
    c, h, w = img.shape
    out = np.zeros((c, flow.shape[1], flow.shape[2]), np.float64)
    for y in range(flow.shape[1]):
        for x in range(flow.shape[2]):
            yf = np.float32(y) + flow[0, y, x]; xf = np.float32(x) + flow[1, y, x]
            if mode == "cpu" and (yf < 0 or yf > h - 1 or xf < 0 or xf > w - 1):
                continue
            x0, y0 = int(np.floor(xf)), int(np.floor(yf))
            ax, ay = float(xf) - x0, float(yf) - y0
            for dy, wy in ((0, 1 - ay), (1, ay)):
                for dx, wx in ((0, 1 - ax), (1, ax)):
                    yy, xx = y0 + dy, x0 + dx
                    if mode == "cpu":
                        yy, xx = min(yy, h - 1), min(xx, w - 1)
                    elif not (0 <= yy < h and 0 <= xx < w):
                        continue
                    out[:, y, x] += wy * wx * img[:, yy, xx]
    return out


@pytest.mark.parametrize("name", ["warp_fringe_37x53", "warp_wide_8x600", "warp_batch_resize", "warp_extreme_16x32"])
def test_oracle_warp_matches_reference_kernel_golden(oracle, golden_dir, name):
    """A2 pinned: the fixtures are outputs of the reference's OWN warp kernel (stnbdhw/BilinearSamplerBDHW.cu:1-109 compiled for
    gfx950, run on an MI355X by tests/golden/make_golden.py --warp).  The fp32 restatement is bit-identical to the kernel built
    without FMA contraction (NaN / +-inf / |flow| >= 2^31 included: the same saturating conversion), the double-rounded one stays
    within 1e-6 relative of both builds."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    img, flow = g["img"], g["flow"]
    scale = max(1.0, float(np.abs(img).max()))
    for b in range(img.shape[0]):
        o32 = oracle.warp(img[b], flow[b], "stn_f32")
        assert np.array_equal(o32, g["out_nofma"][b], equal_nan=True)
        if name == "warp_extreme_16x32":
            continue        # (fp32 products of 1e20-sized weights overflow where the double-rounded form does not: fp32 form only)
        o64 = oracle.warp(img[b], flow[b], "stn")
        for ref in (g["out"][b], g["out_nofma"][b]):
            assert np.isfinite(ref).all() and np.abs(o64 - ref).max() <= 1e-6 * scale
    assert np.isnan(g["out_nofma"]).any() == (name == "warp_extreme_16x32")


@pytest.mark.parametrize("mode", ["stn", "cpu"])
def test_oracle_warp(oracle, mode):
    rng = np.random.default_rng(3)
    img = rng.standard_normal((3, 17, 23)).astype(np.float32)
    flow = (rng.standard_normal((2, 17, 23)) * 4).astype(np.float32)
    flow[:, 0, 0] = 0; flow[:, 1, 1] = (-1.5, -0.25); flow[:, 2, 2] = (30, 30)
    got = oracle.warp(img, flow, mode)
    assert np.abs(got - _warp_numpy(img, flow, mode)).max() < 1e-5
    ident = oracle.warp(img, np.zeros_like(flow), mode)
    assert np.array_equal(ident, img)


def test_oracle_min_filter_matches_maxpool(oracle):
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(4)
    cert = (rng.random((31, 45)) > 0.2).astype(np.float32)
    got = oracle.min_filter(cert, 7)
    ref = 1 - F.max_pool2d(1 - torch.from_numpy(cert)[None, None], 7, 1, 3)[0, 0].numpy()
    assert np.array_equal(got, ref)
    grey = rng.random((9, 9)).astype(np.float32)
    ref = 1 - F.max_pool2d(1 - torch.from_numpy(grey)[None, None], 7, 1, 3)[0, 0].numpy()
    assert np.abs(oracle.min_filter(grey, 7) - ref).max() < 2e-7


def test_oracle_assemble_and_preprocess(oracle):
    rng = np.random.default_rng(5)
    fr = rng.random((3, 6, 7)).astype(np.float32); wp = rng.random((3, 6, 7)).astype(np.float32)
    ce = (rng.random((6, 7)) > 0.5).astype(np.float32)
    mean = np.array([103.939, 116.779, 123.68], np.float32)[:, None, None]
    pre = lambda a: a[::-1] * np.float32(255) - mean                  # preprocess.lua:57-62
    x = oracle.assemble(fr, wp, ce)
    assert np.allclose(x[:3], pre(fr), atol=1e-4) and np.allclose(x[3:6], pre(wp) * ce, atol=1e-4) and np.array_equal(x[6], ce)
    x0 = oracle.assemble(fr, None, None)
    assert np.allclose(x0[:3], pre(fr), atol=1e-4) and (x0[3:] == 0).all()
    assert np.allclose(oracle.deprocess(oracle.preprocess(fr)), fr, atol=1e-6)


# ---------------------------------------------------------------------------------------------- network
def test_oracle_net_matches_torch_golden(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "tiny_net_io.npz"))
    layers = t7.extract_layers(t7.load(os.path.join(golden_dir, "tiny_model.t7"))["model"])
    y = oracle.net_forward(layers, g["x"])
    assert y.shape == g["y"].shape
    assert np.abs(y - g["y"]).max() < 2e-3          # fp32 storage between layers vs the fp64 restatement


def test_oracle_layers_vs_torch(oracle):
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(6)
    x = rng.standard_normal((5, 13, 18)).astype(np.float32)
    for k, s, p, co in [(3, 1, 0, 8), (3, 2, 1, 4), (9, 1, 4, 3), (1, 1, 0, 2)]:
        w = rng.standard_normal((co, 5, k, k)).astype(np.float32); b = rng.standard_normal(co).astype(np.float32)
        ref = F.conv2d(torch.from_numpy(x)[None].double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), s, p)[0]
        assert np.abs(oracle.conv2d(x, w, b, s, p) - ref.numpy()).max() < 1e-4
    g = rng.random(5).astype(np.float32); bt = rng.standard_normal(5).astype(np.float32)
    ref = F.instance_norm(torch.from_numpy(x)[None].double(), weight=torch.from_numpy(g).double(), bias=torch.from_numpy(bt).double(), eps=1e-5)[0]
    assert np.abs(oracle.instnorm_(x.copy(), g, bt, 1e-5) - ref.numpy()).max() < 1e-5
    assert np.array_equal(oracle.reflect_pad(x, 3, 3, 3, 3), F.pad(torch.from_numpy(x)[None], (3, 3, 3, 3), mode="reflect")[0].numpy())
    assert np.array_equal(oracle.upsample(x, 2), F.interpolate(torch.from_numpy(x)[None], scale_factor=2, mode="nearest")[0].numpy())
    blk = rng.standard_normal((5, 9, 14)).astype(np.float32)
    assert np.array_equal(oracle.shave_add(blk, x, 2), blk + x[:, 2:-2, 2:-2])


def test_oracle_image_model_layers_vs_torch(oracle):
    """nn.SpatialFullConvolution / nn.SpatialBatchNormalization(evaluate) of the first-frame image models"""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(8)
    x = rng.standard_normal((6, 7, 9)).astype(np.float32)
    w = rng.standard_normal((6, 4, 3, 3)).astype(np.float32); b = rng.standard_normal(4).astype(np.float32)
    ref = F.conv_transpose2d(torch.from_numpy(x)[None].double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(),
                             stride=2, padding=1, output_padding=1)[0].numpy()
    got = oracle.full_conv2d(x, w, b, 2, 1, 1)
    assert got.shape == (4, 14, 18) and np.abs(got - ref).max() < 1e-5
    mean = rng.standard_normal(6).astype(np.float32); var = rng.uniform(0.5, 2, 6).astype(np.float32)
    g = rng.random(6).astype(np.float32); bt = rng.standard_normal(6).astype(np.float32)
    ref = F.batch_norm(torch.from_numpy(x)[None].double(), torch.from_numpy(mean).double(), torch.from_numpy(var).double(),
                       torch.from_numpy(g).double(), torch.from_numpy(bt).double(), False, 0.1, 1e-5)[0].numpy()
    assert np.abs(oracle.batchnorm_eval_(x.copy(), mean, var, g, bt, 1e-5) - ref).max() < 1e-5


def _torch_forward(layers, x):
    """the layer list in torch, double precision: what Torch7's modules compute [recalled semantics, SURVEY.md Appendix C]"""
    import torch
    import torch.nn.functional as F
    T = lambda a: torch.from_numpy(np.asarray(a)).double()
    for L in layers:
        t = L["type"]
        if t == "pad": x = F.pad(x, (L["l"], L["r"], L["t"], L["b"]), mode="replicate" if L["mode"] == "replicate" else "reflect")
        elif t == "conv": x = F.conv2d(x, T(L["w"]), None if L["b"] is None else T(L["b"]), L["stride"], L["pad"])
        elif t == "in": x = F.instance_norm(x, weight=T(L["gamma"]), bias=T(L["beta"]), eps=L["eps"])
        elif t == "relu": x = F.relu(x)
        elif t == "up": x = F.interpolate(x, scale_factor=L["s"], mode="nearest")
        elif t == "res":
            y = _torch_forward(L["block"], x); s = L["shave"]
            x = y + (x[:, :, s:-s, s:-s] if s else x)
        elif t == "tanh": x = torch.tanh(x)
        elif t == "mul": x = x * L["k"]
        elif t == "identity": pass
        else: raise ValueError(t)
    return x


@pytest.mark.parametrize("ptype", ["reflect-start", "none", "reflect", "replicate", "zero"])
def test_every_padding_type_oracle_vs_torch_and_both_readers(oracle, favlib, tmp_path, ptype):
    """train_video.lua:25 -padding_type: the five forms models_video.lua:10-53,65-80 builds (padding modules in front of every convolution
    for reflect / replicate, zero-padded block convolutions for zero, shaved skips for none / reflect-start).  The emitter mirrors the
    builder, both readers agree on the layer list, and the oracle's forward matches a torch (double) restatement."""
    import torch
    p = str(tmp_path / "m.t7")
    t7.make_synthetic_checkpoint(p, arch="c9s1-8,d16,R16,R16,U2,c3s1-8,c9s1-3", seed=5, padding_type=ptype)
    layers = t7.extract_layers(t7.load(p)["model"])
    assert favlib.describe_t7(p) == favlib.describe_layers(layers)
    kinds = [L["type"] for L in layers]
    if ptype in ("reflect", "replicate"):
        assert kinds[0] == "pad" and layers[0]["l"] == 4 and layers[0]["mode"] == ptype and kinds.count("pad") == 3
        blk = [L for L in layers if L["type"] == "res"][0]
        assert [b["type"] for b in blk["block"]] == ["pad", "conv", "in", "relu", "pad", "conv", "in"] and blk["shave"] == 0 and blk["block"][1]["pad"] == 0
    elif ptype == "zero":
        blk = [L for L in layers if L["type"] == "res"][0]
        assert "pad" not in kinds and blk["shave"] == 0 and blk["block"][0]["pad"] == 1
    else:
        assert (kinds[0] == "pad") == (ptype == "reflect-start") and [L for L in layers if L["type"] == "res"][0]["shave"] == 2
        if ptype == "reflect-start": assert layers[0]["l"] == 8                     # 2 blocks x 2 px at 1/2 resolution
    x = np.random.default_rng(2).standard_normal((7, 28, 36)).astype(np.float32)
    y = oracle.net_forward(layers, x)
    ref = _torch_forward(layers, torch.from_numpy(x)[None].double())[0].numpy()
    assert y.shape == ref.shape == ((3, 28, 36) if ptype not in ("none",) else ref.shape)
    assert np.abs(y - ref).max() < 2e-3, float(np.abs(y - ref).max())


def _convs(ls):
    for L in ls:
        if L["type"] == "conv": yield L
        if L["type"] == "res": yield from _convs(L["block"])


def _ins(ls):
    for L in ls:
        if L["type"] == "in": yield L
        if L["type"] == "res": yield from _ins(L["block"])


def test_canonical_architecture_shapes(tmp_path):
    p = str(tmp_path / "m.t7")
    t7.make_synthetic_checkpoint(p, seed=1)
    layers = t7.extract_layers(t7.load(p)["model"])
    assert layers[0] == {"type": "pad", "l": 40, "r": 40, "t": 40, "b": 40, "mode": "reflect"}        # train_video.lua:319-325
    n = sum(L["w"].size + L["b"].size for L in _convs(layers)) + sum(2 * len(L["gamma"]) for L in _ins(layers))
    assert n == 1679235                                                              # 6.72 MB


# ---------------------------------------------------------------------------------------------- libfav host side
def test_c_abi_exports_every_declared_symbol(favlib):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "fav.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(fav_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    L = favlib.lib()
    for name in declared:
        assert hasattr(L, name), f"libfav.so does not export {name}"
    assert declared == set(favlib.EXPORTS)


def test_release_library_reads_only_the_documented_environment_variables(favlib):
    """the library that ships must not change its kernels because of a stray FAV_NO_WINO in somebody's environment (round-5 review): every
    kernel-selection / tuning / debug switch goes through diag_env() (csrc/fav_internal.h) and exists in libfav_diag.so only.  The names
    a binary can pass to getenv are in its string table: the release library's are exactly the documented ones."""
    import re
    pkg = os.path.join(ROOT, "fast-artistic-videos_amd")
    names = lambda f: set(m.decode() for m in re.findall(rb"FAV_[A-Z0-9_]{2,}", open(os.path.join(pkg, f), "rb").read()))
    documented = {"FAV_SIDE_CUS", "FAV_SIDE_QUEUES", "FAV_ROCTX"}
    constants = {"FAV_PRECISION_BF16_OPERANDS", "FAV_PRECISION_FP32"}                 # (enumerators quoted in error messages)
    assert names("libfav.so") - constants == documented, names("libfav.so")
    if os.path.exists(os.path.join(pkg, "libfav_diag.so")):
        diag = names("libfav_diag.so")
        assert {"FAV_NO_WINO", "FAV_WINO_F2", "FAV_W4_GRID", "FAV_NO_CHECK_PREP"} <= diag and documented <= diag
    for exe, allowed in (("bin/fav_stylize", {"FAV_RCCL_TIMEOUT_S", "FAV_TEST_WORKER_FAIL"}), ("bin/fav_stylize_vr", {"FAV_RCCL_TIMEOUT_S", "FAV_TEST_WORKER_FAIL"}),
                         ("bin/consistencyChecker", {"FAV_CC_", "FAV_CC_DAEMON", "FAV_CC_IDLE_S", "FAV_CC_REPLY_S", "FAV_CC_TIMING", "FAV_GPU"})):
        assert names(exe) <= allowed, (exe, names(exe))


def test_no_gpu_means_loud_failure(favlib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = favlib.lib()
    assert L.fav_device_count() == -6                      # FAV_ENODEVICE
    assert b"no CPU fallback" in L.fav_last_error()
    h = C.c_void_p()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert L.fav_net_create(os.path.join(root, "tests/golden/tiny_model.t7").encode(), 0, C.byref(h)) == -6
    with pytest.raises(favlib.FavError):
        favlib.warp(None, None)


def test_cpp_t7_reader_matches_python_reader(favlib, golden_dir, tmp_path):
    path = os.path.join(golden_dir, "tiny_model.t7")
    assert favlib.describe_t7(path) == favlib.describe_layers(t7.extract_layers(t7.load(path)["model"]))
    p = str(tmp_path / "canon.t7")
    t7.make_synthetic_checkpoint(p, seed=3)
    assert favlib.describe_t7(p) == favlib.describe_layers(t7.extract_layers(t7.load(p)["model"]))
    blob = favlib.pack_checkpoint(p)
    assert blob[:4] == b"FAVB" and len(blob) > 1679235 * 4
    for inorm in (True, False):      # first-frame image models: SpatialFullConvolution, optional BatchNorm
        pi = str(tmp_path / f"img{int(inorm)}.t7")
        t7.make_synthetic_checkpoint(pi, arch=t7.IMAGE_ARCH, seed=4, in_channels=3, use_instance_norm=inorm)
        d = favlib.describe_t7(pi)
        assert d == favlib.describe_layers(t7.extract_layers(t7.load(pi)["model"]))
        assert "fullconv 128 64 3 2 1 adj=1" in d and ("in 64" in d if inorm else "bn 64" in d)
    # malformed inputs fail with a status, never crash
    bad = str(tmp_path / "bad.t7")
    with open(bad, "wb") as f:
        f.write(open(p, "rb").read()[:1000])
    with pytest.raises(favlib.FavError):
        favlib.describe_t7(bad)
    with pytest.raises(favlib.FavError):
        favlib.describe_t7(str(tmp_path / "missing.t7"))


def test_host_formats_roundtrip(favlib, oracle, tmp_path):
    from PIL import Image
    uv = synth.random_flow(11, 7, 1)
    p = str(tmp_path / "f.flo"); oracle.write_flo(p, uv)
    assert np.array_equal(favlib.read_flo(p), uv)
    img = synth.random_frame(9, 13, 2)
    pp = str(tmp_path / "i.ppm"); oracle.write_pnm(pp, img)
    assert np.array_equal(favlib.read_pnm(pp), img)
    with open(pp, "wb") as f:
        f.write(b"P6\n# a comment\n13 9\n255\n" + img.tobytes())
    assert np.array_equal(favlib.read_pnm(pp), img)
    pg = str(tmp_path / "m.pgm"); favlib.write_pgm(pg, img[..., 0])
    assert open(pg, "rb").read() == b"P5\n13 9\n255\n" + img[..., 0].tobytes()      # CMatrix.h:1064
    png = str(tmp_path / "o.png"); favlib.write_png(png, img)
    assert np.array_equal(np.asarray(Image.open(png)), img)
    with pytest.raises(favlib.FavError):
        favlib.read_flo(str(tmp_path / "nope.flo"))


# ------------------------------------------------------------------------------------------ cube-map orchestration (8f rank 1)
def test_vr_static_maps_host_vs_oracle(favlib):
    """The perspective / equirectangular maps are host code (double, Lua operation order): libfav's C++ and the numpy
    restatement of vr_helper.lua must agree bit for bit, and every crop column / row must be defined exactly once."""
    import vr_oracle as V
    for (hp, wp, ow, oh) in [(96, 96, 32, 32), (64, 80, 20, 24), (224, 224, 64, 64)]:
        want = [V.warp_map_left(hp, ow, wp), V.warp_map_right(hp, ow, wp), V.warp_map_top(wp, oh, hp), V.warp_map_bottom(wp, oh, hp)]
        for k, m in enumerate(want):
            got = favlib.vr_map_host(k, hp, wp, ow if k < 2 else oh)
            assert np.array_equal(got, m), (hp, wp, k)
            defined = (m[0] != 99999)
            assert int(defined.any(axis=0).sum()) == (ow if k < 2 else wp) and int(defined.any(axis=1).sum()) == (hp if k < 2 else oh)
    e = V.equirect_map(96 - 2, 96 - 2, 32 - 1, 32 - 1, 128, 64)
    assert np.array_equal(favlib.vr_map_host(4, 96, 96, 32, 3, 128, 64), e)
    # every equirectangular sample lands inside the 6-face strip
    ys = e[0] + np.arange(64, dtype=np.float32)[:, None]; xs = e[1] + np.arange(128, dtype=np.float32)[None, :]
    assert ys.min() >= 0 and ys.max() <= 94 and xs.min() >= 0 and xs.max() <= 6 * 94


def test_vr_oracle_properties(oracle, golden_dir):
    """Size-independent properties of the orchestration restatement: rotations compose, the left/right (top/bottom) masks
    mirror each other, a constant-colour cube stays constant through prior blending, post-blend and both output maps."""
    import vr_oracle as V
    from fav_amd import t7
    a = np.random.default_rng(0).random((3, 5, 7)).astype(np.float32)
    assert np.array_equal(V.rotate_minus90(V.rotate90(a)), a) and np.array_equal(V.rotate180(V.rotate180(a)), a)
    assert np.array_equal(V.rotate90(V.rotate90(a)), V.rotate180(a))
    m = V.median_filter(a, 3)
    assert m.shape == (3, 3, 5) and np.array_equal(m[0, 0, 0], np.sort(a[0, :3, :3].ravel())[4])
    layers = t7.extract_layers(t7.load(os.path.join(golden_dir, "tiny_model.t7"))["model"])
    vr = V.VRStylizer(layers, 64, 64, overlap_w=24, overlap_h=24, median=3, out_equi_w=64, out_equi_h=32)
    # (the left/right and top/bottom maps are near mirrors, not exact ones: the Lua loops run over fractional coordinates)
    for a_, b_ in ((vr.mask_left[0], vr.mask_right[0][:, ::-1]), (vr.mask_top[0], vr.mask_bottom[0][::-1])):
        assert a_.min() >= 0 and a_.max() <= 1 and abs(int((a_ > 0).sum()) - int((b_ > 0).sum())) <= 0.1 * (a_ > 0).sum() and np.abs(a_ - b_).mean() < 0.05
    assert vr.mask_all.max() == 1.0 and vr.mask_all_div.min() == 1.0 and vr.mask_all_div.max() <= 2.0
    const = np.full((3, 64, 64), 0.25, np.float32)
    vr.last = [const.copy() for _ in range(6)]
    vr._finish()
    inner = (slice(None), slice(12, 52), slice(12, 52))
    for k in range(6):
        assert np.abs(vr.blended[k][inner] - 0.25).max() <= 1e-6        # away from the outer rim the blend is an average of equals
    assert np.abs(vr.equi - 0.25).max() <= 1e-5 and np.abs(vr.cubemap - 0.25).max() <= 1e-5
    u = V.fill_uniform(3, 11, 16, 16)
    assert u.min() >= 0 and u.max() < 1 and 0.4 < u.mean() < 0.6 and not np.array_equal(u, V.fill_uniform(3, 12, 16, 16))


def test_vr_cli_rejects_what_it_does_not_provide(favlib):
    """fav_stylize_vr parses the whole flag set of fast_artistic_video_vr.lua:21-76 and fails loudly (no GPU needed to get
    there) on the CPU backend and on the options outside the path."""
    exe = os.path.join(ROOT, "fast-artistic-videos_amd", "bin", "fav_stylize_vr")
    base = [exe, "-input_pattern", "f_%05d-%d.ppm", "-flow_pattern", "x/backward_[%d]_{%d}.flo", "-occlusions_pattern", "x/reliable_[%d]_{%d}.pgm",
            "-model_vid", "m.t7", "-model_img", "self", "-overlap_pixel_w", "128", "-overlap_pixel_h", "128", "-out_equi", "-out_cubemap",
            "-fill_occlusions", "uniform-random", "-backend", "cuda", "-use_cudnn", "1"]
    for extra, msg in ((["-gpu", "-1"], "no CPU backend"), (["-gpu", "0", "-evaluate"], "outside the hot-path scope"),
                       (["-gpu", "0", "-backward"], "not provided"), (["-gpu", "0", "-continue_with", "3"], "not provided"),
                       (["-gpu", "0", "-no_such_flag", "1"], "unknown option"), ([], "no CPU backend")):
        r = subprocess.run(base + extra, capture_output=True, text=True)
        assert r.returncode != 0 and msg in r.stderr, (extra, r.stderr)
    r = subprocess.run([exe, "-gpu", "0"], capture_output=True, text=True)
    assert r.returncode != 0 and "Must give -input_pattern" in r.stderr


def test_cpp_t7_reader_survives_damaged_files(favlib, golden_dir, tmp_path):
    """Error behaviour of the checkpoint reader (core.lua:39-43 prints an error and stops): truncated and bit-flipped .t7
    files must come back as a status + message, never as a crash or a hang; run in a child process so a crash is seen."""
    import textwrap
    data = open(os.path.join(golden_dir, "tiny_model.t7"), "rb").read()
    rng = np.random.default_rng(0)
    cases = []
    for k, cut in enumerate(sorted(set([0, 1, 3, 4, 7, 8, 11, 12, 40, len(data) // 2, len(data) - 1] + list(rng.integers(1, len(data), 24))))):
        p = tmp_path / f"cut{k}.t7"; p.write_bytes(data[:int(cut)]); cases.append(str(p))
    for k in range(40):
        b = bytearray(data)
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(0, min(len(b), 4000)))] = int(rng.integers(0, 256))       # headers, sizes, class names, dims
        p = tmp_path / f"flip{k}.t7"; p.write_bytes(bytes(b)); cases.append(str(p))
    script = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {os.path.join(ROOT, "fast-artistic-videos_amd", "python")!r})
        import fav_amd
        ok = bad = 0
        for p in {cases!r}:
            try:
                fav_amd.describe_t7(p); ok += 1
            except fav_amd.FavError as e:
                assert str(e), p
                bad += 1
        print(ok, bad)
    """)
    r = subprocess.run([os.sys.executable, "-c", script], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    ok, bad = map(int, r.stdout.split())
    assert ok + bad == len(cases) and bad >= 30          # every truncation fails; a flipped payload byte may still parse


def test_into_buffer_readers(favlib, oracle, tmp_path):
    """fav_read_flo_into_host / fav_read_pnm_into_host: the loaders' no-allocation variants (pinned staging in fav_stylize)."""
    L = favlib.lib()
    h, w = 23, 31
    uv = synth.random_flow(h, w, 1); img = synth.random_frame(h, w, 2); m = (synth.random_frame(h, w, 3)[..., 0])
    pf, pi, pm = (str(tmp_path / n) for n in ("a.flo", "a.ppm", "a.pgm"))
    oracle.write_flo(pf, uv); oracle.write_pnm(pi, img); oracle.write_pnm(pm, m)
    W, H, ch = C.c_int(), C.c_int(), C.c_int()
    fb = np.zeros(h * w * 2, np.float32)
    assert L.fav_read_flo_into_host(pf.encode(), fb.ctypes.data_as(C.c_void_p), C.c_size_t(fb.size), C.byref(W), C.byref(H)) == 0
    assert (W.value, H.value) == (w, h) and np.array_equal(fb.reshape(h, w, 2), uv)
    ib = np.zeros(h * w * 3 + 5, np.uint8)
    assert L.fav_read_pnm_into_host(pi.encode(), ib.ctypes.data_as(C.c_void_p), C.c_size_t(ib.size), C.byref(W), C.byref(H), C.byref(ch)) == 0
    assert ch.value == 3 and np.array_equal(ib[:h * w * 3].reshape(h, w, 3), img)
    mb = np.zeros(h * w, np.uint8)
    assert L.fav_read_pnm_into_host(pm.encode(), mb.ctypes.data_as(C.c_void_p), C.c_size_t(mb.size), C.byref(W), C.byref(H), C.byref(ch)) == 0
    assert ch.value == 1 and np.array_equal(mb.reshape(h, w), m)
    # too small a buffer is an error, not an overrun
    assert L.fav_read_flo_into_host(pf.encode(), fb.ctypes.data_as(C.c_void_p), C.c_size_t(fb.size - 1), C.byref(W), C.byref(H)) == -1
    assert L.fav_read_pnm_into_host(pi.encode(), ib.ctypes.data_as(C.c_void_p), C.c_size_t(h * w * 3 - 1), C.byref(W), C.byref(H), C.byref(ch)) == -1
    assert b"does not fit" in L.fav_last_error()


def test_contractive_checkpoint_contracts_and_random_init_does_not(oracle, tmp_path):
    """The premise of the whole-clip free-running gate (tests/test_gpu_parity.py,
    scripts/parity_clip.py): with the first convolution's prior + certainty weights scaled by CONTRACTIVE_GAIN the recurrent map
    frame -> frame CONTRACTS -- two oracle chains that differ by +-5e-6 at frame 1 come closer every frame until they sit in the
    fp32 noise -- while the plain random-init checkpoint amplifies the same perturbation ~3x per frame (why free-running parity
    was ungateable on it: profiles/r03_parity_sensitivity_control_c2.json).  No GPU involved."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import parity_clip as PC
    from fav_amd import t7
    res = {}
    for name, gain in (("contractive", PC.CONTRACTIVE_GAIN), ("random-init", 1.0)):
        p = str(tmp_path / (name + ".t7"))
        t7.make_synthetic_checkpoint(p, seed=3, recurrent_gain=gain)
        res[name] = PC.run_control(oracle, p, 96, 128, 5, mode="flow3", log=lambda s: None)["per_frame"]
    c, r = [x["rms"] for x in res["contractive"]], [x["rms"] for x in res["random-init"]]
    assert c[1] < 0.8 * c[0] and c[2] < 0.8 * c[1] and c[4] < 0.3 * c[0], c
    assert r[4] > 10 * r[0], r


@pytest.mark.parametrize("case", ["uniform", "ties", "mixed", "zeros", "negatives", "structure-like", "stagnate"])
def test_multi_block_sequential_sum_scheme(case):
    """the scheme of the multi-block CMatrix::avg sum (round 4: predicted binades, chunk transducers, native crossings, hand-over to
    the one-block finisher) restated in numpy (tests/util/seqsum_model.py) against the plain sequential fp32 loop -- bit for bit"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "util"))
    import seqsum_model as SM
    rng = np.random.default_rng(sum(map(ord, case))); n = 150001
    if case == "uniform": x = rng.random(n, dtype=np.float32)
    elif case == "ties": x = (rng.integers(0, 4096, n) * 2.0 ** -12).astype(np.float32)
    elif case == "mixed": x = (10.0 ** rng.uniform(-9, 3, n)).astype(np.float32)
    elif case == "zeros":
        x = rng.random(n, dtype=np.float32); x[rng.random(n) < 0.7] = 0; x[:5000] = 0
    elif case == "negatives":
        x = rng.random(n, dtype=np.float32); x[rng.random(n) < 0.001] *= -1
    elif case == "stagnate":
        x = np.full(n, 1e-3, np.float32); x[0] = 60000.0
    else: x = (rng.random(n) ** 6).astype(np.float32)
    got, info = SM.model(x)
    want = SM.seq_sum(x)
    assert got.tobytes() == want.tobytes(), (case, got, want, info)
    if case in ("uniform", "ties", "mixed", "structure-like"):
        assert info["halted_at"] is None and info["rounds"] <= 20, info          # the fast path carries the whole array
    if case in ("zeros", "negatives"):
        assert info["halted_at"] is not None, info                               # the finisher takes over where the premise breaks
