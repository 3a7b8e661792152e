"""Generate the committed golden fixtures.  Run in the build container (needs /root/reference for the
mask fixtures: they are outputs of the reference's OWN consistencyChecker, compiled by oracle/Makefile
into oracle/_ref/, on seeded synthetic inputs).  The network fixture comes from a PyTorch-CPU fp64
restatement (the reference's Lua/Torch7 stack cannot run offline) -- it cross-checks the C oracle, it
is not a reference output.

    python tests/golden/make_golden.py
    python tests/golden/make_golden.py --warp --out gpurun_out/golden      (ON THE GPU BOX: warp fixtures)

The warp fixtures (warp_*.npz) are outputs of the reference's OWN warp kernel (stnbdhw/BilinearSamplerBDHW.cu:1-109 compiled for
gfx950 by oracle/Makefile into oracle/_ref/libwarp_ref{,_nofma}.so); that kernel only runs on a GPU, so they are generated there
(`gpurun -- python tests/golden/make_golden.py --warp --out gpurun_out/golden`) and copied into tests/golden/.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "fast-artistic-videos_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O          # noqa: E402
from fav_amd import synth, t7   # noqa: E402


def mask_fixture(name, h, w, kind, seed):
    if kind == "smooth":
        bw = synth.backward_flow(h, w, seed); fw = synth.forward_flow_from_backward(bw, seed + 1)
    else:
        bw = synth.random_flow(h, w, seed); fw = synth.random_flow(h, w, seed + 1)
    img = synth.smooth_frame(h, w, seed + 2)
    d = tempfile.mkdtemp()
    O.write_flo(d + "/a.flo", bw); O.write_flo(d + "/b.flo", fw); O.write_pnm(d + "/i.ppm", img)
    subprocess.check_call([O.REF_CHECKER, d + "/a.flo", d + "/b.flo", d + "/o3.pgm"], stdout=subprocess.DEVNULL)
    subprocess.check_call([O.REF_CHECKER, d + "/a.flo", d + "/b.flo", d + "/o4.pgm", d + "/i.ppm"], stdout=subprocess.DEVNULL)
    m3, m4 = O.read_pnm(d + "/o3.pgm"), O.read_pnm(d + "/o4.pgm")
    np.savez_compressed(os.path.join(HERE, name), bw=bw, fw=fw, img=img, mask3=m3, mask4=m4)
    print(name, "reliable% 3-arg", (m3 == 255).mean() * 100, "4-arg", (m4 == 255).mean() * 100, "flips", int((m3 != m4).sum()))


def net_fixture():
    import torch
    import torch.nn.functional as F
    path = os.path.join(HERE, "tiny_model.t7")
    t7.make_synthetic_checkpoint(path, arch="c9s1-8,d16,d32,R32,R32,U2,c3s1-16,U2,c9s1-3", seed=7)
    layers = t7.extract_layers(t7.load(path)["model"])

    def tnet(layers, x):
        for L in layers:
            t = L["type"]
            if t == "pad": x = F.pad(x, (L["l"], L["r"], L["t"], L["b"]), mode="reflect")
            elif t == "conv": x = F.conv2d(x, torch.from_numpy(L["w"]).double(), torch.from_numpy(L["b"]).double(), L["stride"], L["pad"])
            elif t == "in": x = F.instance_norm(x, weight=torch.from_numpy(L["gamma"]).double(), bias=torch.from_numpy(L["beta"]).double(), eps=L["eps"])
            elif t == "relu": x = F.relu(x)
            elif t == "res":
                y = tnet(L["block"], x); s = L["shave"]; x = y + x[:, :, s:x.shape[2] - s, s:x.shape[3] - s]
            elif t == "up": x = F.interpolate(x, scale_factor=L["s"], mode="nearest")
            elif t == "tanh": x = torch.tanh(x)
            elif t == "mul": x = x * L["k"]
        return x
    rng = np.random.default_rng(11)
    x = (rng.standard_normal((7, 40, 56)) * 50).astype(np.float32)
    y = tnet(layers, torch.from_numpy(x)[None].double())[0].numpy()
    np.savez_compressed(os.path.join(HERE, "tiny_net_io.npz"), x=x, y=y)
    print("tiny net", y.shape, float(np.abs(y).max()))


def warp_cases():
    """seeded inputs of the warp fixtures: name -> (img [B][C][H][W], flow [B][2][Ho][Wo])"""
    rng = np.random.default_rng(4242)
    cases = {}
    img = rng.standard_normal((1, 3, 37, 53)).astype(np.float32)
    flow = (rng.standard_normal((1, 2, 37, 53)) * 5).astype(np.float32)
    flow[0, :, 0, :4] = [[-1.5] * 4, [-0.5, 0.0, 0.25, -2.0]]        # the 1-px fringe band where the taps leave the image one by one
    flow[0, :, -1, -3:] = 0.75
    flow[0, :, 5, 5] = [0.0, 0.0]; flow[0, :, 6, 6] = [1.0, -1.0]     # exact integer offsets (weight 1 / 0 taps)
    flow[0, 1, 7, :] = (53 - 1) - np.arange(53)                       # lands exactly on the last column
    flow[0, 1, 8, :] = -1 - np.arange(53)                             # exactly one column outside
    cases["warp_fringe_37x53"] = (img, flow)
    # wider than the 512 columns one block row of the reference's launch covers (BilinearSamplerBDHW.cu:58,119)
    cases["warp_wide_8x600"] = (rng.standard_normal((1, 1, 8, 600)).astype(np.float32), (rng.standard_normal((1, 2, 8, 600)) * 3).astype(np.float32))
    # batched, and the output takes the FLOW's size (BilinearSamplerBDHW.lua:54-82)
    cases["warp_batch_resize"] = (rng.standard_normal((2, 3, 20, 31)).astype(np.float32), (rng.standard_normal((2, 2, 24, 40)) * 4).astype(np.float32))
    img = (rng.random((1, 2, 16, 32)) + 0.5).astype(np.float32)
    flow = (rng.standard_normal((1, 2, 16, 32)) * 2).astype(np.float32)
    ext = np.array([np.nan, np.inf, -np.inf, 2.0 ** 31, -2.0 ** 31, 3e9, -3e9, 1e20, -1e20, -0.0, 1e-40, 2147483520.0, -2147483520.0,
                    2.0 ** 31 - 200, 65536.5, -65536.5], np.float32)
    flow[0, 0, 2, :16] = ext; flow[0, 1, 3, :16] = ext                # one axis extreme, the other ordinary
    flow[0, 0, 4, :16] = ext; flow[0, 1, 4, :16] = ext[::-1]          # both
    flow[0, 0, 5, :16] = ext; flow[0, 1, 5, :16] = 0.0                # extreme dy on an exact column
    cases["warp_extreme_16x32"] = (img, flow)
    return cases


def warp_fixture(out_dir):
    import torch
    assert torch.cuda.is_available() and O.warp_ref_available(), "the warp fixtures need a GPU and oracle/_ref/libwarp_ref*.so"
    os.makedirs(out_dir, exist_ok=True)
    dev = torch.device("cuda:0")
    for name, (img, flow) in warp_cases().items():
        a = O.warp_ref_gpu(torch.from_numpy(img).to(dev), torch.from_numpy(flow).to(dev), contract=True).cpu().numpy()
        b = O.warp_ref_gpu(torch.from_numpy(img).to(dev), torch.from_numpy(flow).to(dev), contract=False).cpu().numpy()
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), img=img, flow=flow, out=a, out_nofma=b)
        fin = np.isfinite(a) & np.isfinite(b)
        print(name, a.shape, "non-finite outputs", int((~np.isfinite(b)).sum()), "max |contracted - uncontracted|", float(np.abs(a - b)[fin].max()),
              "differing elements", int((a[fin] != b[fin]).sum()))


if __name__ == "__main__":
    if "--warp" in sys.argv:
        warp_fixture(sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else HERE)
        sys.exit(0)
    O.build()
    assert os.path.exists(O.REF_CHECKER), "oracle/_ref/consistencyChecker missing (needs /root/reference)"
    mask_fixture("mask_smooth_64x96.npz", 64, 96, "smooth", 100)
    mask_fixture("mask_rand_120x160.npz", 120, 160, "rand", 200)
    mask_fixture("mask_smooth_180x320.npz", 180, 320, "smooth", 300)
    net_fixture()
