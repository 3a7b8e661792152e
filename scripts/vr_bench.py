"""BASELINE config 5 on one GPU: 360-degree cube-map path, 6 x 1504x1504 faces per frame (overlap 128, equirect 2560x1440 as in
stylizeVRVideo_deepflow.sh:76-83), canonical architecture with synthetic weights, inputs resident in HBM.
usage: python scripts/vr_bench.py [--frames 6] [--face 1504] [--arch 2x|1.5x|<architecture string>]
(--arch: a checkpoint with more filters, as the reference's published VR models have -- README.md:141)"""
import argparse, json, os, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-artistic-videos_amd", "python"))
import numpy as np
import torch
import fav_amd
from fav_amd import t7

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=6)
ap.add_argument("--face", type=int, default=1504)
ap.add_argument("--overlap", type=int, default=128)
ap.add_argument("--arch", default="", help="2x | 1.5x | an architecture string of models_video.lua (default: the canonical one)")
a = ap.parse_args()
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import wide_bench
arch = {"": t7.CANONICAL_ARCH, "2x": wide_bench.WIDE2, "1.5x": wide_bench.WIDE15}.get(a.arch, a.arch)
dev = torch.device("cuda:0")
with tempfile.TemporaryDirectory() as d:
    ck = os.path.join(d, "m.t7"); t7.make_synthetic_checkpoint(ck, arch=arch, seed=1)
    net = fav_amd.Net(ck, 0)
hp = a.face
vr = fav_amd.VR(net, hp, hp, overlap_w=a.overlap, overlap_h=a.overlap, median=3, out_equi_w=2560, out_equi_h=1440, fill_random=True, seed=3)
g = torch.Generator(device="cpu").manual_seed(0)
faces = [torch.randint(0, 256, (hp, hp, 3), dtype=torch.uint8, generator=g).to(dev) for _ in range(6)]
flows = [(torch.randn((hp, hp, 2), generator=g) * 2).to(dev) for _ in range(6)]
certs = [((torch.rand((hp, hp), generator=g) > 0.1).to(torch.uint8) * 255).to(dev) for _ in range(6)]


def frame(fr):
    for m in range(6):
        i = fr * 6 + m + 1
        lib = fav_amd.lib()
        fav_amd._check(lib.fav_vr_face(vr.h, i, fav_amd._p(faces[m]), fav_amd._p(flows[m]) if i >= 7 else None,
                                       fav_amd._p(certs[m]) if i >= 7 else None, None, fav_amd._stream()))
    return vr.finish_frame()


frame(0); frame(1); torch.cuda.synchronize()
t0 = time.perf_counter()
for fr in range(2, 2 + a.frames):
    e, c = frame(fr)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"workload": "VR cube map, 6 x %dx%d faces/frame, overlap %d, equirect 2560x1440 + cube map, 1 GPU" % (hp, hp, a.overlap), "arch": arch,
                  "frames": a.frames, "frames_per_s": round(a.frames / dt, 3), "faces_per_s": round(6 * a.frames / dt, 2),
                  "ms_per_frame": round(dt / a.frames * 1e3, 2), "equi": list(e.shape), "cubemap": list(c.shape)}))
