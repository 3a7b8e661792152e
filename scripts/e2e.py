#!/usr/bin/env python3
"""File -> PNG throughput of bin/fav_stylize on RAM-backed files (the same run bench.py's `e2e` block makes), with the CLI's own
timing breakdown, for a few host configurations.  usage: e2e.py [frames] ; env FAV_E2E_VARIANTS="name:flag value ...;..." """
import json, os, subprocess, sys, shutil, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-artistic-videos_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle as O
from fav_amd import synth, t7
H, W, N = 720, 1280, int(sys.argv[1]) if len(sys.argv) > 1 else 300
d = tempfile.mkdtemp(prefix="fav_e2e_", dir="/dev/shm")
os.makedirs(d + "/flow"); os.makedirs(d + "/src")
model = d + "/canonical.t7"; t7.make_synthetic_checkpoint(model, seed=1234)
for k in range(4):
    O.write_pnm(f"{d}/src/f{k}.ppm", synth.random_frame(H, W, k))
    bw = synth.backward_flow(H, W, 10 + k); O.write_flo(f"{d}/src/b{k}.flo", bw); O.write_flo(f"{d}/src/w{k}.flo", synth.forward_flow_from_backward(bw, 20 + k))
t_old = time.time() - 30.0            # finished inputs (host/fav_poll.h: anything younger than -poll_settle is watched first)
for f in os.listdir(d + "/src"): os.utime(f"{d}/src/{f}", (t_old, t_old))
for i in range(1, N + 1):
    os.symlink(f"{d}/src/f{i % 4}.ppm", f"{d}/frame_{i:05d}.ppm")
    if i > 1:
        os.symlink(f"{d}/src/b{i % 4}.flo", f"{d}/flow/backward_{i}_{i-1}.flo"); os.symlink(f"{d}/src/w{i % 4}.flo", f"{d}/flow/forward_{i-1}_{i}.flo")
exe = os.path.join(ROOT, "fast-artistic-videos_amd", "bin", "fav_stylize")
base = [exe, "-input_pattern", d + "/frame_%05d.ppm", "-flow_pattern", d + "/flow/backward_[%d]_{%d}.flo", "-forward_flow_pattern", d + "/flow/forward_{%d}_[%d].flo",
        "-structure", "0", "-model_vid", model, "-model_img", "self", "-gpu", "0", "-timing", "1"]
variants = os.environ.get("FAV_E2E_VARIANTS", "gpu_png:-png_encoder gpu;gpu_png_w2:-png_encoder gpu -writers 2;host_png1:-png_encoder host -png_level 1")
for v in variants.split(";"):
    name, flags = v.split(":", 1)
    env = dict(os.environ)
    words = []
    for wd in flags.split():            # NAME=VALUE words are environment settings of the run, the rest are flags
        if "=" in wd and not wd.startswith("-"): k, val = wd.split("=", 1); env[k] = val
        else: words.append(wd)
    t0 = time.time()
    wrap = [w.replace("{name}", name) for w in env.pop("FAV_E2E_WRAP", "").split()]      # e.g. "rocprofv3 --kernel-trace --stats -d gpurun_out/p_{name} -o t --"
    r = subprocess.run(wrap + base + ["-output_prefix", f"{d}/o_{name}/out"] + words, capture_output=True, text=True, env=env)
    wall = time.time() - t0
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    print(name, "wall %.2f s" % wall, line[-1] if line else ("FAILED " + r.stderr[-300:]), flush=True)
    for l in r.stdout.splitlines():
        if l.startswith("thread CPU seconds"): print("   ", l, flush=True)
    for l in r.stderr.splitlines():
        if l.startswith("loop trace"): print("   ", l, flush=True)
    shutil.rmtree(f"{d}/o_{name}", ignore_errors=True)
shutil.rmtree(d, ignore_errors=True)
