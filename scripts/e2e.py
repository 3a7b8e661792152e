#!/usr/bin/env python3
"""End-to-end CLI throughput on RAM-backed files (/dev/shm): P6 frames + .flo in, PNG out, through bin/fav_stylize.
Not the bench metric (that excludes file I/O and PCIe); reported in DESIGN.md next to it."""
import json, os, subprocess, sys, time, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-artistic-videos_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle as O
from fav_amd import synth, t7
H, W, N = 720, 1280, int(sys.argv[1]) if len(sys.argv) > 1 else 60
d = "/dev/shm/fav_e2e"; shutil.rmtree(d, ignore_errors=True); os.makedirs(d + "/flow"); os.makedirs(d + "/src")
model = d + "/canonical.t7"; t7.make_synthetic_checkpoint(model, seed=1234)
for k in range(4):
    O.write_pnm(f"{d}/src/f{k}.ppm", synth.random_frame(H, W, k))
    bw = synth.backward_flow(H, W, 10 + k); O.write_flo(f"{d}/src/b{k}.flo", bw); O.write_flo(f"{d}/src/w{k}.flo", synth.forward_flow_from_backward(bw, 20 + k))
chk = os.path.join(ROOT, "fast-artistic-videos_amd", "bin", "consistencyChecker")
t0 = time.time()
for k in range(4):
    subprocess.check_call([chk, f"{d}/src/b{k}.flo", f"{d}/src/w{k}.flo", f"{d}/src/r{k}.pgm", f"{d}/src/f{k}.ppm"], stdout=subprocess.DEVNULL)
t_chk = (time.time() - t0) / 4
for i in range(1, N + 1):
    os.symlink(f"{d}/src/f{i % 4}.ppm", f"{d}/frame_{i:05d}.ppm")
    if i > 1:
        os.symlink(f"{d}/src/b{i % 4}.flo", f"{d}/flow/backward_{i}_{i-1}.flo"); os.symlink(f"{d}/src/w{i % 4}.flo", f"{d}/flow/forward_{i-1}_{i}.flo")
        os.symlink(f"{d}/src/r{i % 4}.pgm", f"{d}/flow/reliable_{i}_{i-1}.pgm")
exe = os.path.join(ROOT, "fast-artistic-videos_amd", "bin", "fav_stylize")
base = [exe, "-input_pattern", d + "/frame_%05d.ppm", "-flow_pattern", d + "/flow/backward_[%d]_{%d}.flo", "-occlusions_pattern", d + "/flow/reliable_[%d]_{%d}.pgm",
        "-model_vid", model, "-model_img", "self", "-gpu", "0", "-timing", "1", "-writers", os.environ.get("FAV_E2E_WRITERS", "16")]
res = {"frames": N, "consistencyChecker_process_s_per_pair_4arg": round(t_chk, 4)}
for name, extra in [("cert_mode_png1", ["-output_prefix", d + "/o1/out"]), ("cert_mode_png0", ["-output_prefix", d + "/o2/out", "-png_level", "0"]),
                    ("fused_3arg_png1", ["-output_prefix", d + "/o3/out", "-forward_flow_pattern", d + "/flow/forward_{%d}_[%d].flo", "-structure", "0"]),
                    ("fused_4arg_png1", ["-output_prefix", d + "/o4/out", "-forward_flow_pattern", d + "/flow/forward_{%d}_[%d].flo", "-structure", "1"])]:
    r = subprocess.run(base + extra, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    res[name] = json.loads(line[-1])["fps_end_to_end"] if line else ("FAILED: " + r.stderr[-200:])
    if line and os.environ.get("FAV_E2E_VERBOSE"): res[name + "_detail"] = json.loads(line[-1])
print(json.dumps(res))
shutil.rmtree(d, ignore_errors=True)
