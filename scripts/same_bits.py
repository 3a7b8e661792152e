#!/usr/bin/env python3
"""Are two builds of libfav bit-identical on a clip?  usage: same_bits.py <a.so> <b.so> [frames]
Runs the canonical network on a seeded 1280x720 clip (first frame + recurrent frames in the checker's 4- and 3-argument modes, PNG encode
included) once per library, each in its own process (FAV_AMD_LIB), and compares SHA-256 of every frame's float output, mask and PNG."""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json, hashlib
sys.path.insert(0, os.path.join(%r, "fast-artistic-videos_amd", "python"))
import numpy as np, torch
import fav_amd
from fav_amd import synth, t7
H, W, N = 720, 1280, int(sys.argv[1])
ck = sys.argv[2]
dev = torch.device("cuda:0")
net = fav_amd.Net(ck, 0)
out = []
for structure in (1, 0):
    st = fav_amd.Stream(net, H, W)
    png_out, png_n = st.png_buffers()
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    h = lambda t: hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:16]
    f0 = synth.smooth_frame(H, W, 900)
    o, _ = st.first_frame(T(f0)); st.encode_png_into(png_out, png_n)
    out.append([structure, 0, h(o), "", h(png_out[: int(png_n.item())])])
    for i in range(1, N):
        fr = synth.smooth_frame(H, W, 900 + i); bw = synth.backward_flow(H, W, 950 + i); fw = synth.forward_flow_from_backward(bw, 990 + i)
        o, _ = st.next_frame_flow(T(fr), T(bw), T(fw), use_structure=bool(structure)); st.encode_png_into(png_out, png_n)
        out.append([structure, i, h(o), h(st.last_mask()), h(png_out[: int(png_n.item())])])
    del st
net.check()
print("HASHES " + json.dumps(out))
''' % ROOT
def run(lib, n, ck):
    env = dict(os.environ, FAV_AMD_LIB=os.path.abspath(lib))
    r = subprocess.run([sys.executable, "-c", CHILD, str(n), ck], env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("HASHES ")]
    if not line: raise SystemExit("child failed for %s:\n%s" % (lib, r.stderr[-2000:]))
    return json.loads(line[-1][7:])
if __name__ == "__main__":
    a, b = sys.argv[1], sys.argv[2]; n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    sys.path.insert(0, os.path.join(ROOT, "fast-artistic-videos_amd", "python"))
    from fav_amd import t7
    ck = "/tmp/same_bits_canonical.t7"; t7.make_synthetic_checkpoint(ck, seed=4321)
    ha, hb = run(a, n, ck), run(b, n, ck)
    same = ha == hb
    for x, y in zip(ha, hb):
        print("mode %s frame %d  out %s %s  mask %s %s  png %s %s  %s" % ("4-arg" if x[0] else "3-arg", x[1], x[2], y[2], x[3], y[3], x[4], y[4], "same" if x == y else "DIFFERENT"))
    print("RESULT: %s and %s are %s on %d frames x 2 checker modes (float output, mask, PNG bytes)" % (os.path.basename(a), os.path.basename(b), "BIT-IDENTICAL" if same else "NOT identical", n))
    sys.exit(0 if same else 1)
