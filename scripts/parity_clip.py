"""Whole-clip parity of the recurrent pipeline against the CPU oracle ("teacher-forced per frame AND
free-running over the full clip, both reported").  TEST INFRASTRUCTURE: run on the GPU box; it drives the product through the
C ABI (fav_amd.Stream) and the checker (oracle/) side by side and never mixes them.

  teacher-forced : the oracle computes frame i from the GPU's own frame i-1 (isolates the error of ONE step)
  free-running   : the oracle runs its own recurrent chain from frame 1 (accumulated drift over the clip)

Modes:
  cert  (BASELINE config 2) : masks are precomputed files written by the REFERENCE's own consistencyChecker
                              (oracle/_ref/consistencyChecker <backward.flo> <forward.flo> <out.pgm> <frame.ppm>, the 4-argument call
                              of makeOptFlow_deepflow.sh:59-60), consumed through fav_stream_next_frame_cert
  flow3 (BASELINE config 3) : the 3-argument check fused on the GPU (fav_stream_next_frame_flow); the oracle's mask must be
                              byte-identical

    python scripts/parity_clip.py --config 3 --frames 60 --out gpurun_out/parity_c3_freerun.json

Whole clip on the contractive checkpoint (`--gain`, t7.build_model's recurrent_gain: frame -> frame then contracts, so free-running
parity is a gate over ALL frames).  A 1280x720 oracle frame costs ~3 s on the GPU box's 16 CPUs (300 frames = 15 of the round's 90
GPU-minutes spent waiting for a CPU), so config 3's 300 frames are split across the two machines -- both chains are deterministic
functions of the seeded clip:
    GPU box :  python scripts/parity_clip.py --config 3 --frames 300 --gain 0.05 --gpu-dump gpurun_out/c3_contractive_gpu.npz
               (the GPU's free-running chain: 4096 fixed sample positions x 3 channels of EVERY frame as fp32, the full fp32 frame
                at four check points, the mask's byte sum of every frame)
    anywhere:  python scripts/parity_clip.py --config 3 --frames 300 --gain 0.05 --compare gpurun_out/c3_contractive_gpu.npz \
                      --out profiles/parity_c3_freerun_contractive.json
               (the oracle's own free-running chain on the same clip, compared at those positions / frames)
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-artistic-videos_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


CONTRACTIVE_GAIN = 0.05      # recurrent_gain of the contractive synthetic checkpoint (oracle-vs-perturbed-oracle control: x0.45-0.6 per frame
                             # down to the fp32 noise floor at 640x360; tests/test_cpu_oracle.py gates the contraction)


def psnr8(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else float(10 * np.log10(255.0 ** 2 / mse))


def _f01(u8):
    return np.transpose(u8, (2, 0, 1)).astype(np.float32) / np.float32(255)


def make_pool(h, w, seed, pool):
    """`pool` distinct seeded frames / flow pairs (SURVEY 8d synthetic inputs); a clip cycles through them with a stride that
    changes every lap, so consecutive (frame, flow) pairings do not repeat before pool^2 frames"""
    from fav_amd import synth
    frames = [synth.smooth_frame(h, w, seed + i) for i in range(pool)]
    bws = [synth.backward_flow(h, w, seed + 100 + i) for i in range(pool)]
    fws = [synth.forward_flow_from_backward(bws[i], seed + 200 + i) for i in range(pool)]
    return frames, bws, fws


def run_clip(fav, O, model_path, h, w, n_frames, mode="flow3", seed=1000, pool=8, teacher=True, free=True, budget_s=None,
             threads=None, log=print):
    import torch
    from fav_amd import t7
    dev = torch.device("cuda:0")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    layers = t7.extract_layers(t7.load(model_path)["model"])
    frames, bws, fws = make_pool(h, w, seed, pool)
    if threads:
        O.set_threads(threads)
    net = fav.Net(model_path, 0)
    st = fav.Stream(net, h, w)
    tmp = tempfile.mkdtemp(prefix="parity_clip_")
    rows = []
    prior_influence = None
    t_start = time.time()
    ref_free = O.Stylizer(layers) if free else None
    prev_gpu = None
    for i in range(n_frames):
        lap, k = divmod(i, pool)
        fi, gi = k, (k * (2 * lap + 1) + lap) % pool
        frame, bw, fw = frames[fi], bws[gi], fws[gi]
        row = {"frame": i + 1}
        if i == 0:
            o, u = st.first_frame(T(frame), want_u8=True)
            mask = None
        elif mode == "cert":
            a, b, p, m = (os.path.join(tmp, x) for x in ("bw.flo", "fw.flo", "f.ppm", "rel.pgm"))
            O.write_flo(a, bw); O.write_flo(b, fw); O.write_pnm(p, frame)
            subprocess.check_call([O.REF_CHECKER, a, b, m, p], stdout=subprocess.DEVNULL)     # the reference's own binary
            mask = O.read_pnm(m)
            o, u = st.next_frame_cert(T(frame), T(bw), T(mask), want_u8=True)
        else:
            o, u = st.next_frame_flow(T(frame), T(bw), T(fw), want_u8=True)
            mask = O.consistency(bw, fw)
            row["mask_mismatch_bytes"] = int((st.last_mask().cpu().numpy() != mask).sum())
        net.check()
        g = o.cpu().numpy(); g8 = u.cpu().numpy()
        if mask is not None:
            row["reliable_pct"] = round(float((mask == 255).mean() * 100), 2)
        cert01 = None if mask is None else mask.astype(np.float32) / np.float32(255)
        if teacher:
            tf = O.Stylizer(layers)
            if i == 0:
                r = tf.first(_f01(frame))
            else:
                tf.last = prev_gpu
                r = tf.next(_f01(frame), bw, cert01)
            row["teacher_max_abs"] = float(np.abs(g - r).max()); row["teacher_psnr8_db"] = round(psnr8(g8, O.to_u8_hwc(r)), 2)
        if free:
            if i == 1:      # how much the recurrent inputs matter: the same step with an all-occluded prior (one extra oracle frame)
                probe = O.Stylizer(layers); probe.last = ref_free.last; probe.count = ref_free.count
                r_np = probe.next(_f01(frame), bw, np.zeros_like(cert01))
            r = ref_free.first(_f01(frame)) if i == 0 else ref_free.next(_f01(frame), bw, cert01)
            if i == 1:
                prior_influence = float(np.abs(r - r_np).max())
            row["free_max_abs"] = float(np.abs(g - r).max()); row["free_psnr8_db"] = round(psnr8(g8, O.to_u8_hwc(r)), 2)
            row["free_rms"] = float(np.sqrt(np.mean((g.astype(np.float64) - r) ** 2)))
        prev_gpu = g
        rows.append(row)
        log(json.dumps(row))
        if budget_s and time.time() - t_start > budget_s:
            log(f"time budget of {budget_s} s reached after {i + 1} frames")
            break
    out = {"model": os.path.basename(model_path), "H": h, "W": w, "mode": mode, "frames_compared": len(rows),
           "frames_requested": n_frames, "pool": pool, "seed": seed, "seconds": round(time.time() - t_start, 1), "per_frame": rows}
    for key in ("teacher_max_abs", "free_max_abs"):
        v = [r[key] for r in rows if key in r]
        if v:
            out[key + "_worst"] = max(v); out[key + "_last"] = v[-1]
    for key in ("teacher_psnr8_db", "free_psnr8_db"):
        v = [r[key] for r in rows if key in r]
        if v:
            out[key + "_min"] = min(v); out[key + "_mean"] = round(float(np.mean(v)), 2)
    mm = [r["mask_mismatch_bytes"] for r in rows if "mask_mismatch_bytes" in r]
    if mm:
        out["mask_mismatch_bytes_total"] = int(sum(mm))
    if prior_influence is not None:
        out["prior_influence_max_abs"] = prior_influence      # frame 2 with vs without its prior (oracle): the recurrent path is live
    return out


def clip_inputs(frames, bws, fws, pool, i):
    lap, k = divmod(i, pool)
    fi, gi = k, (k * (2 * lap + 1) + lap) % pool
    return frames[fi], bws[gi], fws[gi]


SAMPLES = 4096
CHECKPOINTS = (1, 100, 200, -1)          # frames kept whole (1-based; -1 = the last)


def sample_positions(h, w):
    rng = np.random.default_rng(4242)
    return rng.integers(0, h, SAMPLES), rng.integers(0, w, SAMPLES)


def gpu_dump(fav, model_path, h, w, n_frames, path, seed=1000, pool=8, log=print):
    """the GPU's free-running chain over the whole clip, reduced to what fits gpurun_out (see the module docstring)"""
    import torch
    dev = torch.device("cuda:0")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    frames, bws, fws = make_pool(h, w, seed, pool)
    dfr, dbw, dfw = [T(a) for a in frames], [T(a) for a in bws], [T(a) for a in fws]
    ys, xs = sample_positions(h, w)
    tys, txs = torch.from_numpy(ys).to(dev), torch.from_numpy(xs).to(dev)
    net = fav.Net(model_path, 0)
    st = fav.Stream(net, h, w)
    samples = np.zeros((n_frames, 3, SAMPLES), np.float32)
    mask_sums = np.zeros(n_frames, np.int64)
    mask_sha = [""] * n_frames         # SHA-256 of every mask's H*W bytes: 300 whole masks (276 MB) do not fit gpurun_out, their digests do
    full = {}
    cps = {c if c > 0 else n_frames for c in CHECKPOINTS if c <= n_frames}
    for i in range(n_frames):
        fi, bi, wi = clip_inputs(range(pool), range(pool), range(pool), pool, i)
        if i == 0:
            o, _ = st.first_frame(dfr[fi])
        else:
            o, _ = st.next_frame_flow(dfr[fi], dbw[bi], dfw[wi])
            mk = st.last_mask().cpu().numpy()
            mask_sums[i] = int(mk.astype(np.int64).sum())
            mask_sha[i] = hashlib.sha256(np.ascontiguousarray(mk).tobytes()).hexdigest()
        samples[i] = o[:, tys, txs].cpu().numpy()
        if i + 1 in cps:
            full[f"full_{i + 1}"] = o.cpu().numpy()
    net.check()
    np.savez(path, samples=samples, mask_sums=mask_sums, mask_sha256=np.array(mask_sha), h=h, w=w, seed=seed, pool=pool, frames=n_frames, **full)
    log(f"wrote {path}: {n_frames} frames, {len(full)} whole frames")


def compare_dump(O, model_path, path, threads=None, log=print):
    """the oracle's own free-running chain on the same clip against the GPU's dumped chain"""
    from fav_amd import t7
    d = np.load(path)
    h, w, seed, pool, n_frames = int(d["h"]), int(d["w"]), int(d["seed"]), int(d["pool"]), int(d["frames"])
    layers = t7.extract_layers(t7.load(model_path)["model"])
    frames, bws, fws = make_pool(h, w, seed, pool)
    if threads:
        O.set_threads(threads)
    ys, xs = sample_positions(h, w)
    ref = O.Stylizer(layers)
    rows = []
    t0 = time.time()
    for i in range(n_frames):
        frame, bw, fw = clip_inputs(frames, bws, fws, pool, i)
        row = {"frame": i + 1}
        if i == 0:
            r = ref.first(_f01(frame))
        else:
            mask = O.consistency(bw, fw)
            row["mask_byte_sum_equal"] = bool(int(mask.astype(np.int64).sum()) == int(d["mask_sums"][i]))
            if "mask_sha256" in d.files:      # every byte of the mask, through its digest (dumps older than round 5 carry the byte sums only)
                row["mask_bytes_equal"] = bool(hashlib.sha256(np.ascontiguousarray(mask).tobytes()).hexdigest() == str(d["mask_sha256"][i]))
            r = ref.next(_f01(frame), bw, mask.astype(np.float32) / np.float32(255))
        g = d["samples"][i]
        diff = np.abs(g - r[:, ys, xs])
        row["sampled_max_abs"] = float(diff.max()); row["sampled_rms"] = float(np.sqrt(np.mean(diff.astype(np.float64) ** 2)))
        key = f"full_{i + 1}"
        if key in d.files:
            gf = d[key]
            row["full_max_abs"] = float(np.abs(gf - r).max()); row["full_psnr8_db"] = round(psnr8(O.to_u8_hwc(gf), O.to_u8_hwc(r)), 2)
        rows.append(row); log(json.dumps(row))
    out = {"what": "free-running over the whole clip: the GPU's chain (run on the MI355X, dumped) against the oracle's own chain on the same seeded clip; "
                   "every frame at %d fixed sample positions x 3 channels, whole frames at the check points" % SAMPLES,
           "model": os.path.basename(model_path), "H": h, "W": w, "mode": "flow3", "frames_compared": len(rows), "pool": pool, "seed": seed,
           "seconds_oracle": round(time.time() - t0, 1), "per_frame": rows,
           "sampled_max_abs_worst": max(r["sampled_max_abs"] for r in rows), "sampled_max_abs_last": rows[-1]["sampled_max_abs"],
           "full_frames": {str(r["frame"]): {"max_abs": r["full_max_abs"], "psnr8_db": r["full_psnr8_db"]} for r in rows if "full_max_abs" in r},
           "mask_byte_sums_equal_on_all_frames": all(r.get("mask_byte_sum_equal", True) for r in rows),
           "mask_bytes_equal_on_all_frames": (all(r["mask_bytes_equal"] for r in rows if "mask_bytes_equal" in r) if any("mask_bytes_equal" in r for r in rows) else None),
           "mask_comparison": "SHA-256 of the H*W mask bytes of every frame (GPU side computed on the box, oracle side here): equal digests = byte-for-byte equal masks",
           "gate": "2e-4 de-processed / 50 dB (BASELINE.md section 4)"}
    out["within_gate"] = bool(out["sampled_max_abs_worst"] <= 2e-4 and all(v["max_abs"] <= 2e-4 and v["psnr8_db"] >= 50 for v in out["full_frames"].values())
                              and out["mask_byte_sums_equal_on_all_frames"] and out["mask_bytes_equal_on_all_frames"] is not False)
    return out


def effective_cpus():
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except (OSError, ValueError):
        pass
    return n


def run_control(O, model_path, h, w, n_frames, mode="cert", seed=1000, pool=8, eps=5e-6, threads=None, log=print):
    """CPU-only control for the free-running numbers: TWO oracle chains on the same clip, the second one started from the first
    chain's frame 1 plus uniform noise of +-eps (the size of one step's GPU-vs-oracle error).  If their distance grows like the
    GPU-vs-oracle distance does, the growth is the sensitivity of the recurrent map itself (synthetic, random-init weights), not an
    implementation error."""
    from fav_amd import t7
    layers = t7.extract_layers(t7.load(model_path)["model"])
    frames, bws, fws = make_pool(h, w, seed, pool)
    if threads:
        O.set_threads(threads)
    a, b = O.Stylizer(layers), O.Stylizer(layers)
    rng = np.random.default_rng(99)
    rows = []
    for i in range(n_frames):
        lap, k = divmod(i, pool)
        fi, gi = k, (k * (2 * lap + 1) + lap) % pool
        frame, bw, fw = frames[fi], bws[gi], fws[gi]
        if i == 0:
            ra = a.first(_f01(frame))
            rb = (ra + rng.uniform(-eps, eps, ra.shape)).astype(np.float32)
            b.last = rb; b.count = a.count
        else:
            mask = O.consistency(bw, fw, frame if mode == "cert" else None)
            cert01 = mask.astype(np.float32) / np.float32(255)
            ra = a.next(_f01(frame), bw, cert01); rb = b.next(_f01(frame), bw, cert01)
        row = {"frame": i + 1, "max_abs": float(np.abs(ra - rb).max()), "rms": float(np.sqrt(np.mean((ra.astype(np.float64) - rb) ** 2))),
               "psnr8_db": round(psnr8(O.to_u8_hwc(ra), O.to_u8_hwc(rb)), 2)}
        rows.append(row); log(json.dumps(row))
    growth = [rows[k + 1]["rms"] / rows[k]["rms"] for k in range(min(8, len(rows) - 1)) if rows[k]["rms"] > 0]
    return {"what": "oracle vs oracle with frame 1 perturbed by uniform noise of +-%g (no GPU involved)" % eps, "H": h, "W": w, "mode": mode, "seed": seed,
            "frames": len(rows), "rms_growth_per_frame_first_8": [round(g, 2) for g in growth],
            "rms_growth_geomean": round(float(np.exp(np.mean(np.log(growth)))), 3) if growth else None, "per_frame": rows}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--control", action="store_true", help="CPU-only: oracle vs perturbed oracle on the same clip (sensitivity of the recurrent map)")
    ap.add_argument("--config", type=int, default=3, help="BASELINE config: 2 = 640x360 cert path, 3 = 1280x720 fused 3-argument check")
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--no-teacher", action="store_true")
    ap.add_argument("--budget-s", type=float, default=None)
    ap.add_argument("--threads", type=int, default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--gain", type=float, default=1.0, help="recurrent_gain of the synthetic checkpoint (%g = the contractive one)" % CONTRACTIVE_GAIN)
    ap.add_argument("--gpu-dump", default=None, help="GPU side of the split whole-clip run: write the GPU chain's samples to this .npz")
    ap.add_argument("--compare", default=None, help="oracle side of the split whole-clip run: compare against this .npz")
    a = ap.parse_args()
    import oracle as O
    from fav_amd import t7
    fav_amd = None
    if not a.control and not a.compare:
        import fav_amd
    O.build()
    model = os.path.join(tempfile.mkdtemp(), "canonical.t7")
    t7.make_synthetic_checkpoint(model, seed=3, recurrent_gain=a.gain)
    if a.config == 2:
        h, w, n, mode = 360, 640, a.frames or 32, "cert"
    else:
        h, w, n, mode = 720, 1280, a.frames or 60, "flow3"
    nthreads = a.threads or effective_cpus()        # (the GPU box shows 256 hardware threads under a 16-CPU quota)
    if a.gpu_dump:
        os.makedirs(os.path.dirname(os.path.abspath(a.gpu_dump)), exist_ok=True)
        gpu_dump(fav_amd, model, h, w, n, a.gpu_dump)
        return
    if a.compare:
        res = compare_dump(O, model, a.compare, threads=nthreads, log=lambda s_: print(s_, flush=True))
        res["recurrent_gain"] = a.gain
    elif a.control:
        res = run_control(O, model, h, w, n, mode=mode, threads=nthreads)
    else:
        res = run_clip(fav_amd, O, model, h, w, n, mode=mode, teacher=not a.no_teacher, budget_s=a.budget_s, threads=nthreads)
    res["baseline_config"] = a.config
    res["host_threads"] = nthreads
    res.setdefault("recurrent_gain", a.gain)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "per_frame"}))


if __name__ == "__main__":
    main()
