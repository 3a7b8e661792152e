"""Frame rate of the per-frame hot path on a checkpoint with MORE FILTERS than the canonical one (the reference builds the network from
any architecture string, models_video.lua:55-140; its published VR checkpoints "have more filters", README.md:141): the same step as
bench.py's `value` -- fused 3-argument check + certainty erosion + warp + assembly + network + de-process + PNG encode, inputs in HBM --
at 1280x720 with synthetic weights.  bench.py imports run() for its `extra.wide_arch_*` keys; run directly (under rocprofv3 for the
kernel statistics) it prints one JSON line.
usage: python scripts/wide_bench.py [--arch c9s1-64,d128,...] [--steps 12] [--warmup 3] [--size 720x1280]"""
import argparse, json, os, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-artistic-videos_amd", "python"))

WIDE2 = "c9s1-64,d128,d256,R256,R256,R256,R256,R256,U2,c3s1-128,U2,c9s1-3"      # every filter count of the canonical string doubled
WIDE15 = "c9s1-48,d96,d192,R192,R192,R192,R192,R192,U2,c3s1-96,U2,c9s1-3"      # x1.5 (runs zero-padded to 64 / 128 / 256)
FP32_MFMA_PEAK_TFLOPS = 157.3

# kernel ids of fav_net::timed_conv that are NOT the layer's minimal-filtering / fragment-order kernel (the generic implicit GEMM
# reports its N tile: 32 / 64 / 128; 2xx / 3xx = the round-1/2 halo-resident direct forms; 7 / 8 = the direct first layer)
FALLBACK_IDS = {32, 64, 128, 7, 8}


def run(arch=WIDE2, steps=12, warmup=3, h=720, w=1280, frames=None, bws=None, fws=None, seed=4321):
    import numpy as np
    import torch
    import fav_amd
    from fav_amd import synth, t7
    dev = torch.device("cuda:%d" % torch.cuda.current_device())
    with tempfile.TemporaryDirectory() as d:
        ck = os.path.join(d, "wide.t7")
        t7.make_synthetic_checkpoint(ck, arch=arch, seed=seed)
        net = fav_amd.Net(ck, dev.index or 0)
    if frames is None:
        ring = 3
        frames_h = [synth.random_frame(h, w, seed + i) for i in range(ring)]
        bw_h = [synth.backward_flow(h, w, seed + 10 + i) for i in range(ring)]
        fw_h = [synth.forward_flow_from_backward(bw_h[i], seed + 20 + i) for i in range(ring)]
        frames = [torch.from_numpy(a).to(dev) for a in frames_h]
        bws = [torch.from_numpy(a).to(dev) for a in bw_h]
        fws = [torch.from_numpy(a).to(dev) for a in fw_h]
    ring = len(frames)
    stream = fav_amd.Stream(net, h, w)
    png_out, png_n = stream.png_buffers()

    def step(i):
        k = i % ring
        stream.next_frame_flow(frames[k], bws[k], fws[k], use_structure=False, want_f32=False, want_u8=False)
        stream.encode_png_into(png_out, png_n)

    stream.first_frame(frames[0], want_f32=False, want_u8=False)
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # a few more steps with HIP events around every convolution (outside the timed loop)
    net.profile_enable(True)
    nprof = 4
    for i in range(nprof):
        step(i)
    torch.cuda.synchronize()
    net.profile_enable(False)
    net.check()
    prof = net.profile_read()
    conv_ms = sum(ms for ms, n, macs, kid in prof) / nprof
    flop = sum(2.0 * macs * n for ms, n, macs, kid in prof) / nprof
    per_kernel = {}
    for ms, n, macs, kid in prof:
        if n:
            k = per_kernel.setdefault(str(kid), [0.0, 0.0]); k[0] += ms / nprof; k[1] += 2.0 * macs * n / nprof
    ids = [kid for ms, n, macs, kid in prof]
    out = {"arch": arch, "frame": [w, h], "steps": steps, "frames_per_s": round(steps / dt, 3), "ms_per_frame": round(dt / steps * 1e3, 4),
           "conv_gflop_per_frame": round(flop / 1e9, 2), "conv_stack_ms_per_frame": round(conv_ms, 4),
           "conv_stack_algorithmic_tflops": round(flop / (conv_ms * 1e-3) / 1e12, 2) if conv_ms > 0 else None,
           "conv_stack_algorithmic_frac_of_fp32_mfma_peak": round(flop / (conv_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4) if conv_ms > 0 else None,
           "kernel_ids": ids, "fallback_layers": [i for i, k in enumerate(ids) if k in FALLBACK_IDS or 200 <= k < 400],
           "per_kernel_ms_tflops": {k: [round(v[0], 4), round(v[1] / (v[0] * 1e-3) / 1e12, 1)] for k, v in per_kernel.items()},
           "png_bytes": int(png_n.item())}
    del stream, net
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default=WIDE2)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", default="720x1280")
    a = ap.parse_args()
    arch = {"2x": WIDE2, "1.5x": WIDE15}.get(a.arch, a.arch)
    hh, ww = (int(v) for v in a.size.split("x"))
    print(json.dumps(run(arch, a.steps, a.warmup, hh, ww)))
