#!/usr/bin/env python3
"""bench.py -- stylised frames/sec of the per-frame hot path at 1280x720 on N MI355X (one node).

One "step" = one pass of the hot path over one frame of one video stream:
    on-GPU consistency check (forward+backward flow) -> certainty erosion -> warp of the previous
    stylised frame -> 7-channel assembly + reflection pad -> transformer network -> de-process
with all inputs (uint8 frame, backward .flo payload, forward .flo payload, previous output) already
resident in HBM (BASELINE.json configs[2]: "1280x720 x 300 frames, on-GPU warp + consistencyChecker +
net fused").  Each rank owns one independent video stream (the path shards across streams only: frame i
needs frame i-1's output), so N GPUs = N streams, weak scaling, no data-path collective; the only
collective is the RCCL broadcast of the packed weight blob from rank 0 before the timed region.

Launch:  python bench.py [--gpus 1] [--steps K] [--warmup W]
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                --master-port P bench.py --gpus N --steps K --warmup W
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "fast-artistic-videos_amd", "python"))

H, W = 720, 1280
FLOP_PER_FRAME = 305_651_220_480          # useful conv FLOPs of the canonical net at 1280x720
FP32_MFMA_PEAK_TFLOPS = 157.3             # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def effective_cpus():
    """usable CPUs: scheduler affinity capped by the cgroup CPU quota (the GPU box shows 256 threads under a 16-CPU quota)"""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline_and_parity(layers, frame1, bw, fw, prev_state, gpu_out, gpu_out_u8, gpu_mask):
    """The oracle (a port of the reference's CPU path, fast_artistic_video_core.lua:161-180) timed on this host's cores on ONE
    1280x720 recurrent step (min of 3 runs), and -- since that frame is computed anyway -- compared with what the GPU path
    produced from the same inputs (teacher-forced: both start from the GPU's previous stylised frame).  Outside the timed
    region; the oracle is the checker here, never the thing measured as `value`."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle as O
    O.build()
    cores = effective_cpus()
    O.set_threads(cores)            # one OpenMP thread per usable CPU (not per visible hardware thread)
    f1 = np.transpose(frame1, (2, 0, 1)).astype(np.float32) / np.float32(255)
    b, f = np.ascontiguousarray(bw), np.ascontiguousarray(fw)
    times = []
    for _ in range(3):
        st = O.Stylizer(layers)
        st.last = prev_state
        t0 = time.perf_counter()
        mask = O.consistency(b, f)
        r1 = st.next(f1, b, mask.astype(np.float32) / np.float32(255))
        times.append(time.perf_counter() - t0)
    dt = min(times)
    base = {"value": round(1.0 / dt, 5), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"1 frame of 1280x720 (3-arg mask + min-filter + warp + assemble + net + deprocess) through oracle/ "
                      f"(C, fp64 accumulation, OpenMP) on {cores} threads (= usable CPUs: {len(os.sched_getaffinity(0))} visible, cgroup quota applied), min of 3 runs ({', '.join('%.2f' % t for t in times)} s)"}
    ref_u8 = O.to_u8_hwc(r1)
    mse = float(np.mean((ref_u8.astype(np.float64) - gpu_out_u8.astype(np.float64)) ** 2))
    max_abs = float(np.abs(r1 - gpu_out).max())
    parity = {"psnr_db": 99.0 if mse == 0 else round(10 * np.log10(255.0 ** 2 / mse), 2),
              "max_abs": float("%.3e" % max_abs), "max_abs_tanh150_space": float("%.3e" % (max_abs * 255.0)),
              "mask_mismatch_bytes": int((mask != gpu_mask).sum()),
              "vs": "oracle/ (CPU restatement) on the same 1280x720 inputs, one recurrent step, teacher-forced from the GPU's previous frame; "
                    "max_abs in de-processed [0,1] units (gate 2e-4), PSNR of the 8-bit frames (gate 50 dB), mask bytes (gate 0)"}
    return base, parity


def reference_checker_baseline(bw, fw, frame):
    """The reference's OWN consistencyChecker (compiled by oracle/Makefile into oracle/_ref, 1 thread as shipped), timed per
    1280x720 flow pair including its file I/O on RAM-backed files; min of 3."""
    exe = os.path.join(ROOT, "oracle", "_ref", "consistencyChecker")
    if not os.path.exists(exe):
        return None
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    d = tempfile.mkdtemp(prefix="fav_refchk_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        a, b, i, o = (os.path.join(d, n) for n in ("bw.flo", "fw.flo", "img.ppm", "out.pgm"))
        O.write_flo(a, bw); O.write_flo(b, fw); O.write_pnm(i, frame)
        res = {}
        for name, args in (("3arg", [exe, a, b, o]), ("4arg", [exe, a, b, o, i])):
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); subprocess.check_call(args, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL); ts.append(time.perf_counter() - t0)
            res[name] = min(ts)
        return {"kind": "reference", "cores": 1, "unit": "masks/s", "value": round(1.0 / res["3arg"], 3), "value_4arg": round(1.0 / res["4arg"], 3),
                "sample": "consistencyChecker (reference sources, g++ -O3) on one 1280x720 flow pair incl. .flo read + .pgm write on /dev/shm, "
                          "min of 3: %.3f s (3-arg) / %.3f s (4-arg, image structure)" % (res["3arg"], res["4arg"])}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def e2e_block(ckpt, frames_h, bw_h, fw_h, nframes=300):
    """File -> PNG rate of the drop-in CLI (fast_artistic_video.lua:93-97,160-170): bin/fav_stylize over `nframes` RAM-backed
    1280x720 P6 frames + backward/forward .flo, fused on-GPU 3-argument check, PNGs written back to RAM.  Everything the
    in-HBM `value` leaves out is inside: decode, H2D, D2H, deflate, file writes."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    exe = os.path.join(ROOT, "fast-artistic-videos_amd", "bin", "fav_stylize")
    if not os.path.exists(exe) or not os.path.isdir("/dev/shm"):
        return {"error": "bin/fav_stylize or /dev/shm missing"}
    d = tempfile.mkdtemp(prefix="fav_e2e_", dir="/dev/shm")
    try:
        os.makedirs(d + "/src"); os.makedirs(d + "/flow")
        ring = len(frames_h)
        for k in range(ring):
            O.write_pnm(f"{d}/src/f{k}.ppm", frames_h[k]); O.write_flo(f"{d}/src/b{k}.flo", bw_h[k]); O.write_flo(f"{d}/src/w{k}.flo", fw_h[k])
        for i in range(1, nframes + 1):
            os.symlink(f"{d}/src/f{i % ring}.ppm", f"{d}/frame_{i:05d}.ppm")
            if i > 1:
                os.symlink(f"{d}/src/b{i % ring}.flo", f"{d}/flow/backward_{i}_{i-1}.flo"); os.symlink(f"{d}/src/w{i % ring}.flo", f"{d}/flow/forward_{i-1}_{i}.flo")
        base = [exe, "-input_pattern", d + "/frame_%05d.ppm", "-flow_pattern", d + "/flow/backward_[%d]_{%d}.flo",
                "-forward_flow_pattern", d + "/flow/forward_{%d}_[%d].flo",
                "-model_vid", ckpt, "-model_img", "self", "-gpu", "0", "-timing", "1"]
        out = {"frames": nframes, "host_threads": os.cpu_count(), "usable_cpus": effective_cpus(), "pipeline": "P6 + 2 x .flo from /dev/shm -> H2D -> fused 3-arg check + warp + net -> D2H -> PNG to /dev/shm",
               "h2d_bytes_per_frame": H * W * (3 + 8 + 8), "d2h_bytes_per_frame": H * W * 3}
        # 3-argument check = the workload of `value`; the 4-argument (image-structure) check is what makeOptFlow_deepflow.sh:59 runs
        for name, lvl, structure in (("png_level_1", "1", "0"), ("png_level_0", "0", "0"), ("png_level_1_4arg_check", "1", "1")):
            r = subprocess.run(base + ["-structure", structure, "-output_prefix", f"{d}/o{lvl}/out", "-png_level", lvl], capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                out[name] = {"error": (r.stderr or r.stdout)[-300:]}
                continue
            j = json.loads(line[-1])
            n_png = len([f for f in os.listdir(f"{d}/o{lvl}") if f.endswith(".png")])
            out[name] = {"fps": j["fps_end_to_end"], "seconds": j["seconds"], "png_written": n_png, "png_writers": j.get("png_writers"),
                         "wait_loader_s": j["wait_loader_s"], "wait_png_pool_s": j["wait_png_pool_s"],
                             "setup_s": j.get("setup_s"), "png_tail_s": j.get("png_tail_s"),
                             "steady_state_fps": round(j["frames"] / max(1e-9, j["seconds"] - (j.get("setup_s") or 0.0) - (j.get("png_tail_s") or 0.0)), 3)}
            shutil.rmtree(f"{d}/o{lvl}", ignore_errors=True)
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--structure", type=int, default=0, help="1 = 4-argument (image-structure) checker mode")
    ap.add_argument("--no-profile", action="store_true", help="no per-convolution HIP events in the timed region (A/B of their cost; the roofline block is then empty)")
    ap.add_argument("--profile-every", type=int, default=4, help="HIP events around the convolutions on every n-th step of the timed region")
    ap.add_argument("--lookahead", type=int, default=-1, help="1 = compute the masks of the next two frames on the side queues "
                    "(fav_stream_prefetch_mask: mask + certainty erosion, off the critical path); default: on for --structure 1 only")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU legs (oracle timing + parity block, reference checker)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the file->PNG run of bin/fav_stylize after the timed region")
    ap.add_argument("--force-dist", action="store_true", help="initialise the RCCL process group and run the weight broadcast even at "
                    "world size 1 (exercises the N>1 launch path on a 1-GPU box)")
    ap.add_argument("--no-extra", action="store_true", help="skip the informational 4-argument-mode pass after the timed region "
                    "(used for the rocprofv3 runs, so that the per-kernel averages cover the timed configuration only)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import fav_amd
    from fav_amd import synth, t7

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libfav has no CPU fallback")
    ndev = torch.cuda.device_count()
    if world > ndev:
        raise SystemExit(f"bench.py: {world} ranks but only {ndev} GPUs visible (one process per GPU: the persistent convolution "
                         "grids own the device, include/fav.h)")
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    cdev = dev                                                  # device of the collective buffers
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)          # RCCL over xGMI
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.lookahead < 0:
        args.lookahead = 1 if args.structure else 0      # (3-argument mode: 574 with, 582 frames/s without the look-ahead at r02ze; 544 / 548 at r02n -- scripts/ab_lookahead.sh)

    # ---- weights: rank 0 parses the (synthetic, canonical-architecture) .t7 and broadcasts the packed blob
    ckpt = os.path.join(tempfile.gettempdir(), f"fav_bench_canonical_{os.getpid()}.t7")
    from fav_amd import shard
    blob = None
    if rank == 0:
        t7.make_synthetic_checkpoint(ckpt, seed=1234)
        blob = fav_amd.pack_checkpoint(ckpt)
    t_b = time.perf_counter()
    blob = shard.broadcast_blob(blob, cdev, force=args.force_dist)   # RCCL over xGMI: 6.7 MB, once, before the timed region
    t_b = time.perf_counter() - t_b
    net = fav_amd.Net(blob=blob, device=local)
    stream = fav_amd.Stream(net, H, W)

    # ---- synthetic inputs, resident in HBM (seed = 1234 + stream id); a ring of distinct frames/flows
    ring = 4
    seed = 1234 + rank
    frames_h = [synth.random_frame(H, W, seed + i) for i in range(ring)]
    bw_h = [synth.backward_flow(H, W, seed + 10 + i) for i in range(ring)]
    fw_h = [synth.forward_flow_from_backward(bw_h[i], seed + 20 + i) for i in range(ring)]
    frames = [torch.from_numpy(a).to(dev) for a in frames_h]
    bws = [torch.from_numpy(a).to(dev) for a in bw_h]
    fws = [torch.from_numpy(a).to(dev) for a in fw_h]
    out8 = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)

    def step(i):
        k = i % ring
        if args.lookahead:      # the masks of frames i+1, i+2 depend only on inputs that are already resident: queue them
            k2 = (i + 2) % ring  # on the side queues BEFORE frame i's network so they overlap it (two in flight)
            stream.prefetch_mask(frames[k2], bws[k2], fws[k2], use_structure=bool(args.structure))
        stream.next_frame_flow(frames[k], bws[k], fws[k], use_structure=bool(args.structure), want_f32=False, out_u8=out8)

    stream.first_frame(frames[0], want_f32=False, out_u8=out8)
    if args.lookahead:
        stream.prefetch_mask(frames[0], bws[0], fws[0], use_structure=bool(args.structure))
        stream.prefetch_mask(frames[1 % ring], bws[1 % ring], fws[1 % ring], use_structure=bool(args.structure))
    for i in range(args.warmup):
        step(i)
    # HIP events around every convolution launch (the roofline block) on every `--profile-every`-th step of the timed region: the
    # events are not free -- with their default system-scope fence they cost 0.18 ms per frame (6 us on either side of every
    # convolution in the rocprofv3 trace), without it (hipEventDisableSystemFence) still 2 % of a frame when placed on every launch
    # of every step (A/B: scripts/ab_profile_events.sh) -- and the product path (bin/fav_stylize) records none
    pe = 0 if args.no_profile else max(1, args.profile_every)
    n_prof_steps = 0
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        on = pe > 0 and i % pe == 0
        net.profile_enable(on)
        n_prof_steps += 1 if on else 0
        step(i)
    t_enq = time.perf_counter() - t0                    # host time to ENQUEUE the steps (the GPU runs behind)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    net.check()                                         # no stream-K hand-off timed out inside the timed region
    net.profile_enable(False)
    prof = net.profile_read()
    dt = shard.max_over_ranks(dt, cdev)

    # not part of `value`: the same loop with the checker's 4-argument (image-structure) mode, which is what
    # makeOptFlow_deepflow.sh:59 runs in production; its masks are computed two frames ahead on the side queues
    extra = {}
    if world == 1 and not args.structure and not args.no_extra:
        def step4(i):
            k2 = (i + 2) % ring
            stream.prefetch_mask(frames[k2], bws[k2], fws[k2], use_structure=True)
            stream.next_frame_flow(frames[i % ring], bws[i % ring], fws[i % ring], use_structure=True, want_f32=False, out_u8=out8)
        stream.prefetch_mask(frames[0], bws[0], fws[0], use_structure=True)
        stream.prefetch_mask(frames[1], bws[1], fws[1], use_structure=True)
        for i in range(4):
            step4(i)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        n4 = max(8, args.steps // 2)
        for i in range(4, 4 + n4):
            step4(i)
        torch.cuda.synchronize()
        extra["frames_per_s_4arg_structure_mode_lookahead"] = round(n4 / (time.perf_counter() - t1), 3)
        # informational, NOT the parity mode and not `value`: the optional fast mode with bf16 operands in the halo-resident 3x3
        # convolutions (fav_net_set_precision; tests gate it at >= 45 dB PSNR against the fp32 oracle)
        net.profile_enable(False)
        net.set_precision(True)
        for i in range(4):
            stream.next_frame_flow(frames[i % ring], bws[i % ring], fws[i % ring], use_structure=False, want_f32=False, out_u8=out8)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for i in range(4, 4 + n4):
            stream.next_frame_flow(frames[i % ring], bws[i % ring], fws[i % ring], use_structure=False, want_f32=False, out_u8=out8)
        torch.cuda.synchronize()
        extra["frames_per_s_bf16_operand_fast_mode_3arg"] = round(n4 / (time.perf_counter() - t1), 3)
        net.set_precision(False)

    if rank == 0:
        fps = world * args.steps / dt
        # roofline of the dominant kernel: the ten 3x3 128->128 residual convolutions (91.7 of the 152.8 GMAC per frame).  Round 2:
        # Winograd F(2x2,3x3) on the fp32 matrix cores (kernels_wino.hip, kernel id 528).  `achieved` = ALGORITHMIC FLOPs (the direct
        # convolution's 2 * MACs) / HIP-event time of those launches; the kernel EXECUTES 16/36 of them, so
        # `achieved` may exceed the fp32 MFMA peak -- `executed_tflops` / `executed_frac` give the matrix-pipe view.
        dom = [(ms, n, macs) for (ms, n, macs, kid) in prof if kid == 528 and n > 0]
        dom_name = "conv3_wino_kernel (ten 3x3 128->128 residual convolutions, Winograd F(2x2,3x3), fp32 MFMA 32x32x2)"
        exec_ratio = 16.0 / 36.0
        if not dom:     # FAV_NO_WINO: the halo-resident direct form
            dom = [(ms, n, macs) for (ms, n, macs, kid) in prof if kid == 428 and n > 0]
            dom_name = "conv3_halo_kernel<128, false> (ten 3x3 128->128 residual convolutions, stream-K, halo-resident operand)"
            exec_ratio = 1.0
        if not dom:     # FAV_NO_H3: fall back to the generic 128-wide instance
            dom = [(ms, n, macs) for (ms, n, macs, kid) in prof if kid == 128 and n > 0]
            dom_name = "conv_mfma_kernel<128,2,2,0,true> (3x3 128->128 residual convolutions + 64->128 stride-2)"
            exec_ratio = 1.0
        flops = sum(2.0 * macs * n for ms, n, macs in dom); secs = sum(ms for ms, n, macs in dom) / 1e3
        nl = sum(n for ms, n, macs in dom)
        achieved = flops / secs / 1e12 if secs > 0 else 0.0
        conv_ms = sum(ms for ms, n, macs, kid in prof) / max(1, n_prof_steps)
        # HBM bytes per launch from the PMC passes (scripts/gpu_pmc.sh -> profiles/pmc_traffic.json).  The file records the hash of
        # the kernel source it was measured on: a stale file (kernel changed since) is refused rather than reported.
        traffic, traffic_note = None, "profiles/pmc_traffic.json missing"
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            ksrc = os.path.join(ROOT, "fast-artistic-videos_amd", "csrc", tj.get("source", "kernels_conv.hip"))
            khash = hashlib.sha256(open(ksrc, "rb").read()).hexdigest()[:16] if os.path.exists(ksrc) else "?"
            if not tj.get("kernel", "").startswith(dom_name.split(" ")[0].split("<")[0]):
                traffic_note = "pmc_traffic.json is for another kernel (%s)" % tj.get("kernel")
            elif tj.get("source_sha16") != khash:
                traffic_note = "stale: measured on %s %s, current %s -- re-run scripts/gpu_pmc.sh" % (tj.get("source"), tj.get("source_sha16"), khash)
            else:
                traffic, traffic_note = tj["hbm_bytes_per_launch"], ("rocprofv3 --pmc FETCH_SIZE (x2, gfx950 calibration) + WRITE_SIZE, separate passes, %s %s"
                                                                     % (tj.get("source"), khash))
        per_kernel = {}
        for ms, n, macs, kid in prof:
            if n:
                k = per_kernel.setdefault(str(kid), [0.0, 0.0]); k[0] += ms / max(1, n_prof_steps); k[1] += 2.0 * macs * n / max(1, n_prof_steps)
        line = {
            "metric": "stylized frames/sec @1280x720, per-frame hot path (mask + warp + assembly + net + deprocess) with inputs resident in HBM; "
                      "file->PNG rate in `e2e`, PSNR vs CPU ref in `parity`",
            "value": round(fps, 3), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "1280x720 fused on-GPU consistency check (%s) + min-filter + warp + assemble + transformer net "
                                   "(c9s1-32,d64,d128,R128x5,U2,c3s1-64,U2,c9s1-3, reflect-start pad 40) + deprocess, inputs in HBM, "
                                   "1 independent stream per GPU" % ("4-arg" if args.structure else "3-arg"),
                       "frame": [W, H], "streams": world, "parallelism": f"{world} independent streams, no data-path collective"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_note": traffic_note,
                         "kernel": dom_name,
                         "executed_tflops": round(achieved * exec_ratio, 3), "executed_frac": round(achieved * exec_ratio / FP32_MFMA_PEAK_TFLOPS, 4),
                         "note": "achieved = algorithmic (direct-convolution) FLOPs / time; the Winograd kernel executes 16/36 of them on the matrix pipe, "
                                 "so frac > 1 means faster than any direct fp32 convolution could run" if exec_ratio < 1 else "direct form: executed = algorithmic",
                         "avg_launch_us": round(secs / max(1, nl) * 1e6, 2), "launches": nl,
                         "timed_with": "HIP events (no system fence) around every convolution launch of every %d-th step of the timed region (%d of %d steps)" % (max(1, pe), n_prof_steps, args.steps),
                         "conv_stack_ms_per_frame": round(conv_ms, 4),
                         "conv_stack_tflops": round(FLOP_PER_FRAME / (conv_ms * 1e-3) / 1e12, 3) if conv_ms > 0 else None,
                         "conv_stack_frac": round(FLOP_PER_FRAME / (conv_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4) if conv_ms > 0 else None,
                         "per_kernel_ms_tflops": {k: [round(v[0], 4), round(v[1] / (v[0] * 1e-3) / 1e12, 1)] for k, v in per_kernel.items()}},
        }
        line["extra"] = extra
        # the launch path is not the limiter: host enqueue time per frame vs GPU time per frame (why a HIP graph would not help:
        # kernel boundaries cost the same GPU-side in a replayed graph, MI355X_MICROARCH.md "boundary" row)
        line["extra"]["host_enqueue_ms_per_frame"] = round(t_enq / args.steps * 1e3, 4)
        if use_dist:
            line["extra"]["weight_broadcast"] = {"backend": "nccl (RCCL)", "world": world, "bytes": len(blob), "seconds_incl_first_call_setup": round(t_b, 4)}
        if world == 1 and not args.no_cpu_baseline:
            # one more recurrent step on a fresh stream, float outputs kept: GPU result and oracle result of the SAME inputs
            layers = t7.extract_layers(t7.load(ckpt)["model"])
            pst = fav_amd.Stream(net, H, W)
            o0, _ = pst.first_frame(frames[0])
            o1, u1 = pst.next_frame_flow(frames[1], bws[1], fws[1], use_structure=False, want_u8=True)
            torch.cuda.synchronize(); net.check()
            base, parity = cpu_baseline_and_parity(layers, frames_h[1], bw_h[1], fw_h[1], o0.cpu().numpy(), o1.cpu().numpy(), u1.cpu().numpy(),
                                                   pst.last_mask().cpu().numpy())
            ref = reference_checker_baseline(bw_h[1], fw_h[1], frames_h[1])
            if ref is not None:
                base["reference_consistency_checker"] = ref
            line["cpu_baseline"] = base
            line["parity"] = parity
            del pst
        if world == 1 and not args.no_e2e:
            del stream; torch.cuda.synchronize()
            line["e2e"] = e2e_block(ckpt, frames_h, bw_h, fw_h)
        print(json.dumps(line), flush=True)
        try:
            os.remove(ckpt)
        except OSError:
            pass
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
