#!/usr/bin/env python3
"""bench.py -- stylised frames/sec of the per-frame hot path at 1280x720 on N MI355X (one node).

One "step" = one pass of the hot path over one frame of one video stream:
    on-GPU consistency check (forward+backward flow) -> certainty erosion -> warp of the previous
    stylised frame -> 7-channel assembly + reflection pad -> transformer network -> de-process
    -> image.save: the bytes of the PNG file, produced on the device (A9; round 3)
with all inputs (uint8 frame, backward .flo payload, forward .flo payload, previous output) already
resident in HBM and the PNG file's bytes left in HBM (BASELINE.json configs[2]: "1280x720 x 300 frames, on-GPU warp + consistencyChecker +
net fused").  Each rank owns one independent video stream (the path shards across streams only: frame i
needs frame i-1's output), so N GPUs = N streams, weak scaling, no data-path collective; the only
collective is the RCCL broadcast of the packed weight blob from rank 0 before the timed region.

Launch:  python bench.py [--gpus 1] [--steps K] [--warmup W]
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                --master-port P bench.py --gpus N --steps K --warmup W
Rank 0 prints ONE JSON line.

`value` follows the bench contract (K timed steps after W warm-up steps, inputs resident in HBM, barrier + synchronize on both
sides, max over ranks).  The FILE -> PNG rate of the product CLI -- BASELINE.json's "end-to-end" -- is in the same line as
`end_to_end_fps` (top level) and the `e2e` block (bin/fav_stylize over RAM-backed files: decode, H2D, the same GPU work, D2H,
write; at --gpus N through the product's own launcher `fav_stylize -streams ... -gpus N`), with a sustained leg (>= 3000 frames,
shader clock sampled) and the host CPU milliseconds per frame.
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


def cpu_baseline_and_parity(layers, frame1, bw, fw, prev_state, gpu_out, gpu_out_u8, gpu_mask, structure=1):
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
        mask = O.consistency(b, f, np.ascontiguousarray(frame1) if structure else None)      # (the timed configuration's checker mode)
        r1 = st.next(f1, b, mask.astype(np.float32) / np.float32(255))
        times.append(time.perf_counter() - t0)
    dt = min(times)
    base = {"value": round(1.0 / dt, 5), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"1 frame of 1280x720 ({'4' if structure else '3'}-arg mask + min-filter + warp + assemble + net + deprocess) through oracle/ "
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
        out = {"kind": "reference", "cores": 1, "unit": "masks/s", "value": round(1.0 / res["3arg"], 3), "value_4arg": round(1.0 / res["4arg"], 3),
               "sample": "consistencyChecker (reference sources, g++ -O3) on one 1280x720 flow pair incl. .flo read + .pgm write on /dev/shm, "
                         "min of 3: %.3f s (3-arg) / %.3f s (4-arg, image structure)" % (res["3arg"], res["4arg"])}
        # the process-level drop-in timed the way the reference's driver calls it (makeOptFlow_deepflow.sh:59-60: one process per mask),
        # and its list mode (one GPU context for N pairs); outputs compared byte for byte with the reference binary's
        mine = os.path.join(ROOT, "fast-artistic-videos_amd", "bin", "consistencyChecker")
        if os.path.exists(mine):
            o2 = os.path.join(d, "out2.pgm")
            drop, want = {}, {}
            # round 5: the first call leaves a resident helper behind (host/consistency_checker.cpp) and later calls hand it their argv:
            # `*_s_per_call` is what a sequence of calls costs (min of 3 after the call that starts the helper); `*_fresh_process_s` is a
            # call that computes in its own process (FAV_CC_DAEMON=0: rounds 1-4's form)
            run = os.path.join(d, "run"); os.mkdir(run, 0o700)
            env_h = dict(os.environ, XDG_RUNTIME_DIR=run, FAV_CC_IDLE_S="60"); env_p = dict(os.environ, FAV_CC_DAEMON="0")
            for name, extra in (("3arg", []), ("4arg", [i])):
                subprocess.check_call([exe, a, b, o] + extra, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                want[name] = open(o, "rb").read()
                ts = []
                for _ in range(4):
                    t0 = time.perf_counter(); subprocess.check_call([mine, a, b, o2] + extra, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env_h); ts.append(time.perf_counter() - t0)
                drop[name + "_s_per_call"] = round(min(ts[1:]), 4); drop[name + "_first_call_s"] = round(ts[0], 4)
                drop[name + "_bytes_equal_reference"] = open(o2, "rb").read() == want[name]
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter(); subprocess.check_call([mine, a, b, o2] + extra, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env_p); ts.append(time.perf_counter() - t0)
                drop[name + "_fresh_process_s"] = round(min(ts), 4)
                drop[name + "_fresh_process_bytes_equal_reference"] = open(o2, "rb").read() == want[name]
            try:      # end the helper (its pid is in the lock file; it would leave by itself after FAV_CC_IDLE_S)
                import signal
                import glob
                for lk in glob.glob(os.path.join(run, "fav-cc", "gpu0*.lock")):
                    os.kill(int(open(lk).read().split()[0]), signal.SIGTERM)
            except Exception:
                pass
            drop["reference_s_per_call"] = {"3arg": round(res["3arg"], 4), "4arg": round(res["4arg"], 4)}
            npairs = 40
            for name, extra in (("3arg", []), ("4arg", [i])):
                lst = os.path.join(d, "pairs_%s.txt" % name)
                with open(lst, "w") as f:
                    for k in range(npairs):
                        f.write(" ".join([a, b, os.path.join(d, "bo_%d.pgm" % k)] + extra) + "\n")
                t0 = time.perf_counter(); subprocess.check_call([mine, "-batch", lst], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL); dt = time.perf_counter() - t0
                drop[name + "_batch_s_per_pair"] = round(dt / npairs, 5)
                drop[name + "_batch_bytes_equal_reference"] = all(open(os.path.join(d, "bo_%d.pgm" % k), "rb").read() == want[name] for k in (0, npairs // 2, npairs - 1))
            drop["note"] = ("bin/consistencyChecker (this repo, mask on the GPU) as a process: same argv, wall time per call incl. file reads and the .pgm write on /dev/shm; "
                            "`*_s_per_call` = through the resident helper the first call leaves behind (min of 3 calls after it), `*_first_call_s` = the call that starts the helper "
                            "(one HIP context), `*_fresh_process_s` = FAV_CC_DAEMON=0 (a HIP context per call, rounds 1-4); `-batch list.txt` = %d pairs in one process.  "
                            "Reference binary beside it: %.3f / %.3f s per call" % (npairs, res["3arg"], res["4arg"]))
            out["gpu_drop_in_process"] = drop
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


class ClockSampler:
    """shader clock while a leg runs: rocm-smi polled from a thread (0.5 s period); best effort, None when unavailable"""

    def __init__(self):
        import threading
        self.vals, self.stop = [], threading.Event()
        self.t = threading.Thread(target=self._run, daemon=True)

    def _read(self):
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--json"], capture_output=True, text=True, timeout=5)
            j = json.loads(r.stdout)
            for card in j.values():
                for k, v in card.items():
                    if "sclk" in k.lower() and "mhz" in str(v).lower():
                        return float(str(v).lower().replace("(", "").replace(")", "").replace("mhz", "").strip())
        except Exception:
            return None
        return None

    def _run(self):
        while not self.stop.is_set():
            v = self._read()
            if v:
                self.vals.append(v)
            self.stop.wait(0.5)

    def __enter__(self):
        self.t.start(); return self

    def __exit__(self, *a):
        self.stop.set(); self.t.join(timeout=10)

    def summary(self):
        if not self.vals:
            return None
        return {"samples": len(self.vals), "min_mhz": min(self.vals), "mean_mhz": round(sum(self.vals) / len(self.vals), 1), "max_mhz": max(self.vals),
                "source": "rocm-smi --showclocks, 0.5 s period, while the leg ran"}


def _cli_leg(base, extra_args, out_dirs, timeout=900, taskset=None, verify=None, prefix="out", exe_override=None):
    """one run of the CLI; `verify` = (reference run, frames, pixel_frames or None): every PNG it wrote is compared byte for byte with
    the in-process run of the same inputs (scripts/e2e_content.py) before the directory is removed -- `png_mismatch_frames` must be 0"""
    import e2e_content as EC
    cmd = (["taskset", "-c", taskset] if taskset else []) + (exe_override or []) + base + extra_args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        for od in out_dirs:
            shutil.rmtree(od, ignore_errors=True)
        return {"error": (r.stderr or r.stdout)[-400:]}
    j = json.loads(lines[-1])
    res = {"fps": j["fps_end_to_end"], "seconds": j["seconds"], "frames": j["frames"]}
    for k in ("png_writers", "wait_loader_s", "wait_png_pool_s", "setup_s", "png_tail_s", "png_encoder", "host_cpu_ms_per_frame", "cpu_ms_per_frame_loaders", "cpu_ms_per_frame_writers", "cpu_ms_per_frame_writers_waiting_for_the_copy", "cpu_ms_per_frame_main", "png_mb_per_frame", "usable_cpus",
              "gpus", "streams", "fps_per_gpu", "weight_broadcast_ms", "rccl_comm_init_ms"):
        if k in j:
            res[k] = j[k]
    if "setup_s" in j:
        res["steady_state_fps"] = round(j["frames"] / max(1e-9, j["seconds"] - (j.get("setup_s") or 0.0) - (j.get("png_tail_s") or 0.0)), 3)
    n_png = 0
    chk = {"png_checked": 0, "png_mismatch_frames": 0, "png_missing_frames": 0, "png_unexpected_files": 0}
    for od in out_dirs:
        if os.path.isdir(od):
            n_png += len([f for f in os.listdir(od) if f.endswith(".png")])
            if verify is not None:
                ref, nfr, pixel_frames = verify
                one = EC.verify_dir_pixels(od, prefix, ref, pixel_frames) if pixel_frames is not None else EC.verify_dir(od, prefix, ref, nfr)
                for k, v in one.items():
                    if isinstance(v, int) and k != "first_bad_frame":
                        chk[k] = chk.get(k, 0) + v
                if one.get("first_bad_frame"):
                    chk.setdefault("first_bad_frame", one["first_bad_frame"])
            shutil.rmtree(od, ignore_errors=True)
        elif verify is not None:
            chk["png_missing_frames"] += verify[1]
    res["png_written"] = n_png
    if verify is not None:
        res.update(chk)
    return res


def e2e_block(ckpt, net, frames_h, bw_h, fw_h, nframes=300, world=1, sustained_frames=3000, quick=False, structure=1):
    """File -> PNG rate of the drop-in CLI (fast_artistic_video.lua:93-97,160-170) -- BASELINE.json's "end-to-end": bin/fav_stylize
    over RAM-backed 1280x720 P6 frames + backward/forward .flo, on-GPU consistency check in the headline mode (`structure`: 1 = the
    checker's 4-argument form, the CLI's default and what makeOptFlow_deepflow.sh:59-60 runs; 0 = 3-argument), PNG files written back to RAM.
    Everything the in-HBM `value` leaves out is inside: file reads, decode, H2D, D2H, file writes.  world > 1: the product's own
    launcher (`-streams s0,.. -gpus N`: one worker process per GPU, RCCL broadcast of the weights), one clip per GPU.
    EVERY leg's output is content-checked: each PNG byte for byte against the in-process fav_stream_* run of the same inputs
    (`png_mismatch_frames`, `png_missing_frames`; the -png_encoder host legs, whose deflate streams differ by construction, by decoded
    pixels on every 10th frame)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import oracle as O
    import e2e_content as EC
    import fav_amd
    pkg = os.path.join(ROOT, "fast-artistic-videos_amd")
    exe = os.path.join(pkg, "bin", "fav_stylize")
    if not os.path.exists(exe) or not os.path.isdir("/dev/shm"):
        return {"error": "bin/fav_stylize or /dev/shm missing"}
    d = tempfile.mkdtemp(prefix="fav_e2e_", dir="/dev/shm")
    try:
        have_ref_checker = os.path.exists(EC.REF_CHECKER)
        names = [f"s{k}" for k in range(world)]
        for nm in names:
            EC.make_clip_dir(d, nm, frames_h, bw_h, fw_h, nframes, O, cert=have_ref_checker and world == 1 and not quick)
        pat = "%S" if world > 1 else names[0]
        flow_args = lambda p: ["-input_pattern", f"{d}/{p}/frame_%05d.ppm", "-flow_pattern", f"{d}/{p}/flow/backward_[%d]_{{%d}}.flo"]
        tail = ["-model_vid", ckpt, "-model_img", "self", "-gpu", "0", "-timing", "1"]
        base = [exe] + flow_args(pat) + ["-forward_flow_pattern", f"{d}/{pat}/flow/forward_{{%d}}_[%d].flo"] + tail
        if world > 1:
            base += ["-streams", ",".join(names), "-gpus", str(world)]
        outp = lambda tag: ["-output_prefix", f"{d}/{pat}/o_{tag}/out"]
        outd = lambda tag: [f"{d}/{nm}/o_{tag}" for nm in names]
        out = {"frames_per_stream": nframes, "streams": world, "host_threads": os.cpu_count(), "usable_cpus": effective_cpus(),
               "pipeline": "P6 + 2 x .flo from /dev/shm -> H2D -> %s check + warp + net + PNG encode (GPU) -> D2H (exact size) -> write() to /dev/shm" % ("4-arg (look-ahead)" if structure else "fused 3-arg"),
               "checker_mode_of_the_headline_legs": "4-argument (-structure 1)" if structure else "3-argument (-structure 0)",
               "h2d_bytes_per_frame": H * W * (3 + 8 + 8),
               "content_check": "every PNG of every leg byte for byte against the in-process fav_stream_* run of the same inputs (bit-deterministic GPU path); "
                                "png_mismatch_frames / png_missing_frames must be 0"}
        # the in-process runs the legs are checked against (outside every timed region; this process is idle while a CLI leg runs)
        n_long = sustained_frames if (sustained_frames and world == 1 and not quick) else nframes
        every10 = list(range(10, nframes + 1, 10)) + ([nframes] if nframes % 10 else [])
        t_ref = time.perf_counter()
        hm, om = ("4arg", "3arg") if structure else ("3arg", "4arg")      # headline / other checker mode
        HS, OS = ["-structure", "1" if structure else "0"], ["-structure", "0" if structure else "1"]
        ref3 = EC.reference_run(fav_amd, net, frames_h, bw_h, fw_h, max(n_long, nframes), hm, keep=every10)      # (the HEADLINE mode's reference, whatever its name says)
        out["reference_run_s"] = {"%s_%d_frames" % (hm, max(n_long, nframes)): round(time.perf_counter() - t_ref, 2)}
        # the headline leg: the defaults of the CLI (-png_encoder gpu), 3-argument check = the workload of `value`
        # (at --gpus N a launcher that cannot bring its RCCL communicator up must not hold the bench line back: 4 minutes)
        out["gpu_png"] = _cli_leg(base, HS + outp("gpu"), outd("gpu"), timeout=900 if world == 1 else 240, verify=(ref3, nframes, None))
        if quick:
            return out
        if world == 1:
            # the SAME job through the product's multi-GPU launcher (-streams s0 -gpus 1 -force_dist 1: worker process, RCCL communicator
            # of one rank, ncclBroadcast of the packed weights): the exact command line the N > 1 runs issue, driver-run at N = 1
            lb = [exe] + flow_args("%S") + ["-forward_flow_pattern", f"{d}/%S/flow/forward_{{%d}}_[%d].flo"] + tail + ["-streams", names[0], "-gpus", "1", "-force_dist", "1"]
            out["gpu_png_via_launcher_rccl_world_1"] = _cli_leg(lb, HS + ["-output_prefix", f"{d}/%S/o_ln/out"], outd("ln"), timeout=300, verify=(ref3, nframes, None))
            # what the same job costs when the host deflates (round 2's path) and with the 4-argument (image-structure) check of
            # makeOptFlow_deepflow.sh:59; then the per-GPU share of a 16-CPU quota on an 8-GPU node: two cores
            out["host_zlib_png_level_1"] = _cli_leg(base, HS + ["-png_encoder", "host", "-png_level", "1"] + outp("zl"), outd("zl"), verify=(ref3, nframes, every10))
            t_ref = time.perf_counter()
            ref4 = EC.reference_run(fav_amd, net, frames_h, bw_h, fw_h, nframes, om)
            out["reference_run_s"]["%s_%d_frames" % (om, nframes)] = round(time.perf_counter() - t_ref, 2)
            out["gpu_png_%s_check" % om] = _cli_leg(base, OS + outp("g4"), outd("g4"), verify=(ref4, nframes, None))
            del ref4
            if have_ref_checker:
                # the certainty path exactly as stylizeVideo_deepflow.sh:87-96 calls it -- through the `th` shim, .pgm files written by the
                # REFERENCE's checker binary (4-argument form, makeOptFlow_deepflow.sh:59-60) -- BASELINE config 2's data flow at config 3's size
                masks = [O.read_pnm(f"{d}/src/r{k}.pgm") for k in range(len(frames_h))]
                refc = EC.reference_run(fav_amd, net, frames_h, bw_h, fw_h, nframes, "cert", masks_h=masks)
                th = [os.path.join(pkg, "host", "th"), "fast_artistic_video.lua"] + flow_args(names[0]) + ["-occlusions_pattern", f"{d}/{names[0]}/flow/reliable_[%d]_{{%d}}.pgm",
                      "-backend", "cuda", "-use_cudnn", "1"] + tail
                out["gpu_png_cert_path_via_th_shim"] = _cli_leg(th, outp("ct"), outd("ct"), verify=(refc, nframes, None))
                del refc
            out["host_zlib_two_cores"] = _cli_leg(base, HS + ["-png_encoder", "host", "-png_level", "1", "-num_frames", "100"] + outp("z2"), outd("z2"), taskset="0-1",
                                                  verify=(ref3, 100, list(range(10, 101, 10))))
        # sustained leg: >= 3000 frames (same ring of inputs), shader clock sampled while it runs
        if sustained_frames and world == 1:
            EC.make_clip_dir(d, "long", frames_h, bw_h, fw_h, sustained_frames, O)
            lb = [a.replace(f"{d}/{names[0]}/", f"{d}/long/") for a in base]
            with ClockSampler() as cs:
                leg = _cli_leg(lb, HS + ["-output_prefix", f"{d}/long/o/out"], [f"{d}/long/o"], timeout=1200, verify=(ref3, sustained_frames, None))
            leg["shader_clock"] = cs.summary()
            out["sustained"] = leg
            # the per-GPU share of a 16-CPU quota on an 8-GPU node: the whole process (loaders, submission, file writes, the HIP
            # runtime's own threads) confined to two CPUs
            leg2 = _cli_leg(lb, HS + ["-output_prefix", f"{d}/long/o2/out"], [f"{d}/long/o2"], timeout=1200, taskset="0-1", verify=(ref3, sustained_frames, None))
            leg2["note"] = "taskset -c 0-1 on the %d-frame clip" % sustained_frames
            out["sustained_two_cores"] = leg2
        legs = [v for v in out.values() if isinstance(v, dict) and "png_written" in v]
        out["png_mismatch_frames_all_legs"] = sum(v.get("png_mismatch_frames", 0) + v.get("png_missing_frames", 0) for v in legs)
        out["png_checked_all_legs"] = sum(v.get("png_checked", 0) + v.get("png_checked_by_pixels", 0) for v in legs)
        return out
    except Exception as e:          # the e2e leg must never take the bench line with it
        return {"error": repr(e)[:400]}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def hbm_kernel_block(dev):
    """HBM-bound gather kernels of the path (north_star: "evidenced by rocprof HBM GB/s against peak"): algorithmic bytes per launch
     / live duration (torch events on the stream the operator-level entry points launch on) against 8 TB/s.  The fused
    prep_input_kernel has no operator-level entry: its line comes from the committed rocprofv3 summary of this same command."""
    import torch
    import fav_amd
    px = H * W
    res = {}
    g = torch.Generator(device="cpu"); g.manual_seed(5)
    bw = (torch.randn((H, W, 2), generator=g) * 2).to(dev); fw = (torch.randn((H, W, 2), generator=g) * 2).to(dev)
    img = torch.rand((3, H, W), generator=g).to(dev); flow = (torch.randn((2, H, W), generator=g) * 2).to(dev)
    u8 = (torch.rand((H, W, 3), generator=g) * 255).to(torch.uint8).to(dev)

    def timed(fn, n=30):
        for _ in range(3): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / n          # us, back-to-back launches (includes the kernel boundary)

    out_mask = torch.empty((H, W), dtype=torch.uint8, device=dev)
    ws_b = fav_amd.lib().fav_consistency_workspace_bytes(W, H, 0)
    import ctypes as C
    P = lambda t: C.c_void_p(t.data_ptr())
    S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    cases = {
        "consistency_kernel (A3, 3-arg)": (17 * px, lambda: fav_amd.lib().fav_consistency_u8(P(bw), P(fw), None, P(out_mask), W, H, None, C.c_size_t(0), S())),
    }
    outw = torch.empty((1, 3, H, W), dtype=torch.float32, device=dev)
    cases["warp_kernel (A2, operator form)"] = (32 * px, lambda: fav_amd.lib().fav_warp_bdhw_f32(P(img), P(flow), P(outw), 1, 3, H, W, H, W, 0, S()))
    cap = fav_amd.lib().fav_png_capacity(W, H); wsb = fav_amd.lib().fav_png_workspace_bytes(W, H)
    pout = torch.empty(cap + 4, dtype=torch.uint8, device=dev); pn = torch.zeros(1, dtype=torch.int32, device=dev); pws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    cases["png_rows_kernel + png_pack_kernel (A9, u8 source)"] = (None, lambda: fav_amd.lib().fav_png_encode_rgb8(P(u8), W, H, P(pout), C.c_size_t(cap), P(pn), P(pws), C.c_size_t(wsb), S()))
    for name, (nbytes, fn) in cases.items():
        us = timed(fn)
        if nbytes is None:      # PNG: u8 in + staged out + staged in + packed out
            sz = int(pn.item()); nbytes = 3 * px + 3 * sz
        res[name] = {"algorithmic_bytes": int(nbytes), "us_per_launch": round(us, 2), "gb_per_s": round(nbytes / us / 1e3, 1), "frac_of_8tb_s": round(nbytes / us / 1e3 / 8000.0, 4)}
    # the fused kernels of the per-frame path have no operator-level entry: their lines come from the committed rocprofv3 summaries of
    # this same command (kernel-trace durations + FETCH_SIZE / WRITE_SIZE of separate --pmc passes)
    try:
        import csv
        # which committed summaries: profiles/CURRENT_PROFILE.json names the round's rocprofv3 kernel statistics and PMC summary of this
        # command (written by hand next to the files it names when a round's final profile is committed)
        cur = json.load(open(os.path.join(ROOT, "profiles", "CURRENT_PROFILE.json")))
        ks = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(os.path.join(ROOT, "profiles", cur["kernel_stats"])))}
        pm = json.load(open(os.path.join(ROOT, "profiles", cur["pmc_summary"])))
        # (round 5: check + erosion + input assembly are ONE kernel in the timed configuration -- check_prep_kernel: frame 3 B, two flows 16 B, the
        #  previous output's gathers 12 B in; mask 1 B, certainty 4 B and the padded NHWC8 input out; profiles older than that hold the two kernels)
        for name, alg in (("check_prep_kernel", (3 + 16 + 12 + 5) * px + (H + 80) * (W + 80) * 32),
                          ("prep_input_kernel", (3 + 8 + 4 + 12) * px + (H + 80) * (W + 80) * 32), ("min_filter_kernel<2>", 16 * px + 5 * px)):
            if not any(name in k for k in ks) or not any(name in k for k in pm):
                continue
            us = [v for k, v in ks.items() if name in k][0]
            c = [v for k, v in pm.items() if name in k][0]
            res[name + " (from profiles/%s, %s)" % (cur["kernel_stats"], cur["pmc_summary"])] = {
                "algorithmic_bytes": int(alg), "counter_bytes": int((c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024),
                "us_per_launch": round(us, 2), "gb_per_s": round(alg / us / 1e3, 1), "frac_of_8tb_s": round(alg / us / 1e3 / 8000.0, 4)}
    except Exception as e:
        res["fused_kernels_from_profiles"] = {"error": repr(e)[:200]}
    res["note"] = ("back-to-back launches timed with events on the launching stream (kernel boundaries included); these kernels move 16-60 MB, i.e. 2-8 us at "
                   "8 TB/s: they run in the launch-latency regime.  prep_input_kernel (A2+A6+A7+pad fused, ~60 MB algorithmic): profiles/*_kernel_stats.csv")
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--structure", type=int, default=1, help="1 (default since round 6) = the checker's 4-argument, image-structure form: what makeOptFlow_deepflow.sh:59-60 "
                    "runs in production and what bin/fav_stylize defaults to; 0 = the 3-argument form (the headline of rounds 1-5, now `config.value_in_3arg_mode`)")
    ap.add_argument("--no-profile", action="store_true", help="no per-convolution HIP events in the timed region (A/B of their cost; the roofline block is then empty)")
    ap.add_argument("--profile-every", type=int, default=4, help="HIP events around the convolutions on every n-th step of the timed region")
    ap.add_argument("--lookahead", type=int, default=-1, help="1 = compute the masks of the next two frames on the side queues "
                    "(fav_stream_prefetch_mask: mask + certainty erosion, off the critical path); default: on for --structure 1 only")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU legs (oracle timing + parity block, reference checker)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the file->PNG run of bin/fav_stylize after the timed region")
    ap.add_argument("--force-dist", action="store_true", help="initialise the RCCL process group and run the weight broadcast even at "
                    "world size 1 (exercises the N>1 launch path on a 1-GPU box)")
    ap.add_argument("--sustained-frames", type=int, default=3000, help="length of the sustained file->PNG leg (0 = skip)")
    ap.add_argument("--quick-e2e", action="store_true", help="only the headline file->PNG leg")
    ap.add_argument("--png-async", action="store_true", help="PNG encode on the stream's encoder queue next to the following frame (fav_stream_encode_png_async) "
                    "instead of on the compute queue in front of it (fav_stream_encode_png, the default: what bin/fav_stylize does)")
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
    if use_dist and rank == 0:
        try:
            os.remove(os.path.join(tempfile.gettempdir(), "fav_bench_done_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"))))
        except OSError:
            pass
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
    png_out, png_n = stream.png_buffers()          # A9: the frame leaves the step as the bytes of its PNG file (in HBM)

    def step(i):
        k = i % ring
        if args.lookahead:      # the masks of frames i+1, i+2 depend only on inputs that are already resident: queue them
            k2 = (i + 2) % ring  # on the side queues BEFORE frame i's network so they overlap it (two in flight)
            stream.prefetch_mask(frames[k2], bws[k2], fws[k2], use_structure=bool(args.structure))
        stream.next_frame_flow(frames[k], bws[k], fws[k], use_structure=bool(args.structure), want_f32=False, want_u8=False)
        if not args.png_async:
            stream.encode_png_into(png_out, png_n)
        else:   # the encoder's kernels on the stream's own queue, next to frame i + 1's network (every frame is still encoded in full)
            stream.encode_png_async_into(png_out, png_n)

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
    if world == 1 and not args.no_extra:
        # continuity with rounds 1-2, whose step ended at the de-processed frame (A9 was a host job then): the same loop without the PNG encode
        def frame_only(i):
            if args.lookahead:
                k2 = (i + 2) % ring
                stream.prefetch_mask(frames[k2], bws[k2], fws[k2], use_structure=bool(args.structure))
            stream.next_frame_flow(frames[i % ring], bws[i % ring], fws[i % ring], use_structure=bool(args.structure), want_f32=False, want_u8=False)
        for i in range(args.steps, args.steps + 4):
            frame_only(i)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        n0 = max(8, args.steps // 2)
        for i in range(args.steps + 4, args.steps + 4 + n0):
            frame_only(i)
        torch.cuda.synchronize()
        extra["frames_per_s_without_png_encode"] = round(n0 / (time.perf_counter() - t1), 3)
        # ... and with the encode on the other queue (informational: the form `value` does not use)
        def step_other(i):
            frame_only(i)
            if args.png_async: stream.encode_png_into(png_out, png_n)
            else: stream.encode_png_async_into(png_out, png_n)
        b0 = args.steps + 4 + n0
        for i in range(b0, b0 + 4):
            step_other(i)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for i in range(b0 + 4, b0 + 4 + n0):
            step_other(i)
        torch.cuda.synchronize()
        extra["frames_per_s_png_encode_on_the_compute_queue" if args.png_async else "frames_per_s_png_encode_on_the_streams_encoder_queue"] = round(n0 / (time.perf_counter() - t1), 3)
    if world == 1 and not args.no_extra:
        # the same loop in the OTHER checker mode (like for like: the timed step's own ending).  4-argument: masks two frames ahead on the
        # side queue; 3-argument: the check is part of the frame's first kernel
        other = not bool(args.structure)
        # (a network that has served look-ahead masks keeps its persistent grids at 252 blocks: the 3-argument sibling of a 4-argument run
        #  gets a network and a stream of its own, so that it is the plain 3-argument configuration)
        net_o = net if other else fav_amd.Net(blob=blob, device=local)
        stream_o = stream if other else fav_amd.Stream(net_o, H, W)
        png_o, png_no = (png_out, png_n) if other else stream_o.png_buffers()
        if not other:
            stream_o.first_frame(frames[0], want_f32=False, out_u8=out8)
        def step_o(i):
            if other:
                k2 = (i + 2) % ring
                stream_o.prefetch_mask(frames[k2], bws[k2], fws[k2], use_structure=True)
            stream_o.next_frame_flow(frames[i % ring], bws[i % ring], fws[i % ring], use_structure=other, want_f32=False, want_u8=False)
            if not args.png_async: stream_o.encode_png_into(png_o, png_no)
            else: stream_o.encode_png_async_into(png_o, png_no)
        if other:
            stream_o.prefetch_mask(frames[0], bws[0], fws[0], use_structure=True)
            stream_o.prefetch_mask(frames[1], bws[1], fws[1], use_structure=True)
        for i in range(4):
            step_o(i)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        n4 = max(8, args.steps // 2)
        for i in range(4, 4 + n4):
            step_o(i)
        torch.cuda.synchronize()
        extra["frames_per_s_4arg_structure_mode_lookahead" if other else "frames_per_s_3arg_mode"] = round(n4 / (time.perf_counter() - t1), 3)
        if not other:
            net_o.check(); del stream_o, net_o
    if world == 1 and not args.no_extra:
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

    # not part of `value` either: the same step on checkpoints with MORE FILTERS (README.md:141: the published VR models; the reference
    # builds any architecture string) -- every filter count doubled / x1.5; scripts/wide_bench.py.  Their layers must stay on the
    # minimal-filtering kernels (`fallback_layers` empty) at a conv-stack rate close to the canonical network's
    if world == 1 and not args.no_extra:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import wide_bench
        try:
            wb = wide_bench.run(wide_bench.WIDE2, steps=max(6, args.steps // 4), warmup=2, h=H, w=W, frames=frames, bws=bws, fws=fws)
            extra["wide_arch"] = wb
            extra["wide_arch_frames_per_s"] = wb["frames_per_s"]
            wb15 = wide_bench.run(wide_bench.WIDE15, steps=max(6, args.steps // 4), warmup=2, h=H, w=W, frames=frames, bws=bws, fws=fws)
            extra["wide_arch_x1_5"] = {k: wb15[k] for k in ("arch", "frames_per_s", "ms_per_frame", "conv_stack_algorithmic_tflops", "kernel_ids", "fallback_layers")}
        except Exception as e:      # informational block: never takes the bench line down
            extra["wide_arch"] = {"error": repr(e)[:300]}

    if rank == 0:
        fps = world * args.steps / dt
        # roofline of the dominant kernel: the ten 3x3 128->128 residual convolutions (91.7 of the 152.8 GMAC per frame).  Round 2:
        # Winograd F(2x2,3x3) on the fp32 matrix cores (kernels_wino.hip, kernel id 528).  `achieved` = ALGORITHMIC FLOPs (the direct
        # convolution's 2 * MACs) / HIP-event time of those launches; the kernel EXECUTES 16/36 of them, so
        # `achieved` may exceed the fp32 MFMA peak -- `executed_tflops` / `executed_frac` give the matrix-pipe view.
        # (kernel id 529 = the same kernel forming a pending residual join while it stages its input: three of the ten launches since
        #  round 3 -- they do the work of the three res_add launches they replace and are slower for it)
        # (round 4: kernel ids 728 / 729 = the same ten layers as Winograd F(4x4,3x3), kernels_wino4.hip -- 36 of the direct form's 144
        #  multiply-adds per 4x4 outputs; ids 528 / 529 = F(2x2,3x3) behind FAV_WINO_F2)
        f4 = any(kid in (728, 729) and n > 0 for (ms, n, macs, kid) in prof)
        ids = (728, 729) if f4 else (528, 529)
        dom = [(ms, n, macs) for (ms, n, macs, kid) in prof if kid in ids and n > 0]
        dom_plain = [(ms, n, macs) for (ms, n, macs, kid) in prof if kid == ids[0] and n > 0]
        dom_join = [(ms, n, macs) for (ms, n, macs, kid) in prof if kid == ids[1] and n > 0]
        dom_name = ("conv3_wino4_kernel (ten 3x3 128->128 residual convolutions, Winograd F(4x4,3x3), fp32 MFMA 16x16x4)" if f4 else
                    "conv3_wino_kernel (ten 3x3 128->128 residual convolutions, Winograd F(2x2,3x3), fp32 MFMA 32x32x2)")
        exec_ratio = 36.0 / 144.0 if f4 else 16.0 / 36.0
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
            "metric": "stylized frames/sec @1280x720, per-frame hot path (mask + warp + assembly + net + deprocess + PNG encode on the GPU) with inputs "
                      "resident in HBM; file->PNG rate of the product CLI (BASELINE.json's end-to-end) in `end_to_end_fps` / `e2e`, PSNR vs CPU ref in `parity`",
            "value": round(fps, 3), "unit": "frames/s",
            "value_note": "`value` = in-HBM rate of the whole per-frame hot path (contract: inputs resident when the timed region starts); "
                          "the rate BASELINE.json's end-to-end wording means (files in -> PNG files out through bin/fav_stylize) is `end_to_end_fps`",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "1280x720 fused on-GPU consistency check (%s) + min-filter + warp + assemble + transformer net "
                                   "(c9s1-32,d64,d128,R128x5,U2,c3s1-64,U2,c9s1-3, reflect-start pad 40) + deprocess, inputs in HBM, "
                                   "1 independent stream per GPU" % ("4-arg" if args.structure else "3-arg"),
                       "png_encode": ("compute queue, in front of the next frame (as bin/fav_stylize)" if not args.png_async else
                                      "every frame, on the stream's encoder queue next to the next frame's network (fav_stream_encode_png_async)"),
                       "frame": [W, H], "streams": world, "parallelism": f"{world} independent streams, no data-path collective"},
            # `achieved` / `frac`: the MATRIX-PIPE view -- FLOPs the kernel actually executes on the fp32 MFMA pipe (16/36 of the direct
            # convolution's for Winograd F(2x2,3x3)) / time: a utilisation, never above 1.  `algorithmic_*`: SURVEY 8d's per-unit
            # figure (direct-convolution FLOPs) / the same time, which may exceed the peak because fewer multiplies are executed.
            "roofline": {"bound": "mfma", "achieved": round(achieved * exec_ratio, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved * exec_ratio / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_note": traffic_note,
                         "kernel": dom_name,
                         "algorithmic_tflops": round(achieved, 3), "algorithmic_frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                         "executed_over_algorithmic": round(exec_ratio, 4),
                         "note": "frac = executed MFMA FLOPs / time / peak (utilisation of the matrix pipe); algorithmic_frac = direct-convolution FLOPs / time / peak "
                                 "(> 1: faster than any direct fp32 convolution could run)",
                         "avg_launch_us": round(secs / max(1, nl) * 1e6, 2), "launches": nl,
                         "launches_with_residual_join": sum(n for ms, n, macs in dom_join) if dom_name.startswith("conv3_wino") else 0,
                         "executed_mults_per_direct_mult": "36/144 (F(4x4,3x3))" if f4 else "16/36 (F(2x2,3x3))",
                         "avg_launch_us_with_join": round(sum(ms for ms, n, macs in dom_join) / max(1, sum(n for ms, n, macs in dom_join)) * 1e3, 2) if dom_name.startswith("conv3_wino") and dom_join else None,
                         "avg_launch_us_without_join": round(sum(ms for ms, n, macs in dom_plain) / max(1, sum(n for ms, n, macs in dom_plain)) * 1e3, 2) if dom_name.startswith("conv3_wino") and dom_plain else None,
                         "frac_of_launches_without_join": (round(sum(2.0 * macs * n for ms, n, macs in dom_plain) / (sum(ms for ms, n, macs in dom_plain) / 1e3) / 1e12 * exec_ratio / FP32_MFMA_PEAK_TFLOPS, 4)
                                                           if dom_name.startswith("conv3_wino") and dom_plain else None),
                         "join_note": ("residual joins are launches of their own next to the F(4x4) kernel (since r4s: a join formed inside the convolution costs more than "
                                       "its launch); three of the ten launches have 273-286 units for 252 shares and run as stream-K (a cut unit in two parts)" if f4 else
                                       "three of the ten launches form the residual join z = skip + IN(branch) of the previous block while staging their input and write it out "
                                       "(the work of the res_add launches they replace: 33 MB more to read, 33 MB to write, one more operand transform); `frac` is over all ten"),
                         "timed_with": "HIP events (no system fence) around every convolution launch of every %d-th step of the timed region (%d of %d steps)" % (max(1, pe), n_prof_steps, args.steps),
                         "conv_stack_ms_per_frame": round(conv_ms, 4),
                         "conv_stack_tflops": round(FLOP_PER_FRAME / (conv_ms * 1e-3) / 1e12, 3) if conv_ms > 0 else None,
                         "conv_stack_frac": round(FLOP_PER_FRAME / (conv_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4) if conv_ms > 0 else None,
                         "per_kernel_ms_tflops": {k: [round(v[0], 4), round(v[1] / (v[0] * 1e-3) / 1e12, 1)] for k, v in per_kernel.items()}},
        }
        line["extra"] = extra
        if "frames_per_s_4arg_structure_mode_lookahead" in extra:      # `value`'s sibling: the checker mode makeOptFlow_deepflow.sh:59-60 runs
            line["config"]["value_in_4arg_structure_mode"] = extra["frames_per_s_4arg_structure_mode_lookahead"]
            line["config"]["value_in_4arg_structure_mode_over_value"] = round(extra["frames_per_s_4arg_structure_mode_lookahead"] / fps, 4)
        if "frames_per_s_3arg_mode" in extra:                           # `value` is the 4-argument mode: its 3-argument sibling (the headline of rounds 1-5)
            line["config"]["value_in_3arg_mode"] = extra["frames_per_s_3arg_mode"]
            line["config"]["value_over_value_in_3arg_mode"] = round(fps / extra["frames_per_s_3arg_mode"], 4)
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
            o1, u1 = pst.next_frame_flow(frames[1], bws[1], fws[1], use_structure=bool(args.structure), want_u8=True)
            torch.cuda.synchronize(); net.check()
            base, parity = cpu_baseline_and_parity(layers, frames_h[1], bw_h[1], fw_h[1], o0.cpu().numpy(), o1.cpu().numpy(), u1.cpu().numpy(),
                                                   pst.last_mask().cpu().numpy(), structure=int(bool(args.structure)))
            ref = reference_checker_baseline(bw_h[1], fw_h[1], frames_h[1])
            if ref is not None:
                base["reference_consistency_checker"] = ref
            line["cpu_baseline"] = base
            line["parity"] = parity
            del pst
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["roofline_hbm"] = hbm_kernel_block(dev)
            except Exception as e:
                line["roofline_hbm"] = {"error": repr(e)[:300]}
        line["extra"]["png_bytes_per_frame_in_timed_region"] = int(png_n.item())
        if not args.no_e2e:
            del stream; torch.cuda.synchronize()
            line["e2e"] = e2e_block(ckpt, net, frames_h, bw_h, fw_h, world=world, sustained_frames=args.sustained_frames, quick=args.quick_e2e, structure=int(bool(args.structure)))
            head = line["e2e"].get("gpu_png", {}) if isinstance(line["e2e"], dict) else {}
            line["end_to_end_fps"] = head.get("fps")
            # the bytes behind the end-to-end numbers: PNGs of ALL e2e legs that differ from (or are missing against) the in-process run
            line["png_mismatch_frames"] = line["e2e"].get("png_mismatch_frames_all_legs", head.get("png_mismatch_frames")) if isinstance(line["e2e"], dict) else None
            line["png_checked_frames"] = line["e2e"].get("png_checked_all_legs", head.get("png_checked")) if isinstance(line["e2e"], dict) else None
            line["end_to_end_note"] = ("file -> PNG, bin/fav_stylize%s, %d frames per stream from /dev/shm (BASELINE.json's end-to-end metric); `value` is the in-HBM "
                                       "rate the bench contract defines" % ("" if world == 1 else " -streams ... -gpus %d (one worker process per GPU)" % world, 300))
        print(json.dumps(line), flush=True)
        try:
            os.remove(ckpt)
        except OSError:
            pass
    if use_dist:
        # rank 0 may still be driving the product launcher (e2e at --gpus N: worker processes on ALL GPUs): the other ranks wait on
        # the HOST for its sentinel file -- a device-side barrier would park a spinning RCCL kernel on their GPUs, next to the
        # persistent convolution grids of the workers (include/fav.h: those own the device)
        flag = os.path.join(tempfile.gettempdir(), "fav_bench_done_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none")))
        if rank == 0:
            open(flag, "w").close()
        else:
            t_wait = time.time()
            while not os.path.exists(flag) and time.time() - t_wait < 900:
                time.sleep(0.05)
        dist.barrier()
        if rank == 0:
            try:
                os.remove(flag)
            except OSError:
                pass
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
