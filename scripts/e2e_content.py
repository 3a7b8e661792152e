"""Content check of the file -> PNG runs of bin/fav_stylize: every PNG the CLI wrote is compared BYTE FOR BYTE with the file the
in-process fav_stream_* run of the same inputs produces (the GPU path is bit-deterministic: tests/test_gpu_parity.py), so a lost,
duplicated or stale frame in the CLI's host-ordered protocol (size word polled in host-mapped memory, exact-size DMA on a third
queue, two-slot PNG ring, three-deep input ring, two frames of look-ahead) cannot hide behind `png_written == frames`.
What the CLI replaces: func_save_image (fast_artistic_video.lua:160-170) inside the loop of fast_artistic_video_core.lua:194-211.

TEST / BENCH INFRASTRUCTURE (used by bench.py's e2e block and tests/test_gpu_cli.py): it drives the product through its C ABI
(fav_amd) and its CLI, and uses oracle/ only to WRITE input files (.flo / .ppm) and, in the certainty mode, the reference's own
checker binary (oracle/_ref/consistencyChecker) to make the .pgm files stylizeVideo_deepflow.sh:87-96 passes.
"""
import hashlib
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CHECKER = os.path.join(ROOT, "oracle", "_ref", "consistencyChecker")

try:
    import xxhash

    def _digest(b):
        return xxhash.xxh3_128_digest(b)
except Exception:      # pragma: no cover
    def _digest(b):
        return hashlib.blake2b(b, digest_size=16).digest()


def age_files(d, seconds=30.0):
    """mark every regular file under d as finished `seconds` ago: fav_stylize takes a file younger than -poll_settle (1 s, what
    fast_artistic_video/utils.lua:79's `sleep 1` buys) only once it has stopped changing for that long (host/fav_poll.h) -- inputs a
    test or a bench has just written are FINISHED inputs and say so through their modification time"""
    import time
    t = time.time() - seconds
    for base, _, files in os.walk(d):
        for f in files:
            p = os.path.join(base, f)
            if not os.path.islink(p):
                os.utime(p, (t, t))


def make_clip_dir(d, name, frames_h, bw_h, fw_h, nframes, O, cert=False):
    """RAM-backed clip: `ring` distinct frames / flow pairs under d/src, the clip's files are symlinks onto them (frame i -> source
    i % ring).  cert: also reliable_<i>_<i-1>.pgm, written by the REFERENCE's checker in the 4-argument form of
    makeOptFlow_deepflow.sh:59-60 (`consistencyChecker backward forward out.pgm frame.ppm`)."""
    os.makedirs(f"{d}/{name}/flow")
    ring = len(frames_h)
    if not os.path.isdir(d + "/src"):
        os.makedirs(d + "/src")
        for k in range(ring):
            O.write_pnm(f"{d}/src/f{k}.ppm", frames_h[k]); O.write_flo(f"{d}/src/b{k}.flo", bw_h[k]); O.write_flo(f"{d}/src/w{k}.flo", fw_h[k])
    if cert and not os.path.exists(f"{d}/src/r0.pgm"):
        for k in range(ring):
            subprocess.check_call([REF_CHECKER, f"{d}/src/b{k}.flo", f"{d}/src/w{k}.flo", f"{d}/src/r{k}.pgm", f"{d}/src/f{k}.ppm"],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    age_files(d + "/src")
    for i in range(1, nframes + 1):
        os.symlink(f"{d}/src/f{i % ring}.ppm", f"{d}/{name}/frame_{i:05d}.ppm")
        if i > 1:
            os.symlink(f"{d}/src/b{i % ring}.flo", f"{d}/{name}/flow/backward_{i}_{i-1}.flo"); os.symlink(f"{d}/src/w{i % ring}.flo", f"{d}/{name}/flow/forward_{i-1}_{i}.flo")
            if cert:
                os.symlink(f"{d}/src/r{i % ring}.pgm", f"{d}/{name}/flow/reliable_{i}_{i-1}.pgm")


class Reference:
    """the in-process run: per frame the digest and size of the PNG file fav_stream_encode_png produces; whole files for `keep`"""

    def __init__(self):
        self.digest, self.size, self.kept, self.states = [], [], {}, {}


def reference_run(fav_amd, net, frames_h, bw_h, fw_h, nframes, mode, masks_h=None, keep=(), keep_states=(), device=0):
    """mode: '3arg' / '4arg' = fav_stream_next_frame_flow (fused check, use_structure 0 / 1), 'cert' = fav_stream_next_frame_cert with
    masks_h[k] (u8 [H][W]).  Frame i (1-based) uses source i % ring, as make_clip_dir lays the files out.  keep: 1-based frame numbers
    whose PNG bytes are kept; keep_states: frames whose float state BEFORE and AFTER the frame is kept (teacher-forced oracle checks)."""
    import torch
    dev = torch.device("cuda", device)
    ring = len(frames_h)
    H, W = frames_h[0].shape[:2]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    fr, bw, fw = [T(a) for a in frames_h], [T(a) for a in bw_h], [T(a) for a in fw_h]
    mk = [T(a) for a in masks_h] if masks_h is not None else None
    st = fav_amd.Stream(net, H, W)
    png_out, png_n = st.png_buffers()
    ref = Reference()
    keep, keep_states = set(keep), set(keep_states)
    for i in range(1, nframes + 1):
        k = i % ring
        if i in keep_states and i > 1:
            before = st.state().cpu().numpy()
        if i == 1:
            st.first_frame(fr[k], want_f32=False)
        elif mode == "cert":
            st.next_frame_cert(fr[k], bw[k], mk[k], want_f32=False)
        else:
            st.next_frame_flow(fr[k], bw[k], fw[k], use_structure=(mode == "4arg"), want_f32=False)
        st.encode_png_into(png_out, png_n)
        n = int(png_n.item())
        data = png_out[:n].cpu().numpy().tobytes()
        ref.digest.append(_digest(data)); ref.size.append(n)
        if i in keep:
            ref.kept[i] = data
        if i in keep_states:
            ref.states[i] = (before if i > 1 else None, st.state().cpu().numpy())
    net.check()
    return ref


def verify_dir(out_dir, prefix, ref, nframes):
    """compare <out_dir>/<prefix>-%05d.png, frames 1..nframes, with the reference run; returns the counts the bench line carries"""
    missing, mismatch, first_bad = 0, 0, None
    for i in range(1, nframes + 1):
        p = os.path.join(out_dir, "%s-%05d.png" % (prefix, i))
        try:
            with open(p, "rb") as f:
                data = f.read()
        except OSError:
            missing += 1; first_bad = first_bad or i
            continue
        if len(data) != ref.size[i - 1] or _digest(data) != ref.digest[i - 1]:
            mismatch += 1; first_bad = first_bad or i
    extra = len([f for f in os.listdir(out_dir) if f.endswith(".png")]) - (nframes - missing) if os.path.isdir(out_dir) else 0
    return {"png_checked": nframes, "png_mismatch_frames": mismatch, "png_missing_frames": missing, "png_unexpected_files": max(0, extra),
            "first_bad_frame": first_bad}


def decode_png(data):
    import io
    from PIL import Image
    return np.array(Image.open(io.BytesIO(data)).convert("RGB"))


def verify_dir_pixels(out_dir, prefix, ref, frames):
    """for legs whose FILES differ by construction (-png_encoder host: zlib's deflate stream): the decoded pixels of the listed
    frames against the decoded reference files (ref.kept)"""
    bad = 0
    for i in frames:
        p = os.path.join(out_dir, "%s-%05d.png" % (prefix, i))
        try:
            with open(p, "rb") as f:
                got = decode_png(f.read())
        except OSError:
            bad += 1
            continue
        bad += 0 if np.array_equal(got, decode_png(ref.kept[i])) else 1
    return {"png_checked_by_pixels": len(list(frames)), "png_mismatch_frames": bad}
