"""Drop-in executables on the GPU: `consistencyChecker` (same argv / same PGM bytes as the reference's)
and `fav_stylize` (the flag contract of fast_artistic_video.lua, file-name patterns of
stylizeVideo_deepflow.sh:87-96).  Outputs are compared with the oracle's per-frame loop."""
import os
import subprocess

import numpy as np
import pytest

from fav_amd import synth, t7

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "fast-artistic-videos_amd", "bin")


def _write_clip(oracle, d, h, w, n, seed):
    frames, bws, fws = [], [None], [None]
    os.makedirs(d / "flow", exist_ok=True)
    for i in range(1, n + 1):
        f = synth.smooth_frame(h, w, seed + i); frames.append(f)
        oracle.write_pnm(str(d / f"frame_{i:05d}.ppm"), f)
        if i > 1:
            bw = synth.backward_flow(h, w, seed + 100 + i); fw = synth.forward_flow_from_backward(bw, seed + 200 + i)
            bws.append(bw); fws.append(fw)
            oracle.write_flo(str(d / "flow" / f"backward_{i}_{i-1}.flo"), bw)
            oracle.write_flo(str(d / "flow" / f"forward_{i-1}_{i}.flo"), fw)
    _age(d)
    return frames, bws, fws


def _age(d):
    """finished inputs say so through their modification time (host/fav_poll.h: anything younger than -poll_settle is watched until it
    has stopped changing for that long, like the reference's `sleep 1`, utils.lua:79)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import e2e_content
    e2e_content.age_files(str(d))


def _stop_helper(run_dir):
    """terminate the checker's resident helper whose socket lives under run_dir/fav-cc (its pid is in the lock file)"""
    import signal, time
    import glob
    locks = glob.glob(os.path.join(run_dir, "fav-cc", "gpu0*.lock"))      # (gpu0.lock, or gpu0-<hash>.lock when a device-visibility variable is set)
    if not locks:
        return None
    lock = locks[0]
    try:
        pid = int(open(lock).read().split()[0])
    except (ValueError, IndexError):
        return None
    try:
        os.kill(pid, signal.SIGTERM)
    except ProcessLookupError:
        return pid
    for _ in range(200):
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            break
        time.sleep(0.02)
    return pid


def test_consistency_checker_resident_helper(oracle, favlib, tmp_path):
    """A single call of the four-argument form costs a HIP start-up per process; makeOptFlow_deepflow.sh:59-60 makes two per frame.  The
    first call leaves a resident helper behind (own session, socket in a 0700 directory), later calls hand it their argv: same bytes,
    same exit codes, relative paths resolved against the CALLER's directory, sizes may change between calls; FAV_CC_DAEMON=0 computes
    in the calling process as before; SIGTERM (or FAV_CC_IDLE_S seconds without a request) ends it and removes the socket."""
    import time
    run = tmp_path / "run"; run.mkdir(mode=0o700)
    env = dict(os.environ, XDG_RUNTIME_DIR=str(run), FAV_CC_IDLE_S="60")
    exe = os.path.join(BIN, "consistencyChecker")
    cases = []
    for k, (h, w) in enumerate([(90, 130), (90, 130), (72, 100)]):
        bw = synth.backward_flow(h, w, 10 + k); fw = synth.forward_flow_from_backward(bw, 20 + k); img = synth.smooth_frame(h, w, 30 + k)
        d = tmp_path / ("c%d" % k); d.mkdir()
        oracle.write_flo(str(d / "a.flo"), bw); oracle.write_flo(str(d / "b.flo"), fw); oracle.write_pnm(str(d / "i.ppm"), img)
        cases.append((d, h, w, bw, fw, img))
    try:
        times = []
        for k, (d, h, w, bw, fw, img) in enumerate(cases):
            for four in (False, True):
                t0 = time.perf_counter()
                r = subprocess.run([exe, "a.flo", "b.flo", "o.pgm"] + (["i.ppm"] if four else []), capture_output=True, text=True, cwd=str(d), env=env)   # relative paths
                times.append(time.perf_counter() - t0)
                assert r.returncode == 0 and r.stdout == "o.pgm", (r.returncode, r.stdout, r.stderr)
                want = oracle.consistency(bw, fw, img if four else None)
                assert open(d / "o.pgm", "rb").read() == b"P5\n%d %d\n255\n" % (w, h) + want.tobytes(), (k, four)
        print("consistencyChecker wall time per call: first (starts the helper) %.3f s, then %s" % (times[0], " ".join("%.3f" % t for t in times[1:])))
        assert list((run / "fav-cc").glob("gpu0*.sock"))
        assert min(times[1:]) < times[0]                                 # later calls do not pay for a GPU context
        # failures come back as the exit code and the message
        r = subprocess.run([exe, "a.flo", "missing.flo", "o.pgm"], capture_output=True, text=True, cwd=str(cases[0][0]), env=env)
        assert r.returncode != 0 and "missing.flo" in r.stderr and r.stdout == ""
        # ... and the helper is still there afterwards; the in-process form gives the same bytes
        d, h, w, bw, fw, img = cases[0]
        r = subprocess.run([exe, "a.flo", "b.flo", "o2.pgm", "i.ppm"], capture_output=True, text=True, cwd=str(d), env=dict(env, FAV_CC_DAEMON="0"))
        assert r.returncode == 0 and open(d / "o2.pgm", "rb").read() == b"P5\n%d %d\n255\n" % (w, h) + oracle.consistency(bw, fw, img).tobytes()
        r = subprocess.run([exe, "a.flo", "b.flo", "o3.pgm", "i.ppm"], capture_output=True, text=True, cwd=str(d), env=env)
        assert r.returncode == 0 and open(d / "o3.pgm", "rb").read() == open(d / "o2.pgm", "rb").read()
        # a caller that is ANOTHER BUILD -- here: other FAV_* settings, which the library reads once per process; the same happens after a
        # rebuild of the executable or the library (the build id in every request: executable + libfav.so + FAV_* environment) -- is not served
        # by the old helper: it computes in its own process, the old helper leaves, and the next such call starts a fresh one
        lock = list((run / "fav-cc").glob("gpu0*.lock"))[0]
        old_pid = int(lock.read_text().split()[0])
        env2 = dict(env, FAV_TEST_BUILD_TAG="another-build")
        r = subprocess.run([exe, "a.flo", "b.flo", "o4.pgm", "i.ppm"], capture_output=True, text=True, cwd=str(d), env=env2)
        assert r.returncode == 0 and r.stdout == "o4.pgm" and open(d / "o4.pgm", "rb").read() == open(d / "o2.pgm", "rb").read(), (r.stdout, r.stderr)
        for _ in range(100):
            try:
                os.kill(old_pid, 0); time.sleep(0.05)
            except ProcessLookupError:
                break
        else:
            raise AssertionError("the helper of the other build is still there")
        r = subprocess.run([exe, "a.flo", "b.flo", "o5.pgm", "i.ppm"], capture_output=True, text=True, cwd=str(d), env=env2)      # starts the new build's helper
        assert r.returncode == 0 and open(d / "o5.pgm", "rb").read() == open(d / "o2.pgm", "rb").read()
        new_pid = int(lock.read_text().split()[0])
        assert new_pid != old_pid
        t0 = time.perf_counter()
        r = subprocess.run([exe, "a.flo", "b.flo", "o6.pgm", "i.ppm"], capture_output=True, text=True, cwd=str(d), env=env2)      # ... and is served by it
        assert r.returncode == 0 and open(d / "o6.pgm", "rb").read() == open(d / "o2.pgm", "rb").read() and int(lock.read_text().split()[0]) == new_pid
        assert not list(d.glob("*.tmp*"))                                   # per-process temporary names are renamed or removed
    finally:
        pid = _stop_helper(str(run))
    assert pid is not None
    assert not list((run / "fav-cc").glob("gpu0*.sock"))                # SIGTERM: the helper removes its socket


def test_consistency_checker_binary(oracle, favlib, tmp_path, monkeypatch):
    monkeypatch.setenv("FAV_CC_DAEMON", "0")                            # (the process-per-call form; the resident helper has its own test)
    h, w = 90, 130
    bw = synth.backward_flow(h, w, 1); fw = synth.forward_flow_from_backward(bw, 2); img = synth.smooth_frame(h, w, 3)
    a, b, i, o = (str(tmp_path / n) for n in ("a.flo", "b.flo", "i.ppm", "o.pgm"))
    oracle.write_flo(a, bw); oracle.write_flo(b, fw); oracle.write_pnm(i, img)
    exe = os.path.join(BIN, "consistencyChecker")
    r = subprocess.run([exe, a, b, o], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout == o                       # consistencyChecker.cpp:166 prints the output path
    assert open(o, "rb").read() == b"P5\n%d %d\n255\n" % (w, h) + oracle.consistency(bw, fw).tobytes()
    r = subprocess.run([exe, a, b, o, i], capture_output=True, text=True)
    assert r.returncode == 0
    assert open(o, "rb").read() == b"P5\n%d %d\n255\n" % (w, h) + oracle.consistency(bw, fw, img).tobytes()
    assert subprocess.run([exe, a, str(tmp_path / "missing.flo"), o]).returncode != 0
    # list mode (additive): one argument line per pair, one GPU context; same bytes as the single calls, output paths echoed per line
    lst = str(tmp_path / "pairs.txt")
    o3, o4 = str(tmp_path / "b3.pgm"), str(tmp_path / "b4.pgm")
    open(lst, "w").write(f"# pairs\n{a} {b} {o3}\n\n{a} {b} {o4} {i}\n")
    r = subprocess.run([exe, "-batch", lst], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.split() == [o3, o4], (r.stdout, r.stderr)
    assert open(o3, "rb").read() == b"P5\n%d %d\n255\n" % (w, h) + oracle.consistency(bw, fw).tobytes()
    assert open(o4, "rb").read() == b"P5\n%d %d\n255\n" % (w, h) + oracle.consistency(bw, fw, img).tobytes()
    open(lst, "w").write(f"{a} {b} {o3}\n{a} {tmp_path / 'missing.flo'} {o4}\n")
    assert subprocess.run([exe, "-batch", lst], capture_output=True).returncode != 0


@pytest.mark.parametrize("fused", [False, True])
def test_fav_stylize_matches_oracle_loop(oracle, favlib, tmp_path, golden_dir, fused):
    from PIL import Image
    h, w, n = 48, 64, 4
    frames, bws, fws = _write_clip(oracle, tmp_path, h, w, n, 40)
    model = os.path.join(golden_dir, "tiny_model.t7")
    checker = os.path.join(BIN, "consistencyChecker")
    masks = [None, None]
    for i in range(2, n + 1):     # makeOptFlow_deepflow.sh:59
        if not fused:
            subprocess.check_call([checker, str(tmp_path / "flow" / f"backward_{i}_{i-1}.flo"), str(tmp_path / "flow" / f"forward_{i-1}_{i}.flo"),
                                   str(tmp_path / "flow" / f"reliable_{i}_{i-1}.pgm"), str(tmp_path / f"frame_{i:05d}.ppm")], stdout=subprocess.DEVNULL,
                                  env=dict(os.environ, FAV_CC_DAEMON="0"))      # (no resident helper left behind by this test)
        masks.append(oracle.consistency(bws[i - 1], fws[i - 1], frames[i - 1]))
    _age(tmp_path)
    cmd = [os.path.join(BIN, "fav_stylize"), "-input_pattern", str(tmp_path / "frame_%05d.ppm"),
           "-flow_pattern", str(tmp_path / "flow" / "backward_[%d]_{%d}.flo"),
           "-occlusions_pattern", str(tmp_path / "flow" / "reliable_[%d]_{%d}.pgm"),
           "-output_prefix", str(tmp_path / "out" / "out"), "-backend", "cuda", "-use_cudnn", "1", "-gpu", "0",
           "-model_vid", model, "-model_img", "self"]
    if fused:
        cmd += ["-forward_flow_pattern", str(tmp_path / "flow" / "forward_{%d}_[%d].flo")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Model loaded." in r.stdout and r.stdout.count("Writing output image to") == n
    ref = oracle.Stylizer(t7.extract_layers(t7.load(model)["model"]))
    for i in range(1, n + 1):
        f01 = np.transpose(frames[i - 1], (2, 0, 1)).astype(np.float32) / np.float32(255)
        out = ref.first(f01) if i == 1 else ref.next(f01, bws[i - 1], masks[i].astype(np.float32) / np.float32(255))
        png = np.asarray(Image.open(str(tmp_path / "out" / f"out-{i:05d}.png")))
        want = oracle.to_u8_hwc(out)
        assert png.shape == want.shape
        assert np.abs(png.astype(int) - want.astype(int)).max() <= 1, f"frame {i}"
    assert not os.path.exists(str(tmp_path / "out" / f"out-{n+1:05d}.png"))     # loop stops at the first missing frame
    if not fused:      # additive -temporal_eval_file: the temporal-consistency number of -evaluate (fav.lua:128-151)
        ev = str(tmp_path / "temporal.txt")
        r = subprocess.run(cmd + ["-temporal_eval_file", ev], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        vals = [float(x) for x in open(ev).read().split("\n")[0].split(";")]
        ref2 = oracle.Stylizer(t7.extract_layers(t7.load(model)["model"])); prev = None; want_t = []
        for i in range(1, n + 1):
            f01 = np.transpose(frames[i - 1], (2, 0, 1)).astype(np.float32) / np.float32(255)
            m01 = masks[i].astype(np.float32) / np.float32(255) if i > 1 else None
            out = ref2.first(f01) if i == 1 else ref2.next(f01, bws[i - 1], m01)
            want_t.append(0.0 if i == 1 else oracle.temporal_loss(prev, out, bws[i - 1], m01))
            prev = out
        assert len(vals) == n and vals[0] == 0.0
        assert all(abs(a - b) <= 0.05 * b + 1e-7 for a, b in zip(vals[1:], want_t[1:])), (vals, want_t)
        assert abs(float(open(ev).read().split("\n")[1]) - sum(vals) / n) <= 1e-6 * max(sum(vals) / n, 1e-9)


@pytest.mark.parametrize("encoder", ["gpu", "host"])
def test_fav_stylize_frame_size_not_a_multiple_of_four(oracle, favlib, tmp_path, golden_dir, encoder):
    """70x50 frames (stylizeVideo_deepflow.sh:72-78 lets the user pick any w:h): the PNGs have the network's output size 72x52, the
    recurrent state is that larger image warped on the flow's grid (BilinearSamplerBDHW.lua:71), -continue_with reloads it"""
    from PIL import Image
    h, w, n = 50, 70, 3
    frames, bws, fws = _write_clip(oracle, tmp_path, h, w, n, 60)
    model = os.path.join(golden_dir, "tiny_model.t7")
    cmd = [os.path.join(BIN, "fav_stylize"), "-input_pattern", str(tmp_path / "frame_%05d.ppm"), "-flow_pattern", str(tmp_path / "flow" / "backward_[%d]_{%d}.flo"),
           "-forward_flow_pattern", str(tmp_path / "flow" / "forward_{%d}_[%d].flo"), "-structure", "0", "-output_prefix", str(tmp_path / "out" / "out"), "-gpu", "0",
           "-model_vid", model, "-model_img", "self", "-png_encoder", encoder]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ref = oracle.Stylizer(t7.extract_layers(t7.load(model)["model"]))
    for i in range(1, n + 1):
        f01 = np.transpose(frames[i - 1], (2, 0, 1)).astype(np.float32) / np.float32(255)
        out = ref.first(f01) if i == 1 else ref.next(f01, bws[i - 1], oracle.consistency(bws[i - 1], fws[i - 1]).astype(np.float32) / np.float32(255))
        png = np.asarray(Image.open(str(tmp_path / "out" / f"out-{i:05d}.png")))
        assert png.shape == (52, 72, 3) == oracle.to_u8_hwc(out).shape
        assert np.abs(png.astype(int) - oracle.to_u8_hwc(out).astype(int)).max() <= 1, f"frame {i}"
    # resume from the (larger) PNG of frame 2
    os.remove(tmp_path / "out" / "out-00003.png")
    r = subprocess.run(cmd + ["-continue_with", "3"], capture_output=True, text=True)
    assert r.returncode == 0 and os.path.exists(tmp_path / "out" / "out-00003.png"), r.stderr


@pytest.mark.parametrize("encoder", ["gpu", "host"])
def test_fav_stylize_four_argument_checker_mode_with_two_frames_of_look_ahead(oracle, favlib, tmp_path, golden_dir, encoder):
    """-structure 1 = consistencyChecker's 4-argument form (makeOptFlow_deepflow.sh:59-60: the video driver always passes the frame):
    the CLI computes those masks two frames ahead on the side queues.  Seven frames -- more than the look-ahead, the mask slots and the
    device input ring hold -- against the oracle with the 4-argument masks."""
    from PIL import Image
    h, w, n = 48, 64, 7
    frames, bws, fws = _write_clip(oracle, tmp_path, h, w, n, 90)
    model = os.path.join(golden_dir, "tiny_model.t7")
    cmd = [os.path.join(BIN, "fav_stylize"), "-input_pattern", str(tmp_path / "frame_%05d.ppm"), "-flow_pattern", str(tmp_path / "flow" / "backward_[%d]_{%d}.flo"),
           "-forward_flow_pattern", str(tmp_path / "flow" / "forward_{%d}_[%d].flo"), "-structure", "1", "-output_prefix", str(tmp_path / "out" / "out"), "-gpu", "0",
           "-model_vid", model, "-model_img", "self", "-png_encoder", encoder]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.count("Writing output image to") == n, r.stderr
    ref = oracle.Stylizer(t7.extract_layers(t7.load(model)["model"]))
    differs = 0
    for i in range(1, n + 1):
        f01 = np.transpose(frames[i - 1], (2, 0, 1)).astype(np.float32) / np.float32(255)
        if i == 1:
            out = ref.first(f01)
        else:
            m4 = oracle.consistency(bws[i - 1], fws[i - 1], frames[i - 1])
            differs += int((m4 != oracle.consistency(bws[i - 1], fws[i - 1])).sum())
            out = ref.next(f01, bws[i - 1], m4.astype(np.float32) / np.float32(255))
        png = np.asarray(Image.open(str(tmp_path / "out" / f"out-{i:05d}.png")))
        assert np.abs(png.astype(int) - oracle.to_u8_hwc(out).astype(int)).max() <= 1, f"frame {i}"
    assert differs > 0          # the structure term changes some mask pixels of this clip: the test can tell the two modes apart


def test_fav_stylize_flag_contract(favlib, tmp_path, golden_dir):
    exe = os.path.join(BIN, "fav_stylize")
    model = os.path.join(golden_dir, "tiny_model.t7")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode != 0 and "Must give -input_pattern" in r.stderr                       # fast_artistic_video.lua:177-179
    r = subprocess.run([exe, "-input_pattern", "x_%05d.ppm"], capture_output=True, text=True)
    assert r.returncode != 0 and "Must give -flow_pattern and -occlusions_pattern" in r.stderr  # :180-182
    r = subprocess.run([exe, "-input_pattern", "x", "-flow_pattern", "a", "-occlusions_pattern", "b", "-gpu", "-1"], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU backend" in r.stderr
    r = subprocess.run([exe, "-input_pattern", "x", "-flow_pattern", "a", "-occlusions_pattern", "b", "-model_vid", str(tmp_path / "none.t7")],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "Could not load model" in r.stderr                             # core.lua:41
    # -model_img <file>: the first frame goes through a separate image model (core.lua:59-66,146)
    from PIL import Image
    import oracle as O
    pi = str(tmp_path / "img.t7")
    t7.make_synthetic_checkpoint(pi, arch="c9s1-8,d16,d32,R32,R32,u16,u8,c9s1-3", seed=5, in_channels=3)
    fr = synth.smooth_frame(48, 64, 9); O.write_pnm(str(tmp_path / "one_00001.ppm"), fr)
    r = subprocess.run([exe, "-input_pattern", str(tmp_path / "one_%05d.ppm"), "-create_inconsistent", "-model_vid", model, "-model_img", pi,
                        "-gpu", "0", "-output_prefix", str(tmp_path / "img" / "out")], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.count("Model loaded.") == 2, r.stderr
    want = O.to_u8_hwc(O.deprocess(O.net_forward(t7.extract_layers(t7.load(pi)["model"]),
                                                 O.preprocess(np.transpose(fr, (2, 0, 1)).astype(np.float32) / np.float32(255)))))
    got = np.asarray(Image.open(str(tmp_path / "img" / "out-00001.png")))
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1
    # th shim used by the unmodified shell drivers
    shim = os.path.join(ROOT, "fast-artistic-videos_amd", "host", "th")
    r = subprocess.run([shim, "fast_artistic_video.lua", "-input_pattern", str(tmp_path / "no_%05d.ppm"), "-create_inconsistent",
                        "-model_vid", model, "-model_img", "self", "-gpu", "0"], capture_output=True, text=True)
    assert r.returncode == 0 and "Model loaded." in r.stdout
    # -model_img defaults to the reference's models/checkpoint-candy-image.t7 (fast_artistic_video.lua:24): absent here -> core.lua:41's error
    r = subprocess.run([exe, "-input_pattern", str(tmp_path / "no_%05d.ppm"), "-create_inconsistent", "-model_vid", model, "-gpu", "0"],
                       capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode != 0 and "Could not load model" in r.stderr and "checkpoint-candy-image.t7" in r.stderr


def test_fav_stylize_loop_flags(oracle, favlib, tmp_path, golden_dir):
    """-num_frames, -create_inconsistent and -continue_with (fast_artistic_video_core.lua:189-204; the resume path reloads
    the previous PNG as the 8-bit recurrent state -- documented deviation, the reference's video CLI cannot resume)"""
    from PIL import Image
    h, w, n = 48, 64, 4
    frames, bws, fws = _write_clip(oracle, tmp_path, h, w, n, 70)
    model = os.path.join(golden_dir, "tiny_model.t7")
    layers = t7.extract_layers(t7.load(model)["model"])
    exe = os.path.join(BIN, "fav_stylize")
    common = ["-input_pattern", str(tmp_path / "frame_%05d.ppm"), "-flow_pattern", str(tmp_path / "flow" / "backward_[%d]_{%d}.flo"),
              "-forward_flow_pattern", str(tmp_path / "flow" / "forward_{%d}_[%d].flo"), "-structure", "0", "-model_vid", model, "-model_img", "self", "-gpu", "0"]
    f01 = lambda u8: np.transpose(u8, (2, 0, 1)).astype(np.float32) / np.float32(255)
    # -num_frames 2: only two outputs
    r = subprocess.run([exe] + common + ["-output_prefix", str(tmp_path / "a" / "out"), "-num_frames", "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.exists(tmp_path / "a" / "out-00002.png") and not os.path.exists(tmp_path / "a" / "out-00003.png")
    # -create_inconsistent: every frame is stylised without a prior (func_is_single_image, fast_artistic_video.lua:172)
    r = subprocess.run([exe, "-input_pattern", str(tmp_path / "frame_%05d.ppm"), "-create_inconsistent", "-model_vid", model, "-model_img", "self", "-gpu", "0",
                        "-output_prefix", str(tmp_path / "b" / "out")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for i in range(1, n + 1):
        want = oracle.to_u8_hwc(oracle.Stylizer(layers).first(f01(frames[i - 1])))
        got = np.asarray(Image.open(str(tmp_path / "b" / f"out-{i:05d}.png")))
        assert np.abs(got.astype(int) - want.astype(int)).max() <= 1
    # -continue_with 3 after the 2-frame run: frame 3 warps the reloaded out-00002.png
    r = subprocess.run([exe] + common + ["-output_prefix", str(tmp_path / "a" / "out"), "-continue_with", "3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    prev = np.asarray(Image.open(str(tmp_path / "a" / "out-00002.png")))
    ref = oracle.Stylizer(layers); ref.last = f01(prev)
    m = oracle.consistency(bws[2], fws[2])
    want = oracle.to_u8_hwc(ref.next(f01(frames[2]), bws[2], m.astype(np.float32) / np.float32(255)))
    got = np.asarray(Image.open(str(tmp_path / "a" / "out-00003.png")))
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1
    assert os.path.exists(tmp_path / "a" / "out-00004.png")


def test_fav_stylize_backward(oracle, favlib, tmp_path, golden_dir):
    """-backward (fast_artistic_video_core.lua:189-191): the loop runs i = num_frames-1, ..., 1 with the SAME file-name rule
    ({i-1}, [i]); frame num_frames-1 has no previous output (the reference dereferences a nil last_frame_stylized there: here it is
    stylised without a prior, as -continue_with without a saved frame), frame 1 is a single image by fast_artistic_video.lua:172."""
    from PIL import Image
    h, w, n = 48, 64, 4
    frames, bws, fws = _write_clip(oracle, tmp_path, h, w, n, 90)
    model = os.path.join(golden_dir, "tiny_model.t7")
    layers = t7.extract_layers(t7.load(model)["model"])
    f01 = lambda u8: np.transpose(u8, (2, 0, 1)).astype(np.float32) / np.float32(255)
    r = subprocess.run([os.path.join(BIN, "fav_stylize"), "-input_pattern", str(tmp_path / "frame_%05d.ppm"), "-flow_pattern", str(tmp_path / "flow" / "backward_[%d]_{%d}.flo"),
                        "-forward_flow_pattern", str(tmp_path / "flow" / "forward_{%d}_[%d].flo"), "-structure", "0", "-model_vid", model, "-model_img", "self", "-gpu", "0",
                        "-output_prefix", str(tmp_path / "o" / "out"), "-backward", "-num_frames", str(n)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert not os.path.exists(tmp_path / "o" / f"out-{n:05d}.png")                       # start index is num_frames - 1
    order = [int(l.split("-")[-1].split(".")[0]) for l in r.stdout.splitlines() if l.startswith("Writing output image")]
    assert order == [3, 2, 1], order
    ref = oracle.Stylizer(layers)
    want = {3: ref.first(f01(frames[2]))}
    m = oracle.consistency(bws[1], fws[1])
    want[2] = ref.next(f01(frames[1]), bws[1], m.astype(np.float32) / np.float32(255))
    want[1] = oracle.Stylizer(layers).first(f01(frames[0]))
    for i in (3, 2, 1):
        got = np.asarray(Image.open(str(tmp_path / "o" / f"out-{i:05d}.png")))
        assert np.abs(got.astype(int) - oracle.to_u8_hwc(want[i]).astype(int)).max() <= 1, i


MODES = ["3arg", "4arg_lookahead", "cert_via_th_shim", "3arg_two_cpus", "4arg_png_overlap"]


@pytest.mark.parametrize("mode", MODES)
def test_fav_stylize_config3_content_at_speed(oracle, favlib, cuda, mode):
    """The bytes behind the end-to-end number (missing 1): BASELINE config 3 -- 300 frames of 1280x720 through
    bin/fav_stylize at full rate, RAM-backed files, canonical architecture -- in the four ways the bench and the reference's drivers
    run it: fused 3-argument check (the headline leg), -structure 1 with two frames of look-ahead, the certainty path with .pgm files
    written by the REFERENCE's checker (what stylizeVideo_deepflow.sh:87-96 passes, through the `th` shim), and confined to two CPUs.
    EVERY PNG the CLI wrote must equal, byte for byte, the file the in-process fav_stream_* run of the same inputs produces (the GPU
    path is bit-deterministic), and frames sampled over the clip, the last one included, are decoded and compared with the oracle
    teacher-forced from the GPU's own previous frame (<= 1 LSB; what the CLI replaces: fast_artistic_video.lua:160-170 inside the
    loop of fast_artistic_video_core.lua:194-211)."""
    import shutil, sys, tempfile, json
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import e2e_content as EC
    if mode == "cert_via_th_shim" and not os.path.exists(EC.REF_CHECKER):
        pytest.skip("oracle/_ref/consistencyChecker not built (needs /root/reference at build time)")
    if not os.path.isdir("/dev/shm"):
        pytest.skip("no /dev/shm")
    H, W, N, ring = 720, 1280, 300, 4
    frames_h = [synth.random_frame(H, W, 1234 + i) for i in range(ring)]
    bw_h = [synth.backward_flow(H, W, 1244 + i) for i in range(ring)]
    fw_h = [synth.forward_flow_from_backward(bw_h[i], 1254 + i) for i in range(ring)]
    d = tempfile.mkdtemp(prefix="fav_c3_", dir="/dev/shm")
    try:
        ckpt = os.path.join(d, "canonical.t7")
        t7.make_synthetic_checkpoint(ckpt, seed=1234)
        layers = t7.extract_layers(t7.load(ckpt)["model"])
        EC.make_clip_dir(d, "s0", frames_h, bw_h, fw_h, N, oracle, cert=(mode == "cert_via_th_shim"))
        pkg = os.path.join(ROOT, "fast-artistic-videos_amd")
        common = ["-input_pattern", f"{d}/s0/frame_%05d.ppm", "-flow_pattern", f"{d}/s0/flow/backward_[%d]_{{%d}}.flo"]
        tail = ["-model_vid", ckpt, "-model_img", "self", "-gpu", "0", "-timing", "1", "-output_prefix", f"{d}/s0/o/out"]
        fused = ["-forward_flow_pattern", f"{d}/s0/flow/forward_{{%d}}_[%d].flo"]
        masks = None
        if mode == "3arg":
            cmd, rmode = [os.path.join(BIN, "fav_stylize")] + common + fused + ["-structure", "0"] + tail, "3arg"
        elif mode == "4arg_lookahead":
            cmd, rmode = [os.path.join(BIN, "fav_stylize")] + common + fused + ["-structure", "1"] + tail, "4arg"
        elif mode == "4arg_png_overlap":       # the PNG encoder on the stream's own queue next to the following frame (fav_stream_encode_png_async)
            cmd, rmode = [os.path.join(BIN, "fav_stylize")] + common + fused + ["-structure", "1", "-png_overlap", "1"] + tail, "4arg"
        elif mode == "cert_via_th_shim":
            cmd = [os.path.join(pkg, "host", "th"), "fast_artistic_video.lua"] + common + ["-occlusions_pattern", f"{d}/s0/flow/reliable_[%d]_{{%d}}.pgm",
                                                                                            "-backend", "cuda", "-use_cudnn", "1"] + tail
            rmode, masks = "cert", [oracle.read_pnm(f"{d}/src/r{k}.pgm") for k in range(ring)]
        else:
            cmd, rmode = ["taskset", "-c", "0-1", os.path.join(BIN, "fav_stylize")] + common + fused + ["-structure", "0"] + tail, "3arg"
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert line["frames"] == N
        sampled = [2, 3, 40, 77, 150, 151, 222, 299, 300] if mode == "3arg" else [2, 150, 300]
        net = favlib.Net(ckpt, 0)
        ref = EC.reference_run(favlib, net, frames_h, bw_h, fw_h, N, rmode, masks_h=masks, keep=sampled, keep_states=sampled)
        chk = EC.verify_dir(f"{d}/s0/o", "out", ref, N)
        print(mode, "fps", line["fps_end_to_end"], chk)
        assert chk["png_mismatch_frames"] == 0 and chk["png_missing_frames"] == 0 and chk["png_unexpected_files"] == 0, chk
        # sampled frames, teacher-forced against the oracle: the decoded FILE the CLI wrote vs the oracle's frame from the GPU's previous state
        import parity_clip
        oracle.set_threads(parity_clip.effective_cpus())
        f01 = lambda u8: np.transpose(u8, (2, 0, 1)).astype(np.float32) / np.float32(255)
        worst = 0
        for i in sampled:
            k = i % ring
            before, after = ref.states[i]
            st = oracle.Stylizer(layers); st.last = before
            if rmode == "cert": m = masks[k]
            else: m = oracle.consistency(bw_h[k], fw_h[k], frames_h[k] if rmode == "4arg" else None)
            want = st.next(f01(frames_h[k]), bw_h[k], m.astype(np.float32) / np.float32(255))
            assert np.abs(want - after).max() <= 2e-4, (i, float(np.abs(want - after).max()))
            got = EC.decode_png(open(f"{d}/s0/o/out-{i:05d}.png", "rb").read())
            lsb = int(np.abs(got.astype(int) - oracle.to_u8_hwc(want).astype(int)).max())
            worst = max(worst, lsb)
            assert lsb <= 1, (i, lsb)
        print(mode, "sampled frames vs oracle: worst", worst, "LSB over", len(sampled), "frames")
    finally:
        oracle.set_threads(min(16, len(os.sched_getaffinity(0))))
        shutil.rmtree(d, ignore_errors=True)


def test_fav_stylize_vr_matches_oracle(oracle, favlib, tmp_path, golden_dir):
    """`th fast_artistic_video_vr.lua` contract (stylizeVRVideo_deepflow.sh:68-83 patterns): two frames of six faces, files named
    by face id in processing order {6,1,2,5,3,4}; the equirectangular and cube-map PNGs are compared with oracle/vr_oracle.py."""
    from PIL import Image
    import vr_oracle as V
    hp = wp = 64
    model = os.path.join(golden_dir, "tiny_model.t7")
    rng = np.random.default_rng(5)
    kw = dict(overlap_w=24, overlap_h=24, median=3, out_equi_w=96, out_equi_h=48, fill_random=True, seed=9)
    ref = V.VRStylizer(t7.extract_layers(t7.load(model)["model"]), hp, wp, **kw)
    want = []
    for fr in (1, 2):
        for mode, face in enumerate(V.PROC_ORDER):
            i = (fr - 1) * 6 + mode + 1
            f = synth.smooth_frame(hp, wp, 700 + i)
            oracle.write_pnm(str(tmp_path / f"frame_{fr:05d}-{face}.ppm"), f)
            bw = ce = None
            if fr > 1:
                os.makedirs(tmp_path / f"flow_768-{face}", exist_ok=True)
                bw = synth.backward_flow(hp, wp, 800 + i)
                ce = ((rng.random((hp, wp)) > 0.2) * 255).astype(np.uint8)
                oracle.write_flo(str(tmp_path / f"flow_768-{face}" / f"backward_{fr}_{fr-1}.flo"), bw)
                oracle.write_pnm(str(tmp_path / f"flow_768-{face}" / f"reliable_{fr}_{fr-1}.pgm"), ce)
            ref.face(i, np.transpose(f, (2, 0, 1)).astype(np.float32) / np.float32(255), bw,
                     ce.astype(np.float32) / np.float32(255) if ce is not None else None)
        want.append((oracle.to_u8_hwc(ref.equi), oracle.to_u8_hwc(ref.cubemap)))
    _age(tmp_path)
    cmd = [os.path.join(ROOT, "fast-artistic-videos_amd", "host", "th"), "fast_artistic_video_vr.lua",
           "-input_pattern", str(tmp_path / "frame_%05d-%d.ppm"),
           "-flow_pattern", str(tmp_path / "flow_768-%d" / "backward_[%d]_{%d}.flo"),
           "-occlusions_pattern", str(tmp_path / "flow_768-%d" / "reliable_[%d]_{%d}.pgm"),
           "-output_prefix", str(tmp_path / "res" / "out"), "-backend", "cuda", "-use_cudnn", "1", "-gpu", "0",
           "-model_vid", model, "-model_img", "self", "-overlap_pixel_h", "24", "-overlap_pixel_w", "24",
           "-out_equi", "-out_equi_w", "96", "-out_equi_h", "48", "-out_cubemap", "-fill_occlusions", "uniform-random", "-seed", "9"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for fr in (1, 2):
        e = np.asarray(Image.open(str(tmp_path / "res" / f"out-{fr:05d}_equi.png")))
        c = np.asarray(Image.open(str(tmp_path / "res" / f"out-{fr:05d}_cubemap.png")))
        assert e.shape == want[fr - 1][0].shape and c.shape == want[fr - 1][1].shape
        # free-running comparison through 12 network evaluations: a few LSB of drift, high PSNR
        for got, ref8 in ((e, want[fr - 1][0]), (c, want[fr - 1][1])):
            mse = np.mean((got.astype(np.float64) - ref8.astype(np.float64)) ** 2)
            assert np.abs(got.astype(int) - ref8.astype(int)).max() <= 3 and (mse == 0 or 10 * np.log10(255.0 ** 2 / mse) >= 50.0), f"frame {fr}"
    assert not os.path.exists(str(tmp_path / "res" / "out-00003_equi.png"))
    for bad in (["-gpu", "-1"], ["-evaluate"], ["-continue_with", "2"], ["-backward"]):
        rb = subprocess.run(cmd + bad, capture_output=True, text=True)
        assert rb.returncode != 0 and rb.stderr.strip(), bad


def test_fav_stylize_multi_stream_launcher_rccl(oracle, favlib, tmp_path, golden_dir):
    """BASELINE config 4 plumbing on the one GPU of this box: `-streams a,b -gpus 1 -force_dist 1` takes the worker path -- rank 0
    parses the .t7, the packed blob goes through ncclCommInitRank + ncclBroadcast (RCCL), the net is created from the blob --
    and both videos come out byte-identical to two plain single-video runs."""
    import json
    h, w, n = 48, 64, 3
    model = os.path.join(golden_dir, "tiny_model.t7")
    exe = os.path.join(BIN, "fav_stylize")
    for k, name in enumerate(("a", "b")):
        _write_clip(oracle, tmp_path / name, h, w, n, 300 + 50 * k)
    common = ["-flow_pattern", str(tmp_path / "%S" / "flow" / "backward_[%d]_{%d}.flo"),
              "-forward_flow_pattern", str(tmp_path / "%S" / "flow" / "forward_{%d}_[%d].flo"), "-structure", "1",
              "-model_vid", model, "-model_img", "self", "-gpu", "0", "-timing", "1"]
    r = subprocess.run([exe, "-input_pattern", str(tmp_path / "%S" / "frame_%05d.ppm"), "-output_prefix", str(tmp_path / "multi_%S" / "out"),
                        "-streams", "a,b", "-gpus", "1", "-force_dist", "1"] + common, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "via ncclBroadcast from rank 0" in r.stdout and "[rank 0/1 gpu 0]" in r.stdout
    agg = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{"gpus"')]
    assert len(agg) == 1 and agg[0]["frames"] == 2 * n and agg[0]["streams"] == 2 and agg[0]["fps_end_to_end"] > 0
    for name in ("a", "b"):
        single = [a.replace("%S", name) for a in common]
        r1 = subprocess.run([exe, "-input_pattern", str(tmp_path / name / "frame_%05d.ppm"), "-output_prefix", str(tmp_path / f"single_{name}" / "out")] + single,
                            capture_output=True, text=True, timeout=600)
        assert r1.returncode == 0, r1.stderr
        for i in range(1, n + 1):
            a = open(tmp_path / f"multi_{name}" / f"out-{i:05d}.png", "rb").read()
            b = open(tmp_path / f"single_{name}" / f"out-{i:05d}.png", "rb").read()
            assert a == b, (name, i)


def test_fav_stylize_vr_multi_stream_launcher_rccl(oracle, favlib, tmp_path, golden_dir):
    """BASELINE config 5 plumbing: two 360-degree videos through `fav_stylize_vr -streams a,b -gpus 1 -force_dist 1` (worker process,
    RCCL broadcast of the packed weights, videos back to back) give the same PNG bytes as two plain runs."""
    hp = wp = 64
    model = os.path.join(golden_dir, "tiny_model.t7")
    exe = os.path.join(BIN, "fav_stylize_vr")
    rng = np.random.default_rng(11)
    for k, name in enumerate(("a", "b")):
        d = tmp_path / name
        os.makedirs(d, exist_ok=True)
        for fr in (1, 2):
            for face in (6, 1, 2, 5, 3, 4):
                oracle.write_pnm(str(d / f"frame_{fr:05d}-{face}.ppm"), synth.smooth_frame(hp, wp, 900 + 100 * k + 10 * fr + face))
                if fr > 1:
                    os.makedirs(d / f"flow-{face}", exist_ok=True)
                    oracle.write_flo(str(d / f"flow-{face}" / f"backward_{fr}_{fr-1}.flo"), synth.backward_flow(hp, wp, 950 + 100 * k + face))
                    oracle.write_pnm(str(d / f"flow-{face}" / f"reliable_{fr}_{fr-1}.pgm"), ((rng.random((hp, wp)) > 0.2) * 255).astype(np.uint8))
    _age(tmp_path)
    common = ["-flow_pattern", str(tmp_path / "%S" / "flow-%d" / "backward_[%d]_{%d}.flo"), "-occlusions_pattern", str(tmp_path / "%S" / "flow-%d" / "reliable_[%d]_{%d}.pgm"),
              "-gpu", "0", "-model_vid", model, "-model_img", "self", "-overlap_pixel_h", "24", "-overlap_pixel_w", "24",
              "-out_equi", "-out_equi_w", "96", "-out_equi_h", "48", "-fill_occlusions", "uniform-random", "-seed", "4", "-timing", "1"]
    r = subprocess.run([exe, "-input_pattern", str(tmp_path / "%S" / "frame_%05d-%d.ppm"), "-output_prefix", str(tmp_path / "multi_%S" / "out"),
                        "-streams", "a,b", "-gpus", "1", "-force_dist", "1"] + common, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "via ncclBroadcast from rank 0" in r.stdout and '"frames": 4' in r.stdout
    for name in ("a", "b"):
        single = [a.replace("%S", name) for a in common]
        r1 = subprocess.run([exe, "-input_pattern", str(tmp_path / name / "frame_%05d-%d.ppm"), "-output_prefix", str(tmp_path / f"single_{name}" / "out")] + single,
                            capture_output=True, text=True, timeout=600)
        assert r1.returncode == 0, r1.stderr
        for fr in (1, 2):
            assert open(tmp_path / f"multi_{name}" / f"out-{fr:05d}_equi.png", "rb").read() == open(tmp_path / f"single_{name}" / f"out-{fr:05d}_equi.png", "rb").read(), (name, fr)
