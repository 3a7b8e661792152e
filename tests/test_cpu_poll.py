"""fav_stylize's loader against a producer that is STILL WRITING (no GPU needed: `-load_probe <i>` runs the frame loop's own loader for
frame i -- load_frame_inputs, host/fav_stylize.cpp -- and prints a hash of what it read).

The reference's pipeline (stylizeVideo_deepflow.sh:83-96) runs makeOptFlow_deepflow.sh in the background next to the stylizer; the
checker that script calls writes a FULL-SIZE ALL-255 placeholder first (consistencyChecker/consistencyChecker.cpp:151-152) and
re-opens the file "wb" for the real mask ~0.1 s later (:171): it is empty, then short, then complete.  The reference's consumer
survives that because it sleeps a second after the file shows up (fast_artistic_video/utils.lua:74-80).  host/fav_poll.h gives the
same guarantee through the file's modification time; these tests hold it to that."""
import json
import os
import subprocess
import threading
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "fast-artistic-videos_amd", "bin", "fav_stylize")
REF_CHECKER = os.path.join(ROOT, "oracle", "_ref", "consistencyChecker")

pytestmark = pytest.mark.skipif(not os.path.exists(EXE), reason="bin/fav_stylize not built")

H, W = 90, 160


def _fnv(b):
    h = 1469598103934665603
    for x in np.frombuffer(b, np.uint8).tolist():
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


def _old(path, seconds=30.0):
    t = time.time() - seconds
    os.utime(path, (t, t))


def _ppm(path, rgb):
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (rgb.shape[1], rgb.shape[0])); f.write(rgb.tobytes())


def _flo_bytes(uv):
    return np.float32(202021.25).tobytes() + np.array([uv.shape[1], uv.shape[0]], "<i4").tobytes() + uv.astype("<f4").tobytes()


def _clip(d, h=H, w=W, seed=0):
    """two finished frames + a finished backward flow (all old); the certainty / forward flow are the test's business"""
    rng = np.random.default_rng(seed)
    os.makedirs(d / "flow", exist_ok=True)
    for i in (1, 2):
        _ppm(d / f"frame_{i:05d}.ppm", rng.integers(0, 256, (h, w, 3), dtype=np.uint8)); _old(d / f"frame_{i:05d}.ppm")
    bw = rng.standard_normal((h, w, 2)).astype(np.float32)
    (d / "flow" / "backward_2_1.flo").write_bytes(_flo_bytes(bw)); _old(d / "flow" / "backward_2_1.flo")
    return bw


def _probe(d, *extra, timeout=60):
    """frame 2's inputs through the reference's certainty path (fast_artistic_video.lua:100-103)"""
    cmd = [EXE, "-input_pattern", str(d / "frame_%05d.ppm"), "-flow_pattern", str(d / "flow" / "backward_[%d]_{%d}.flo"),
           "-occlusions_pattern", str(d / "flow" / "reliable_[%d]_{%d}.pgm"), "-load_probe", "2", *extra]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    return r, time.time() - t0


def _json(r):
    assert r.returncode == 0, (r.stdout, r.stderr)
    return json.loads(r.stdout.strip().splitlines()[-1])


def _real_mask(seed=5):
    m = (np.random.default_rng(seed).random((H, W)) < 0.7).astype(np.uint8) * 255
    assert 0 < int((m == 255).sum()) < m.size
    return m


def _produce_like_the_reference_checker(path, mask, placeholder_gap=0.15, torn=True):
    """consistencyChecker.cpp:151-152 (all-255 file of the final size), the computation, then :171 -- fopen "wb" truncates, the payload
    follows in pieces"""
    hdr = b"P5\n%d %d\n255\n" % (mask.shape[1], mask.shape[0])
    with open(path, "wb") as f:
        f.write(hdr + b"\xff" * mask.size)
    time.sleep(placeholder_gap)
    with open(path, "wb") as f:                    # truncated to nothing here
        if torn:
            f.flush(); time.sleep(0.03)
            f.write(hdr); f.flush(); time.sleep(0.03)
            f.write(mask.tobytes()[: mask.size // 2]); f.flush(); time.sleep(0.03)
            f.write(mask.tobytes()[mask.size // 2:])
        else:
            f.write(hdr + mask.tobytes())


def test_finished_inputs_cost_nothing(tmp_path):
    _clip(tmp_path)
    m = _real_mask()
    p = tmp_path / "flow" / "reliable_2_1.pgm"
    p.write_bytes(b"P5\n%d %d\n255\n" % (W, H) + m.tobytes()); _old(p)
    r, dt = _probe(tmp_path)
    j = _json(r)
    assert j["ok"] and (j["W"], j["H"]) == (W, H) and j["cert"] == _fnv(m.tobytes()) and j["cert_255"] == int((m == 255).sum())
    assert j["seconds"] < 0.2, j                   # files older than the settle time are taken at the first look
    assert "Waiting for file" not in r.stdout


@pytest.mark.parametrize("start", ["before-the-file-exists", "on-the-placeholder"])
def test_never_hands_over_the_reference_checkers_placeholder(tmp_path, start):
    _clip(tmp_path)
    m = _real_mask()
    p = tmp_path / "flow" / "reliable_2_1.pgm"
    delay = 0.4 if start == "before-the-file-exists" else 0.0
    th = threading.Thread(target=lambda: (time.sleep(delay), _produce_like_the_reference_checker(str(p), m)))
    if start == "on-the-placeholder":
        th.start(); time.sleep(0.05)               # the consumer's first look finds the full-size all-255 file
        assert p.stat().st_size == len(b"P5\n%d %d\n255\n" % (W, H)) + m.size
    else:
        th.start()
    r, dt = _probe(tmp_path)
    th.join()
    j = _json(r)
    assert j["cert_255"] != m.size, "the all-255 placeholder was taken for the mask"
    assert j["cert"] == _fnv(m.tobytes()) and j["cert_255"] == int((m == 255).sum())
    assert j["seconds"] >= 1.0                     # the default settle time: what utils.lua:79's `sleep 1` buys
    if start == "before-the-file-exists":
        assert 'Waiting for file "%s"' % p in r.stdout                         # utils.lua:76's line


def test_settle_time_is_a_flag_and_zero_restores_take_it_when_it_is_there(tmp_path):
    _clip(tmp_path)
    m = _real_mask()
    p = tmp_path / "flow" / "reliable_2_1.pgm"
    p.write_bytes(b"P5\n%d %d\n255\n" % (W, H) + m.tobytes())          # complete, but written just now
    r, _ = _probe(tmp_path, "-poll_settle", "0.3")
    j = _json(r); assert 0.2 <= j["seconds"] < 1.0 and j["cert"] == _fnv(m.tobytes())
    os.utime(p)
    r, _ = _probe(tmp_path, "-poll_settle", "0")
    j = _json(r); assert j["seconds"] < 0.2 and j["cert"] == _fnv(m.tobytes())


def test_a_short_file_under_a_stalled_producer_is_polled_again_not_an_error(tmp_path):
    """a producer that stalls mid-file for longer than the settle time (a loaded box): the payload is shorter than the header promises
    when the consumer reads it -- polled again, and the complete file is what gets handed over"""
    bw = _clip(tmp_path)
    m = _real_mask()
    p = tmp_path / "flow" / "reliable_2_1.pgm"
    fl = tmp_path / "flow" / "backward_2_1.flo"
    flo = _flo_bytes(bw)

    def producer():
        with open(p, "wb") as f, open(fl, "wb") as g:
            f.write(b"P5\n%d %d\n255\n" % (W, H) + m.tobytes()[:1000]); f.flush()
            g.write(flo[: len(flo) // 3]); g.flush()
            time.sleep(1.2)                                              # 4 x the settle time below
            f.write(m.tobytes()[1000:]); g.write(flo[len(flo) // 3:])
    th = threading.Thread(target=producer); th.start()
    time.sleep(0.05)
    r, _ = _probe(tmp_path, "-poll_settle", "0.3")
    th.join()
    j = _json(r)
    assert j["cert"] == _fnv(m.tobytes()) and j["bw"] == _fnv(bw.tobytes()) and j["seconds"] >= 1.2


def test_a_malformed_file_nobody_is_writing_is_still_an_error(tmp_path):
    _clip(tmp_path)
    p = tmp_path / "flow" / "reliable_2_1.pgm"
    p.write_bytes(b"P5\n%d %d\n255\n" % (W, H) + b"\x00" * 100); _old(p)
    r, dt = _probe(tmp_path)
    assert r.returncode != 0 and "truncated" in r.stderr and dt < 5.0, (r.stderr, dt)


def test_poll_timeout_bounds_the_wait(tmp_path):
    _clip(tmp_path)
    r, dt = _probe(tmp_path, "-poll_timeout", "0.5")
    assert r.returncode != 0 and "timed out waiting for" in r.stderr and "reliable_2_1.pgm" in r.stderr and dt < 5.0
    assert "Waiting for file" in r.stdout


@pytest.mark.skipif(not os.path.exists(REF_CHECKER), reason="oracle/_ref/consistencyChecker (the reference's own binary) not built")
@pytest.mark.parametrize("args", [3, 4])
def test_the_reference_binary_itself_as_the_producer(tmp_path, args):
    """the REAL producer: the reference's consistencyChecker (built from its own sources by oracle/Makefile) writing reliable_2_1.pgm at
    1280x720 while the loader waits for it -- the mask handed over must be the file the checker leaves behind, never its placeholder"""
    h, w = 720, 1280
    rng = np.random.default_rng(9)
    os.makedirs(tmp_path / "flow")
    frame = np.clip(rng.normal(128, 40, (h, w, 3)), 0, 255).astype(np.uint8)
    for i in (1, 2):
        _ppm(tmp_path / f"frame_{i:05d}.ppm", frame); _old(tmp_path / f"frame_{i:05d}.ppm")
    bw = (rng.standard_normal((h, w, 2)) * 0.4).astype(np.float32); fw = (-bw + rng.standard_normal((h, w, 2)) * 0.2).astype(np.float32)
    (tmp_path / "flow" / "backward_2_1.flo").write_bytes(_flo_bytes(bw)); _old(tmp_path / "flow" / "backward_2_1.flo")
    (tmp_path / "flow" / "forward_1_2.flo").write_bytes(_flo_bytes(fw))
    out = tmp_path / "flow" / "reliable_2_1.pgm"
    cmd = [REF_CHECKER, str(tmp_path / "flow" / "backward_2_1.flo"), str(tmp_path / "flow" / "forward_1_2.flo"), str(out)]
    if args == 4:
        cmd.append(str(tmp_path / "frame_00002.ppm"))
    th = threading.Thread(target=lambda: (time.sleep(0.3), subprocess.run(cmd, check=True, capture_output=True)))
    th.start()
    r, _ = _probe(tmp_path)
    th.join()
    j = _json(r)
    payload = out.read_bytes()[len(b"P5\n1280 720\n255\n"):]
    n255 = int((np.frombuffer(payload, np.uint8) == 255).sum())
    assert 0 < n255 < h * w, "the synthetic flows should give a mixed mask"
    assert j["cert_255"] == n255 and j["cert"] == _fnv(payload)
