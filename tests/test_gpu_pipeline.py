"""The reference's LIVE pipeline, unmodified, on top of the drop-ins.

stylizeVideo_deepflow.sh:83-96 starts makeOptFlow_deepflow.sh in the background and the stylizer right next to it: the stylizer polls
for certainty files that are being produced (fast_artistic_video.lua:102, fast_artistic_video/utils.lua:74-80), the flow script calls the
checker twice per frame pair (makeOptFlow_deepflow.sh:59-60) and BUILDS the reference's own checker when none is in place (:10-18) --
and that checker writes a full-size all-255 placeholder before the real mask (consistencyChecker.cpp:151-152,171).

Here the two scripts run byte for byte as the reference ships them (copied at build time into the git-ignored oracle/_ref/pipeline by
`make -C oracle pipeline_ref`; /root/reference does not exist on the GPU box), in a scratch directory laid out like the reference's
root, with
  * `th` = the shim of fast-artistic-videos_amd/host (-> bin/fav_stylize),
  * `ffmpeg` and `run-deepflow.sh` = stubs that hand out a synthetic clip (the flow stub takes its time and writes the .flo in two pieces),
  * the four `read -p` prompts answered on stdin,
once with bin/consistencyChecker dropped in (its resident helper computes masks on the GPU the stylizer is using, at the same time) and
once with NO checker in place, so makeOptFlow_deepflow.sh builds the reference's with the reference's Makefile and the stylizer polls
against the placeholder-writing producer.  Every mask must be the real one and every PNG byte-equal to the in-process run on those masks."""
import os
import shutil
import signal
import stat
import subprocess
import sys
import time

import numpy as np
import pytest

from fav_amd import synth, t7

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fast-artistic-videos_amd")
PIPE = os.path.join(ROOT, "oracle", "_ref", "pipeline")

FFMPEG_STUB = r"""#!/bin/bash
# stand-in for ffmpeg in stylizeVideo_deepflow.sh: `-i <video> [-vf scale=..] <dir>/frame_%05d.ppm` hands out the synthetic frames,
# `-i <dir>/out-%05d.png <name>-stylized.<ext>` records how many stylised frames it was given
last="${@: -1}"
case "$last" in
  *frame_%05d.ppm) cp "$FAV_TEST_SRC"/frame_*.ppm "$(dirname "$last")"/ ;;
  *) in="$2"; ls "$(dirname "$in")"/out-*.png | wc -l > "$last" ;;
esac
"""

DEEPFLOW_STUB = r"""# stand-in for run-deepflow.sh (makeOptFlow_deepflow.sh:3 calls `bash run-deepflow.sh <img1> <img2> <out.flo> <opt_res>`):
# the flow takes a while and the file grows in two pieces, like a real estimator's output
if [ "$#" -ne 4 ]; then echo "run-deepflow stub: expected 4 arguments" >&2; exit 1; fi
src="$FAV_TEST_SRC/$(basename "$3")"
n=$(stat -c %s "$src")
case "$(basename "$3")" in forward_*) sleep 0.05 ;; *) sleep ${FAV_TEST_FLOW_DELAY:-0.15} ;; esac   # (:53 starts the forward flow in the background and :59 reads it as soon as it exists)
head -c $((n / 2)) "$src" > "$3"
sleep 0.05
tail -c +$((n / 2 + 1)) "$src" >> "$3"
"""


def _stop_helper(run_dir):
    import glob
    for lock in glob.glob(os.path.join(run_dir, "fav-cc", "gpu0*.lock")):
        try:
            pid = int(open(lock).read().split()[0]); os.kill(pid, signal.SIGTERM)
        except (ValueError, IndexError, ProcessLookupError, OSError):
            continue
        for _ in range(200):
            try:
                os.kill(pid, 0)
            except ProcessLookupError:
                break
            time.sleep(0.02)


def _wait_group_gone(pgid, timeout):
    t0 = time.time()
    while time.time() - t0 < timeout:
        try:
            os.killpg(pgid, 0)
        except ProcessLookupError:
            return True
        time.sleep(0.1)
    return False


@pytest.mark.skipif(not os.path.exists(os.path.join(PIPE, "stylizeVideo_deepflow.sh")),
                    reason="oracle/_ref/pipeline not populated (make -C oracle pipeline_ref needs /root/reference at build time)")
@pytest.mark.parametrize("checker", ["drop-in", "reference-built-by-the-script"])
def test_unmodified_reference_pipeline(oracle, favlib, cuda, tmp_path, checker):
    import torch
    h, w, n = 192, 320, 7
    # ---- the synthetic clip the stubs hand out
    src = tmp_path / "src"; src.mkdir()
    frames, bws, fws = [], [None], [None]
    for i in range(1, n + 1):
        f = synth.smooth_frame(h, w, 500 + i); frames.append(f)
        oracle.write_pnm(str(src / f"frame_{i:05d}.ppm"), f)
        if i > 1:
            bw = synth.backward_flow(h, w, 600 + i); fw = synth.forward_flow_from_backward(bw, 700 + i)
            bws.append(bw); fws.append(fw)
            oracle.write_flo(str(src / f"backward_{i}_{i-1}.flo"), bw); oracle.write_flo(str(src / f"forward_{i-1}_{i}.flo"), fw)
    ckpt = str(tmp_path / "canonical.t7")
    t7.make_synthetic_checkpoint(ckpt, seed=77)

    # ---- a directory laid out like the reference's root
    root = tmp_path / "refroot"; root.mkdir()
    for s in ("stylizeVideo_deepflow.sh", "makeOptFlow_deepflow.sh"):
        shutil.copy(os.path.join(PIPE, s), root / s)
        assert open(root / s, "rb").read() == open(os.path.join(PIPE, s), "rb").read()
    (root / "run-deepflow.sh").write_text(DEEPFLOW_STUB)
    (root / "fast_artistic_video.lua").write_text("-- placeholder: `th fast_artistic_video.lua` is answered by the th shim\n")
    if checker == "drop-in":
        (root / "consistencyChecker").mkdir()
        os.symlink(os.path.join(PKG, "bin", "consistencyChecker"), root / "consistencyChecker" / "consistencyChecker")
    else:                              # makeOptFlow_deepflow.sh:10-18 finds no binary and runs the reference's own Makefile
        shutil.copytree(os.path.join(PIPE, "consistencyChecker"), root / "consistencyChecker")
        assert not (root / "consistencyChecker" / "consistencyChecker").exists()
    stubs = tmp_path / "stubs"; stubs.mkdir()
    (stubs / "ffmpeg").write_text(FFMPEG_STUB); os.chmod(stubs / "ffmpeg", 0o755)
    run = tmp_path / "run"; run.mkdir(mode=0o700)
    env = dict(os.environ, PATH=f"{stubs}:{os.path.join(PKG, 'host')}:{os.environ['PATH']}", FAV_TEST_SRC=str(src),
               XDG_RUNTIME_DIR=str(run), FAV_CC_IDLE_S="30",
               FAV_TEST_FLOW_DELAY="0.6" if checker == "drop-in" else "0.15")   # (drop-in: keep the stylizer ahead of its producer)
    env.pop("FAV_CC_DAEMON", None)

    # ---- the reference's driver, prompts answered with their defaults: GPU 0, cudnn, original resolution, opt_res 2
    t0 = time.time()
    p = subprocess.Popen(["bash", "stylizeVideo_deepflow.sh", "clip.mp4", ckpt], cwd=str(root), env=env, stdin=subprocess.PIPE,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate("\n\n\n\n", timeout=420)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL); out, err = p.communicate()
        pytest.fail("the pipeline hung:\n" + out[-3000:] + "\n" + err[-3000:])
    finally:
        gone = _wait_group_gone(p.pid, 120)          # the background flow script finishes its last pair after the stylizer is done
        if not gone:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
        _stop_helper(str(run))
    wall = time.time() - t0
    assert p.returncode == 0, (out[-3000:], err[-3000:])
    assert gone, "makeOptFlow_deepflow.sh never finished"
    d = root / "clip"
    flow = d / "flow_default"
    print(f"{checker}: pipeline wall time {wall:.1f} s;", out.count("Waiting for file"), "waits announced")
    assert "Starting optical flow computation as a background task" in out and "Model loaded." in out
    assert out.count("Writing output image to") == n
    assert 'Waiting for file "clip/flow_default/reliable_2_1.pgm"' in out           # the stylizer was ahead of its producer (utils.lua:76)
    assert (root / "clip-stylized.mp4").read_text().split() == [str(n)]             # the script's last step saw every frame
    if checker != "drop-in":
        assert (root / "consistencyChecker" / "consistencyChecker").exists()        # built by the script, with the reference's Makefile

    # ---- every mask the producer left is the real one (never the placeholder), in both directions
    masks = [None, None]
    for i in range(2, n + 1):
        got = oracle.read_pnm(str(flow / f"reliable_{i}_{i-1}.pgm"))
        assert np.array_equal(got, oracle.consistency(bws[i - 1], fws[i - 1], frames[i - 1])), i
        assert 0 < int((got == 255).sum()) < got.size
        assert np.array_equal(oracle.read_pnm(str(flow / f"reliable_{i-1}_{i}.pgm")), oracle.consistency(fws[i - 1], bws[i - 1], frames[i - 2])), i
        masks.append(got)

    # ---- every PNG equals the in-process run on those masks, byte for byte
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    net = favlib.Net(ckpt, 0)
    st = favlib.Stream(net, h, w)
    png_out, png_n = st.png_buffers()
    for i in range(1, n + 1):
        if i == 1:
            st.first_frame(T(frames[0]), want_f32=False)
        else:
            st.next_frame_cert(T(frames[i - 1]), T(bws[i - 1]), T(masks[i]), want_f32=False)
        st.encode_png_into(png_out, png_n)
        want = png_out[: int(png_n.item())].cpu().numpy().tobytes()
        got = open(d / f"out-{i:05d}.png", "rb").read()
        assert got == want, f"frame {i}: the PNG differs from the in-process run ({len(got)} vs {len(want)} bytes)"
    net.check()
    assert not (d / f"out-{n+1:05d}.png").exists()
