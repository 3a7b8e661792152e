"""Where a single `consistencyChecker` call's time goes: one 1280x720 flow pair on /dev/shm,
  * FAV_CC_DAEMON=0 FAV_CC_TIMING=1 : the call computes in its own process; the program prints its own laps (file reads, hipInit + device
    enumeration, context + buffers, copies, code-object load + kernels, write); `ld.so` statistics give the time before main()
  * default                         : the first call starts the resident helper, the following calls go through it
and the reference's CPU binary beside them when oracle/_ref holds it.  usage: python scripts/checker_breakdown.py [3|4]"""
import os, shutil, signal, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-artistic-videos_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from fav_amd import synth
import oracle as O

H, W = 720, 1280
exe = os.path.join(ROOT, "fast-artistic-videos_amd", "bin", "consistencyChecker")
ref = os.path.join(ROOT, "oracle", "_ref", "consistencyChecker")
d = tempfile.mkdtemp(prefix="fav_ccb_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    bw = synth.backward_flow(H, W, 1); fw = synth.forward_flow_from_backward(bw, 2); img = synth.smooth_frame(H, W, 3)
    a, b, i, o = (os.path.join(d, n) for n in ("bw.flo", "fw.flo", "img.ppm", "out.pgm"))
    O.write_flo(a, bw); O.write_flo(b, fw); O.write_pnm(i, img)
    for nargs in ([3, 4] if len(sys.argv) < 2 else [int(sys.argv[1])]):
        args = [a, b, o] + ([i] if nargs == 4 else [])
        print("==== %d-argument form, 1280x720" % nargs)
        if os.path.exists(ref):
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); subprocess.check_call([ref] + args, stdout=subprocess.DEVNULL); ts.append(time.perf_counter() - t0)
            print("reference CPU binary: %.1f ms per call (min of 3)" % (min(ts) * 1e3))
            want = open(o, "rb").read()
        else:
            want = None
        print("-- a call that computes in its own process (FAV_CC_DAEMON=0), second of two runs:")
        for k in range(2):
            t0 = time.perf_counter()
            r = subprocess.run([exe] + args, capture_output=True, text=True, env=dict(os.environ, FAV_CC_DAEMON="0", FAV_CC_TIMING="1", LD_DEBUG="statistics"))
            dt = time.perf_counter() - t0
        for line in r.stderr.splitlines():
            if "timing" in line or "total startup time" in line or "time needed for relocation" in line or "time needed to load objects" in line:
                print("   " + line.strip())
        print("   wall time of the process: %.1f ms%s" % (dt * 1e3, "" if want is None else "   bytes equal to the reference's: %s" % (open(o, "rb").read() == want)))
        run = os.path.join(d, "run%d" % nargs); os.mkdir(run, 0o700)
        env = dict(os.environ, XDG_RUNTIME_DIR=run, FAV_CC_IDLE_S="30")
        ts = []
        for k in range(6):
            t0 = time.perf_counter(); subprocess.check_call([exe] + args, stdout=subprocess.DEVNULL, env=env); ts.append(time.perf_counter() - t0)
        print("-- through the resident helper: first call (starts it) %.1f ms, then %s ms%s" %
              (ts[0] * 1e3, " ".join("%.1f" % (t * 1e3) for t in ts[1:]), "" if want is None else "   bytes equal: %s" % (open(o, "rb").read() == want)))
        try:
            import glob
            for lk in glob.glob(os.path.join(run, "fav-cc", "gpu0*.lock")):
                os.kill(int(open(lk).read().split()[0]), signal.SIGTERM)
        except Exception as e:
            print("   (could not stop the helper: %r)" % (e,))
finally:
    shutil.rmtree(d, ignore_errors=True)
