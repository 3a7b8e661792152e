"""Register / scratch budgets of the hand-written gfx950 kernels, from the compiler's own resource report (hipcc cross-compiles without
a GPU).  The kernels are written against a fixed register file -- 256 VGPRs at two waves per SIMD, 168 at three -- and a K loop that
spills loses what the layout was chosen for; this pins what DESIGN.md states (no scratch in conv3s2w_kernel, the epilogue-only spills
of the Winograd and nine-position kernels) so that an edit or a compiler change that breaks it shows up in the CPU suite."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fast-artistic-videos_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize",
         "--cuda-device-only", "-c", "-Rpass-analysis=kernel-resource-usage", "-I", os.path.join(ROOT, "include")]
FILES = ["kernels_s2.hip", "kernels_up2.hip", "kernels_wino.hip", "kernels_png.hip"]

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


@pytest.fixture(scope="module")
def usage(tmp_path_factory):
    """{kernel symbol: {VGPRs, ScratchSize, Occupancy}} for the three files, compiled concurrently"""
    d = tmp_path_factory.mktemp("kres")
    procs = [(f, subprocess.Popen([HIPCC] + FLAGS + [os.path.join(CSRC, f), "-o", str(d / (f + ".o"))], stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True))
             for f in FILES]
    out = {}
    for f, p in procs:
        err = p.communicate(timeout=900)[1]
        assert p.returncode == 0, err[-2000:]
        cur = None
        for line in err.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = out.setdefault(m.group(1), {})
            m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
            if m and cur is not None:
                cur[m.group(1).split(" ")[0]] = int(m.group(2))
    shutil.rmtree(d, ignore_errors=True)
    return out


def _one(usage, *parts):
    hits = [k for k in usage if all(p in k for p in parts)]
    assert len(hits) == 1, (parts, hits)
    return usage[hits[0]]


def test_stride2_kernels_do_not_spill(usage):
    # (Lb1E: the instantiation that computes more than 128 filters in groups of 128, round 5)
    for inst, waves in (("ILi2ELi4ELb0E", 2), ("ILi4ELi3ELb0E", 3), ("ILi4ELi2ELb0E", 2), ("ILi4ELi3ELb1E", 3)):
        u = _one(usage, "conv3s2w_kernel", inst)
        assert u["ScratchSize"] == 0, (inst, u)
        assert u["Occupancy"] >= waves and u["VGPRs"] <= (256 if waves == 2 else 168), (inst, u)


def test_nine_position_kernel_fits_three_waves_per_simd(usage):
    for inst in ("ILb0E", "ILb1E"):                # one group of 64 filters (canonical) | any number of groups
        u = _one(usage, "conv3_up2w_kernel", inst)
        assert u["Occupancy"] == 3 and u["VGPRs"] <= 168, u
        assert u["ScratchSize"] <= 160, u          # epilogue (the fourth output of every pixel) + a handful of K-loop temporaries


def test_winograd_kernel_keeps_its_accumulators_in_registers(usage):
    for inst in ("ILi0E", "ILi1E", "ILi2E"):       # plain input, pending InstanceNorm, pending residual join
        u = _one(usage, "conv3_wino_kernel", inst)
        assert u["Occupancy"] == 2 and u["VGPRs"] <= 256, (inst, u)
        assert u["ScratchSize"] <= 96, (inst, u)   # epilogue only (scripts/isa_loops.py: no scratch access inside the K loops)


def test_png_encoder_kernels_keep_everything_in_registers_and_lds(usage):
    """the PNG encoder's three kernels index small per-thread arrays (the chunk header, the trailer) and per-lane token state: none of
    it may end up in scratch memory"""
    for name in ("png_rows_kernelILb1E", "png_rows_kernelILb0E", "png_pack_kernel", "png_finish_kernel"):
        u = _one(usage, name)
        assert u["ScratchSize"] == 0 and u["VGPRs"] <= 96, (name, u)
