"""CPU suite: the PNG format the GPU encoder writes (csrc/kernels_png.hip), through its lane-level restatement oracle/png_model.py:
the files are valid PNGs for two independent decoders (PIL / libpng and a minimal zlib-based reader), for every shape class the
kernels distinguish.  The GPU suite (tests/test_gpu_png.py) then compares the kernels' bytes with the model's."""
import io
import zlib

import numpy as np
import pytest

from fav_amd import synth


def _cases():
    rng = np.random.default_rng(5)
    yield "smooth", synth.smooth_frame(40, 97, 1)
    yield "noise", rng.integers(0, 256, (17, 70, 3), dtype=np.uint8)
    yield "flat", np.full((9, 130, 3), 200, np.uint8)
    yield "one-pixel", rng.integers(0, 256, (1, 1, 3), dtype=np.uint8)
    yield "gradient", np.tile((np.arange(300) % 256).astype(np.uint8)[None, :, None], (5, 1, 3))
    yield "runs-across-steps", np.repeat(rng.integers(0, 256, (6, 9, 3), dtype=np.uint8), 37, axis=1)       # runs of 111 bytes: cut at 64
    yield "high-bytes", rng.integers(144, 256, (4, 33, 3), dtype=np.uint8)                                   # 9-bit literals
    yield "short-runs", np.repeat(rng.integers(0, 256, (3, 50, 3), dtype=np.uint8), 2, axis=1)             # runs of 3 (the shortest match)


@pytest.mark.parametrize("name,img", list(_cases()), ids=[n for n, _ in _cases()])
def test_png_model_files_decode_exactly(oracle, name, img):
    import png_model as P
    from PIL import Image
    data = P.encode(img)
    assert len(data) <= P.capacity(img.shape[1], img.shape[0])
    pil = np.array(Image.open(io.BytesIO(data)).convert("RGB"))
    assert np.array_equal(pil, img)
    assert np.array_equal(P.decode(data), img)          # checks every chunk CRC; zlib checks the Adler-32
    # structure: one IDAT, every row either a fixed block closed by a sync marker or one stored block, final empty fixed block
    idat = data[41:-16]
    assert idat[:2] == b"\x78\x01" and idat[-6:-4] == b"\x03\x00"
    n = 3 * img.shape[1] + 1
    segs = [P.encode_row(P.filter_row(img[y])) for y in range(img.shape[0])]
    assert all(len(sg) <= n + 5 for sg in segs) and idat[2:-6] == b"".join(segs)
    assert all(sg.endswith(b"\x00\x00\xff\xff") or (sg[0] == 0 and len(sg) == n + 5) for sg in segs)
    if name == "noise": assert all(len(sg) == n + 5 for sg in segs)                          # incompressible rows are stored
    if name in ("flat", "gradient"): assert all(len(sg) < n // 8 for sg in segs)
    if name == "smooth":                                                                      # an image-like row takes one of the model codes
        choices = [P.encode_row(P.filter_row(img[y]), want_choice=True)[1] for y in range(img.shape[0])]
        assert all(c < len(P.tables()) - 1 for c in choices) and len(data) < 0.8 * img.size      # (97-pixel rows: the 40-byte code header is 14 % of a row)


def test_png_rows_inflate_with_every_code(oracle, favlib):
    """rows that select different codes (sharp to broad residuals, run-heavy, incompressible) against zlib's own inflater in raw mode;
    every ready-made code and its header is a valid, complete deflate code"""
    import png_model as P
    rng = np.random.default_rng(3)
    seen = set()
    rows = [np.arange(200, dtype=np.uint8), np.zeros(500, np.uint8), np.tile(np.array([7, 9, 250], np.uint8), 90), rng.integers(0, 256, 900).astype(np.uint8)]
    for scale in (0.4, 1, 2, 4, 8, 16, 30, 50):
        rows.append((np.clip(np.rint(rng.laplace(0, scale, 1500)), -128, 127).astype(np.int64) % 256).astype(np.uint8))
    run = np.repeat(rng.integers(0, 256, 40).astype(np.uint8), 3 * 20); rows.append(run)
    for row in rows:
        f = np.concatenate([np.array([1], np.uint8), row])
        seg, choice = P.encode_row(f, want_choice=True)
        seen.add(choice)
        assert zlib.decompressobj(-15).decompress(seg + b"\x03\x00") == f.tobytes()
        assert len(seg) <= len(f) + 5
    T = P.tables()
    assert len(seen) >= 6 and len(T) - 1 in seen and len(T) in seen, seen          # several model codes, the fixed code, a stored row
    for t in T[:-1]:
        assert abs(sum(2.0 ** -int(l) for l in t["len"]) - 1.0) < 1e-12 and t["len"].max() <= 15 and t["hdr_bits"] <= 40 * 32


def test_crc_combine_scheme_matches_zlib_beyond_2_pow_24_bytes(favlib):
    """crc32(A || B) from the parts, with the tables the pack / finish kernels index by the bytes of |B|: three tables ended at 2^24
    bytes (a 3840x2160 frame's IDAT is 25 MB: its CRC was wrong and PIL / libpng rejected the file); four cover every 32-bit length"""
    import os
    a = os.urandom(1000)
    ca = zlib.crc32(a)
    for n in (0, 1, 5, 255, 256, 65535, 65536, (1 << 24) - 1, 1 << 24, (1 << 24) + 5, 50_000_003):
        b = bytes(n)
        assert favlib.lib().fav_png_crc32_combine_host(ca, zlib.crc32(b), n) == zlib.crc32(a + b), n
    # a length above 2^31 without materialising it: crc(A || 0^n) for n = n1 + n2 is the combine of the combines
    z1 = zlib.crc32(bytes(40_000_000))
    got = favlib.lib().fav_png_crc32_combine_host(ca, favlib.lib().fav_png_crc32_combine_host(z1, z1, 40_000_000), 80_000_000)
    assert got == favlib.lib().fav_png_crc32_combine_host(favlib.lib().fav_png_crc32_combine_host(ca, z1, 40_000_000), z1, 40_000_000)
