"""Lane-level CPU restatement of the GPU PNG encoder (fast-artistic-videos_amd/csrc/kernels_png.hip).

TEST INFRASTRUCTURE ONLY (imported by tests/): it predicts the exact bytes the two kernels write, so the format can be
validated against zlib / PIL on the CPU (no GPU in the build container) and the GPU output compared byte for byte.

The format (A9, image.save of fast_artistic_video.lua:160-170 -> an RGB8 PNG):
  * one IDAT chunk holding one zlib stream (header 78 01);
  * every image row = PNG filter type 1 (Sub) = one fixed-Huffman deflate block (BFINAL 0, BTYPE 01) whose tokens are literals and
    distance-3 matches (run-length coding of repeated pixels / constant gradients), cut at 64-position boundaries (one wave step),
    followed by an EMPTY STORED BLOCK (the Z_SYNC_FLUSH marker 00 00 FF FF) that re-aligns the stream to a byte boundary -- or, when
    that would be longer than the row itself, ONE stored block (00 | LEN | ~LEN | the filtered bytes) -- so rows are encoded
    independently (one block of waves each) and concatenated at byte granularity;
  * a final empty fixed block (03 00), the Adler-32 of the filtered stream, the chunk CRC-32 -- both combined from per-row parts.
"""
import struct
import zlib

import numpy as np


def _brev(x, n):
    return int(format(x, "0%db" % n)[::-1], 2)


def literal_code(v):
    """(value to OR in LSB-first, bit count) of a literal byte in the fixed Huffman code (RFC 1951 3.2.6)"""
    if v < 144:
        return _brev(0x30 + v, 8), 8
    return _brev(0x190 + v - 144, 9), 9


def match_code(length):
    """distance-3 match of 3..66 bytes: 7-bit length symbol (257..276) + extra bits + 5-bit distance code 2"""
    assert 3 <= length <= 66
    if length <= 10: sym, eb, ev = 254 + length, 0, 0
    elif length <= 18: sym, eb, ev = 265 + (length - 11) // 2, 1, (length - 11) % 2
    elif length <= 34: sym, eb, ev = 269 + (length - 19) // 4, 2, (length - 19) % 4
    else: sym, eb, ev = 273 + (length - 35) // 8, 3, (length - 35) % 8
    val, nb = _brev(sym - 256, 7), 7
    val |= ev << nb; nb += eb
    val |= _brev(2, 5) << nb; nb += 5
    return val, nb


def filter_row(raw):
    """PNG filter type 1 (Sub), bpp = 3: the filter-type byte followed by raw[j] - raw[j-3] (mod 256)"""
    raw = np.asarray(raw, np.uint8).ravel()
    prev = np.concatenate([np.zeros(3, np.uint8), raw[:-3]]) if raw.size > 3 else np.zeros_like(raw)
    if raw.size <= 3:
        prev = np.zeros_like(raw)
    return np.concatenate([np.array([1], np.uint8), (raw.astype(np.int32) - prev.astype(np.int32)).astype(np.uint8)])


def encode_row(f):
    """one row's deflate segment (bytes) as the row wave writes it: 64 positions per step, runs of f[p] == f[p-3] inside a step"""
    n = len(f)
    bits = 0; acc = 0
    acc |= 2; bits = 3                                            # BFINAL = 0, BTYPE = 01 (LSB first: 0, 1, 0)
    for base in range(0, n, 64):
        m = [(base + l) < n and (base + l) >= 3 and f[base + l] == f[base + l - 3] for l in range(64)]
        l = 0
        while l < 64 and base + l < n:
            if m[l]:
                e = l
                while e < 64 and m[e]: e += 1
                run = e - l
                if run >= 3:
                    val, nb = match_code(run)
                    acc |= val << bits; bits += nb
                else:
                    for q in range(l, e):
                        val, nb = literal_code(int(f[base + q])); acc |= val << bits; bits += nb
                l = e
            else:
                val, nb = literal_code(int(f[base + l])); acc |= val << bits; bits += nb
                l += 1
    bits += 7                                                     # end of block (symbol 256 = 0000000)
    bits += 3                                                     # stored block header: BFINAL 0, BTYPE 00
    bits = (bits + 7) & ~7
    nbytes = bits // 8
    out = bytearray(acc.to_bytes(nbytes, "little"))
    out += b"\x00\x00\xff\xff"                                    # LEN = 0, NLEN = ~0
    if len(out) > n + 5:                                          # the stored form is shorter (noise-like rows): 00 | LEN | ~LEN | bytes
        return b"\x00" + struct.pack("<HH", n, n ^ 0xFFFF) + bytes(bytearray(int(x) for x in f))
    return bytes(out)


def row_stride(width):
    n = 3 * width + 1
    return ((n + 5 + 3) & ~3) + 8


def capacity(width, height):
    return 43 + height * row_stride(width) + 6 + 4 + 12


def ihdr(width, height):
    body = b"IHDR" + struct.pack(">IIBBBBB", width, height, 8, 2, 0, 0, 0)
    return b"\x89PNG\r\n\x1a\n" + struct.pack(">I", 13) + body + struct.pack(">I", zlib.crc32(body))


def encode(rgb_hwc):
    """the whole file, assembled the way the pack kernel does"""
    rgb = np.ascontiguousarray(rgb_hwc, np.uint8)
    h, w, _ = rgb.shape
    rows = [filter_row(rgb[y]) for y in range(h)]
    segs = [encode_row(r) for r in rows]
    n = 3 * w + 1
    # Adler-32 from per-row parts (A_k = sum f, B_k = sum (n - i) f[i])
    s1 = 1; s2 = n * h
    for k, r in enumerate(rows):
        a = int(r.astype(np.uint64).sum()); b = int(((n - np.arange(n, dtype=np.uint64)) * r.astype(np.uint64)).sum())
        s1 += a; s2 += b + n * (h - 1 - k) * a
    adler = ((s2 % 65521) << 16) | (s1 % 65521)
    data = b"\x78\x01" + b"".join(segs) + b"\x03\x00" + struct.pack(">I", adler)
    chunk = b"IDAT" + data
    out = ihdr(w, h) + struct.pack(">I", len(data)) + chunk + struct.pack(">I", zlib.crc32(chunk))
    out += struct.pack(">I", 0) + b"IEND" + struct.pack(">I", zlib.crc32(b"IEND"))
    return out


def decode(png_bytes):
    """minimal PNG reader (RGB8, non-interlaced, any filter) -> [H][W][3] u8; independent of PIL"""
    assert png_bytes[:8] == b"\x89PNG\r\n\x1a\n"
    p = 8; idat = b""; w = h = None
    while p < len(png_bytes):
        ln, = struct.unpack(">I", png_bytes[p:p + 4]); typ = png_bytes[p + 4:p + 8]; body = png_bytes[p + 8:p + 8 + ln]
        crc, = struct.unpack(">I", png_bytes[p + 8 + ln:p + 12 + ln])
        assert zlib.crc32(typ + body) == crc, "chunk CRC"
        if typ == b"IHDR":
            w, h, bd, ct, cm, fm, im = struct.unpack(">IIBBBBB", body); assert (bd, ct, cm, fm, im) == (8, 2, 0, 0, 0)
        elif typ == b"IDAT": idat += body
        elif typ == b"IEND": break
        p += 12 + ln
    raw = zlib.decompress(idat)
    n = 3 * w
    out = np.zeros((h, n), np.uint8)
    prev = np.zeros(n, np.int32)
    for y in range(h):
        ft = raw[y * (n + 1)]; line = np.frombuffer(raw, np.uint8, n, y * (n + 1) + 1).astype(np.int32)
        cur = np.zeros(n, np.int32)
        for i in range(n):
            a = cur[i - 3] if i >= 3 else 0; b = prev[i]; c = prev[i - 3] if i >= 3 else 0
            if ft == 0: pr = 0
            elif ft == 1: pr = a
            elif ft == 2: pr = b
            elif ft == 3: pr = (a + b) // 2
            else:
                pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            cur[i] = (line[i] + pr) & 255
        out[y] = cur; prev = cur
    return out.reshape(h, w, 3)
