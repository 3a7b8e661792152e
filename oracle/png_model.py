"""Lane-level CPU restatement of the GPU PNG encoder (fast-artistic-videos_amd/csrc/kernels_png.hip).

TEST INFRASTRUCTURE ONLY (imported by tests/): it predicts the exact bytes the two kernels write, so the format can be
validated against zlib / PIL on the CPU (no GPU in the build container) and the GPU output compared byte for byte.

The format (A9, image.save of fast_artistic_video.lua:160-170 -> an RGB8 PNG):
  * one IDAT chunk holding one zlib stream (header 78 01);
  * every image row = PNG filter type 1 (Sub) = ONE deflate block whose tokens are literals and distance-3 matches (run-length coding
    of repeated pixels / constant gradients), cut at 64-position boundaries (one wave step).  The block is the cheapest, by exact bit
    count, of: a dynamic-Huffman block (BTYPE 10) with one of twelve ready-made codes for Sub-filtered image rows (csrc/png_tables.cpp:
    length-limited Huffman codes of two-sided geometric literal distributions, header included), a fixed-Huffman block (BTYPE 01), a
    stored block (00 | LEN | ~LEN | the filtered bytes).  Huffman blocks are followed by an EMPTY STORED BLOCK (the Z_SYNC_FLUSH
    marker 00 00 FF FF) that re-aligns the stream to a byte boundary -- so rows are encoded independently (one block of waves each)
    and concatenated at byte granularity;
  * a final empty fixed block (03 00), the Adler-32 of the filtered stream, the chunk CRC-32 -- both combined from per-row parts.
"""
import struct
import zlib

import numpy as np


_TABLES = None


def tables():
    """the encoder's Huffman codes, read from libfav (host-only call: no device needed) -- the model restates the kernels' BIT
    LAYOUT and table choice; that the codes and headers are valid deflate is what zlib / PIL check on its output"""
    global _TABLES
    if _TABLES is None:
        import fav_amd
        _TABLES = fav_amd.png_tables()
    return _TABLES


def length_symbol(length):
    """run of 3..66 -> (symbol 257..276, extra bit count, extra value) (RFC 1951 3.2.5)"""
    assert 3 <= length <= 66
    if length <= 10: return 254 + length, 0, 0
    if length <= 18: return 265 + (length - 11) // 2, 1, (length - 11) % 2
    if length <= 34: return 269 + (length - 19) // 4, 2, (length - 19) % 4
    return 273 + (length - 35) // 8, 3, (length - 35) % 8


def filter_row(raw):
    """PNG filter type 1 (Sub), bpp = 3: the filter-type byte followed by raw[j] - raw[j-3] (mod 256)"""
    raw = np.asarray(raw, np.uint8).ravel()
    prev = np.concatenate([np.zeros(3, np.uint8), raw[:-3]]) if raw.size > 3 else np.zeros_like(raw)
    if raw.size <= 3:
        prev = np.zeros_like(raw)
    return np.concatenate([np.array([1], np.uint8), (raw.astype(np.int32) - prev.astype(np.int32)).astype(np.uint8)])


def tokens(f):
    """the row's tokens as the row kernel forms them: 64 positions per step, runs of f[p] == f[p-3] inside a step; a run of >= 3
    becomes one match (distance 3), everything else literals.  -> list of (symbol, extra bits, extra value, is_match)"""
    n = len(f)
    out = []
    for base in range(0, n, 64):
        m = [(base + l) < n and (base + l) >= 3 and f[base + l] == f[base + l - 3] for l in range(64)]
        l = 0
        while l < 64 and base + l < n:
            if m[l]:
                e = l
                while e < 64 and m[e]: e += 1
                if e - l >= 3:
                    sym, eb, ev = length_symbol(e - l)
                    out.append((sym, eb, ev, True))
                else:
                    for q in range(l, e): out.append((int(f[base + q]), 0, 0, False))
                l = e
            else:
                out.append((int(f[base + l]), 0, 0, False)); l += 1
    return out


def encode_row(f, want_choice=False):
    """one row's deflate segment (bytes): the cheapest -- by exact bit count, first minimum in the order model codes 0.., fixed code,
    stored -- of a dynamic-Huffman block with one of the encoder's model codes, a fixed-Huffman block, a stored block.  Huffman
    blocks end with the end-of-block symbol and an empty stored block (00 00 FF FF) that re-aligns the stream to a byte boundary."""
    n = len(f)
    T = tables()
    toks = tokens(f)
    hist = np.zeros(277, np.int64)
    xbits = nmatch = 0
    for sym, eb, ev, is_m in toks:
        hist[sym] += 1; xbits += eb; nmatch += int(is_m)
    costs = []
    for t in T:                                                    # the last table is the fixed code (no header)
        costs.append(3 + t["hdr_bits"] + int((hist * t["len"]).sum()) + xbits + nmatch * t["dist_len"] + int(t["len"][256]))
    sizes = [((c + 3 + 7) >> 3) + 4 for c in costs] + [n + 5]      # + stored-block header bits, padded to a byte, + LEN / NLEN
    choice = int(np.argmin(sizes))                                 # first minimum
    if choice == len(T):
        seg = b"\x00" + struct.pack("<HH", n, n ^ 0xFFFF) + bytes(bytearray(int(x) for x in f))
        return (seg, choice) if want_choice else seg
    t = T[choice]
    acc = (t["btype"] << 1); bits = 3                              # BFINAL = 0, BTYPE LSB first
    hdr = 0
    for i, wd in enumerate(t["hdr"]): hdr |= int(wd) << (32 * i)
    acc |= (hdr & ((1 << t["hdr_bits"]) - 1)) << bits; bits += t["hdr_bits"]
    for sym, eb, ev, is_m in toks:
        acc |= int(t["code"][sym]) << bits; bits += int(t["len"][sym])
        if is_m:
            acc |= ev << bits; bits += eb
            acc |= t["dist_code"] << bits; bits += t["dist_len"]
    acc |= int(t["code"][256]) << bits; bits += int(t["len"][256])
    assert bits == costs[choice]
    bits += 3                                                     # stored block header: BFINAL 0, BTYPE 00
    bits = (bits + 7) & ~7
    seg = acc.to_bytes(bits // 8, "little") + b"\x00\x00\xff\xff"
    assert len(seg) == sizes[choice]
    return (seg, choice) if want_choice else seg


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
