// kernels_png.hip -- A9 on the GPU: image.save("<prefix>-%05d.png", img) (fast_artistic_video.lua:160-170) as two kernels that
// turn the stylised frame into the bytes of an RGB8 PNG file, so that the host only write()s them.
//
// Why: zlib on the host costs 25 ms per 1280x720 frame and core (round 2: Sub filter + Z_RLE, level 1); one GPU at 500 frames/s
// needs 12.5 cores for it and eight GPUs behind a 16-CPU quota cannot have them.  Everything else of the
// per-frame path already runs on the device.
//
// Format (restated lane by lane in oracle/png_model.py, which the CPU suite checks against zlib / PIL and the GPU suite against these
// kernels byte for byte):
//   * one IDAT chunk = one zlib stream (78 01); every image row is PNG filter type 1 (Sub) and ONE deflate block whose tokens are
//     literals and distance-3 matches (repeated pixels and constant gradients become runs after the Sub filter).  The block is the
//     smallest, by exact bit count from the row's token histogram, of: a dynamic-Huffman block with one of twelve ready-made codes for
//     Sub-filtered image rows (png_tables.cpp: the code AND its 35-60 byte header are built once on the host), a fixed-Huffman block,
//     a stored block (5 + n bytes).  Huffman blocks end with an empty stored block (00 00 FF FF, zlib's Z_SYNC_FLUSH marker) that
//     re-aligns the stream to a byte boundary: rows are independent byte strings either way;
//   * png_rows_kernel: four waves per row.  A step = 64 positions: the match predicate f[p] == f[p-3] becomes a 64-bit ballot, run starts
//     and ends come from count-leading / trailing-zero on that word (runs are cut at the step boundary: <= 64 bytes), so the steps
//     are independent up to their bit offset: pass 1 writes a 16-bit token descriptor per position and the token histogram, the
//     block picks the code, pass 2 takes every step's bit count under that code (popcounts of five ballots of the token lengths),
//     one scan over the steps gives the offsets, pass 3 forms the tokens and ORs them into the row's LDS bit buffer at offset +
//     prefix (again from ballots: no cross-lane data movement); Adler-32 parts (sum f, sum (n-i) f[i]);
//   * png_pack_kernel: one block per row.  Prefix sum of the row sizes, the row copied to its final byte offset (dword stores from an
//     LDS copy through a funnel shift, bytes at the two ragged ends), its CRC-32 (16-byte pieces, bitwise) raised to its position:
//     CRC-32 is linear, crc(A || B) = crc(A) * x^(8 |B|) + crc(B) mod P, with x^(8 n) from three 256-entry tables;
//   * png_finish_kernel (one block): XOR of the rows' parts + chunk prelude + trailer, Adler-32 from the rows' parts, header, length,
//     IEND, file size.  (The first version finished in the last block of the pack kernel behind a device-scope fence per block: 88 us,
//     the fences write back every XCD's L2; the kernel boundary costs 3 us.)
//   The output pointer may be device memory or host-mapped pinned memory (fav_stylize passes the latter: the packed bytes cross PCIe
//   once, the host never touches them before write()).
// HBM-bound by construction: 2.8 MB (u8) or 11 MB (planar float, quantisation fused: clamp, x255, truncate as quantize_kernel) in,
// <= 3.1 MB staged + packed out.
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

#include "fav_internal.h"

namespace fav {
namespace {

constexpr uint32_t CRC_POLY = 0xEDB88320u;      // CRC-32 (ISO-HDLC, zlib / PNG), reflected

// a(x) * b(x) mod P in the reflected representation (bit 31 = x^0): zlib's multmodp without its data-dependent exit, so that the
// lanes of a wave do not diverge (32 x (select, xor, shift, select-xor))
__device__ __host__ inline uint32_t crc_mulmod(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
#pragma unroll
    for (int i = 31; i >= 0; --i) {
        p ^= b & (0u - ((a >> i) & 1u));
        b = (b >> 1) ^ (CRC_POLY & (0u - (b & 1u)));
    }
    return p;
}

// x^(8 n) mod P for any 32-bit n from four 256-entry tables indexed by the bytes of n (host-computed once per device, 4 KB):
// x^(8 n) = T0[n & 255] * T1[(n >> 8) & 255] * T2[(n >> 16) & 255] * T3[n >> 24]
// (three tables stopped at n < 2^24 bytes: a 3840x2160 frame's IDAT is 25 MB -- its CRC was wrong; the capacity guard in
//  launch_png_encode keeps n below 2^32)
struct CrcTables { const uint32_t* t; };        // t[256 k .. 256 k + 255] = Tk
__host__ __device__ inline uint32_t crc_xpow8_tab(const uint32_t* t, unsigned n)
{
    uint32_t p = t[n & 255u];
    if (n >> 8) p = crc_mulmod(p, t[256 + ((n >> 8) & 255u)]);
    if (n >> 16) p = crc_mulmod(p, t[512 + ((n >> 16) & 255u)]);
    if (n >> 24) p = crc_mulmod(p, t[768 + (n >> 24)]);
    return p;
}
__device__ inline uint32_t crc_xpow8(const CrcTables& tb, unsigned n) { return crc_xpow8_tab(tb.t, n); }

// standard CRC-32 (init ~0, final ~) of a short byte string in LDS / registers, bitwise
__device__ inline uint32_t crc_bytes(const uint8_t* s, int n)
{
    uint32_t c = 0xFFFFFFFFu;
    for (int i = 0; i < n; ++i) {
        c ^= s[i];
#pragma unroll
        for (int b = 0; b < 8; ++b) c = (c >> 1) ^ (CRC_POLY & (0u - (c & 1u)));
    }
    return ~c;
}

// row geometry shared by the kernels and the host
__host__ __device__ inline int png_row_stride(int W)
{
    const int n = 3 * W + 1;
    return ((n + 5 + 3) & ~3) + 8;      // a row never takes more than its STORED form (5 + n bytes): see the kernel; + 8 for the 2-word OR
}

constexpr int PNG_ROW_WAVES = 8;                // waves per row block; wave w takes the 64-position steps w, w + 8, ... (four until round 6)
constexpr int PNG_HIST = 288;                   // histogram bins per wave: 277 symbols, [280] extra bits of the matches, [281] matches

// ---------------------------------------------------------------------------------------------------------------------------------
// kernel 1: one block of four waves per image row -> stage[row * stride ...], sizes[row], adler[row] = (sum f, sum (n - i) f[i]) mod 65521
// A step = 64 consecutive positions = one wave-wide ballot; runs never cross a step, so the steps are independent up to their bit
// offset.  Pass 1 forms every position's token (symbol, run length) into a 16-bit descriptor and the row's token histogram; from the
// histogram the block takes the EXACT size of the row under each of the ready-made codes (png_tables.cpp), the fixed code and the
// stored form, and picks the smallest; pass 2 sums the token lengths of every step under the chosen code, a scan over the steps
// gives the offsets, pass 3 ORs the tokens into the bit buffer.
template <bool FROM_F32>
__global__ __launch_bounds__(64 * PNG_ROW_WAVES) void png_rows_kernel(const uint8_t* rgb_hwc, const float* rgb_planar, int W, int H, uint8_t* stage,
                                                                      int stride, uint32_t* sizes, uint2* adler, const PngTable* tabs)
{
    extern __shared__ uint32_t lds[];
    __shared__ unsigned long long red_a[PNG_ROW_WAVES], red_b[PNG_ROW_WAVES];
    __shared__ uint32_t hist[PNG_ROW_WAVES][PNG_HIST];
    __shared__ uint32_t tab[PNG_HIST];
    __shared__ uint32_t pick[20];               // [0..12] row size under code k, [13] stored, [16] choice
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = blockIdx.x;
    constexpr int NT = 64 * PNG_ROW_WAVES;
    const int nraw = 3 * W, n = nraw + 1;
    const int nsteps = (n + 63) >> 6;
    const int raw_words = (nraw + 3 + 3) / 4 + 1, f_words = (n + 3) / 4 + 1, desc_words = (n + 1) / 2 + 1, step_words = nsteps + 1, out_words = stride / 4;
    uint32_t* raww = lds;
    uint8_t* f = reinterpret_cast<uint8_t*>(lds + raw_words);
    uint16_t* desc = reinterpret_cast<uint16_t*>(lds + raw_words + f_words);      // per position: 0x8000 | (run - 3) << 9 | symbol, or 0 (no token)
    uint32_t* step_bits = lds + raw_words + f_words + desc_words;                // [nsteps] bit count of a step, then its exclusive prefix
    uint32_t* out = step_bits + step_words;
    const uint8_t* raw;
    if (FROM_F32) {
        // image.save: clamp to [0,1], x255, truncate (quantize_kernel); planar float RGB [3][H][W] -> interleaved bytes
        uint8_t* r8 = reinterpret_cast<uint8_t*>(raww);
        const size_t plane = (size_t)H * W;
        const float* src = rgb_planar + (size_t)row * W;
        for (int x = tid; x < W; x += NT) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float v = src[c * plane + x];
                v = fminf(fmaxf(v, 0.f), 1.f);
                r8[3 * x + c] = (uint8_t)(v * 255.f);
            }
        }
        raw = r8;
    } else {
        // whole dwords around the row, aligned on the ADDRESS (a caller's buffer need not be dword-aligned); the dword that holds the
        // image's last byte is assembled from bytes so that nothing past the caller's H*W*3 bytes is ever read
        const size_t base = (size_t)row * nraw;
        const uintptr_t addr = reinterpret_cast<uintptr_t>(rgb_hwc) + base;
        const unsigned lead = (size_t)(addr & 3u) <= base ? (unsigned)(addr & 3u) : 0u;
        const uint8_t* a0 = rgb_hwc + base - lead;                  // (lead bytes of the previous row; 0 in row 0 even if the buffer is unaligned)
        const int nw = (int)((lead + nraw + 3) / 4);
        const bool ragged = row == H - 1 && ((lead + nraw) & 3);
        if (((reinterpret_cast<uintptr_t>(a0)) & 3u) == 0) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(a0);
            for (int i = tid; i < nw - (ragged ? 1 : 0); i += NT) raww[i] = src[i];
            if (ragged && tid == 0) {
                uint32_t w = 0;
                for (int b = 0; b < (int)((lead + nraw) & 3); ++b) w |= (uint32_t)a0[4 * (nw - 1) + b] << (8 * b);
                raww[nw - 1] = w;
            }
        } else {                                                    // row 0 of an unaligned buffer: bytes
            uint8_t* r8 = reinterpret_cast<uint8_t*>(raww);
            for (int i = tid; i < (int)lead + nraw; i += NT) r8[i] = a0[i];
        }
        raw = reinterpret_cast<const uint8_t*>(raww) + lead;
    }
    for (int i = tid; i < out_words; i += NT) out[i] = 0u;
    for (int i = tid; i < PNG_ROW_WAVES * PNG_HIST; i += NT) (&hist[0][0])[i] = 0u;
    __syncthreads();
    // Sub filter + Adler-32 parts
    unsigned long long sa = 0, sb = 0;
    if (tid == 0) { f[0] = 1; sa = 1; sb = (unsigned long long)n; }
    for (int p = 1 + tid; p < n; p += NT) {
        const int j = p - 1;
        const uint32_t v = (uint32_t)(raw[j] - (j >= 3 ? raw[j - 3] : 0)) & 255u;
        f[p] = (uint8_t)v;
        sa += v; sb += (unsigned long long)(n - p) * v;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { sa += __shfl_xor(sa, d); sb += __shfl_xor(sb, d); }
    if (lane == 0) { red_a[wave] = sa; red_b[wave] = sb; }
    __syncthreads();
    // pass 1: tokens -> descriptors + histogram
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int sidx = wave; sidx < nsteps; sidx += PNG_ROW_WAVES) {
        const int p = (sidx << 6) + lane;
        const bool valid = p < n;
        const uint32_t v = valid ? f[p] : 0u;
        const bool m = valid && p >= 3 && v == f[p - 3];
        const unsigned long long mask = __ballot(m);
        if (!valid) continue;                                       // (after the ballot: every lane takes part in it)
        uint32_t d = 0x8000u | v;                                   // a literal
        if (m) {
            const unsigned long long below = ~mask & lt;
            const int s = below ? 64 - __clzll((long long)below) : 0;
            const unsigned long long above = (~mask >> lane) >> 1;
            const int e = above ? lane + 1 + (__ffsll((long long)above) - 1) : 64;
            const int L = e - s;
            if (L >= 3) {
                if (lane == s) {                                   // the run's first position carries the match, the others nothing
                    const int sym = L <= 10 ? 254 + L : L <= 18 ? 265 + ((L - 11) >> 1) : L <= 34 ? 269 + ((L - 19) >> 2) : 273 + ((L - 35) >> 3);
                    d = 0x8000u | ((uint32_t)(L - 3) << 9) | (uint32_t)sym;
                    atomicAdd(&hist[wave][280], (uint32_t)(sym < 265 ? 0 : (sym - 261) >> 2));
                    atomicAdd(&hist[wave][281], 1u);
                } else d = 0u;
            }
        }
        desc[p] = (uint16_t)d;
        if (d) atomicAdd(&hist[wave][d & 511u], 1u);
    }
    __syncthreads();
    for (int i = tid; i < PNG_HIST; i += NT) {
        uint32_t h = hist[0][i];
#pragma unroll
        for (int w = 1; w < PNG_ROW_WAVES; ++w) h += hist[w][i];
        hist[0][i] = h;
    }
    __syncthreads();
    // the row's exact size under every code (wave w: codes w, w + 8; code 12 is the fixed code), and stored.
    // All code lengths a lane needs (5 symbols x up to 4 codes) are requested in one batch: one memory latency, not one per code.
    {
        constexpr int KPW = (PNG_NTABLES + PNG_ROW_WAVES) / PNG_ROW_WAVES;      // codes per wave: 2
        uint32_t len[KPW][5], hs[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) { const int sy = lane + 64 * i; hs[i] = sy < PNG_NSYM ? hist[0][sy] : 0u; }
#pragma unroll
        for (int q = 0; q < KPW; ++q) {
            const int k = min(wave + PNG_ROW_WAVES * q, PNG_NTABLES);
#pragma unroll
            for (int i = 0; i < 5; ++i) { const int sy = min(lane + 64 * i, PNG_NSYM - 1); len[q][i] = tabs[k].sym[sy] >> 16; }
        }
#pragma unroll
        for (int q = 0; q < KPW; ++q) {
            const int k = wave + PNG_ROW_WAVES * q;
            uint32_t c = 0;
#pragma unroll
            for (int i = 0; i < 5; ++i) c += hs[i] * len[q][i];
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
            if (lane == 0 && k <= PNG_NTABLES) {
                const PngTable& T = tabs[k];
                const uint32_t bits = 3u + T.hdr_bits + c + hist[0][280] + hist[0][281] * T.dist_len + (T.sym[256] >> 16);
                pick[k] = ((bits + 3u + 7u) >> 3) + 4u;             // + the stored-block header of the sync marker, to a byte, + 00 00 FF FF
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        pick[PNG_NTABLES + 1] = (uint32_t)(n + 5);                  // stored: 00 | LEN | ~LEN | n bytes
        int best = 0;
        for (int k = 1; k <= PNG_NTABLES + 1; ++k) if (pick[k] < pick[best]) best = k;      // the first minimum
        pick[16] = (uint32_t)best;
    }
    __syncthreads();
    const int choice = (int)pick[16];
    if (choice == PNG_NTABLES + 1) {
        // one stored block: byte-aligned at both ends, no sync marker needed
        uint8_t* o8 = reinterpret_cast<uint8_t*>(out);
        if (tid == 0) {
            o8[1] = (uint8_t)n; o8[2] = (uint8_t)(n >> 8); o8[3] = (uint8_t)~n; o8[4] = (uint8_t)(~n >> 8);
            sizes[row] = (uint32_t)(n + 5);
            unsigned long long ta = 0, tb2 = 0;
            for (int w = 0; w < PNG_ROW_WAVES; ++w) { ta += red_a[w]; tb2 += red_b[w]; }
            adler[row] = make_uint2((uint32_t)(ta % 65521ull), (uint32_t)(tb2 % 65521ull));
        }
        for (int i = tid; i < n; i += NT) o8[5 + i] = f[i];
        __syncthreads();
        uint32_t* dst = reinterpret_cast<uint32_t*>(stage + (size_t)row * stride);
        const int nwo = (n + 5 + 3) >> 2;
        for (int i = tid; i < nwo; i += NT) dst[i] = out[i];
        return;
    }
    // the chosen code into LDS, its block header into the bit buffer
    const PngTable& T = tabs[choice];
    const uint32_t hdr_bits = T.hdr_bits, dist_len = T.dist_len, dist_code = T.dist_code;
    for (int i = tid; i < PNG_NSYM; i += NT) tab[i] = T.sym[i];
    if (tid == 0) atomicOr(&out[0], T.btype << 1);                  // BFINAL = 0, BTYPE (LSB first)
    for (int i = tid; i < (int)((hdr_bits + 31u) >> 5); i += NT) {
        const uint32_t wd = T.hdr[i];
        atomicOr(&out[i], wd << 3);
        if (wd >> 29) atomicOr(&out[i + 1], wd >> 29);
    }
    __syncthreads();
    // pass 2: bit count of every step under the chosen code
    for (int sidx = wave; sidx < nsteps; sidx += PNG_ROW_WAVES) {
        const int p = (sidx << 6) + lane;
        const uint32_t d = p < n ? desc[p] : 0u;
        uint32_t nb = 0;
        if (d) {
            const uint32_t sym = d & 511u;
            nb = tab[sym] >> 16;
            if (sym >= 257u) nb += (sym < 265u ? 0u : (sym - 261u) >> 2) + dist_len;
        }
        const int total = __popcll(__ballot(nb & 1u)) + 2 * __popcll(__ballot(nb & 2u)) + 4 * __popcll(__ballot(nb & 4u)) + 8 * __popcll(__ballot(nb & 8u)) +
                          16 * __popcll(__ballot(nb & 16u));
        if (lane == 0) step_bits[sidx] = (uint32_t)total;
    }
    __syncthreads();
    // exclusive scan of the step bit counts (wave 0; 64 steps per pass)
    if (wave == 0) {
        uint32_t carry = 3u + hdr_bits;
        for (int s0 = 0; s0 < nsteps; s0 += 64) {
            const int i = s0 + lane;
            const uint32_t x = i < nsteps ? step_bits[i] : 0u;
            uint32_t incl = x;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if (lane >= d) incl += y; }
            if (i < nsteps) step_bits[i] = carry + incl - x;
            carry += __shfl(incl, 63);
        }
        if (lane == 0) step_bits[nsteps] = carry;
    }
    __syncthreads();
    // pass 3: form the tokens and OR them in
    for (int sidx = wave; sidx < nsteps; sidx += PNG_ROW_WAVES) {
        const int p = (sidx << 6) + lane;
        const uint32_t d = p < n ? desc[p] : 0u;
        uint32_t nb = 0, val = 0;
        if (d) {
            const uint32_t sym = d & 511u, e = tab[sym];
            nb = e >> 16; val = e & 0xFFFFu;
            if (sym >= 257u) {
                const uint32_t L = ((d >> 9) & 63u) + 3u;
                const uint32_t eb = sym < 265u ? 0u : (sym - 261u) >> 2;
                const uint32_t ev = L <= 10u ? 0u : L <= 18u ? (L - 11u) & 1u : L <= 34u ? (L - 19u) & 3u : (L - 35u) & 7u;
                val |= ev << nb; nb += eb;
                val |= dist_code << nb; nb += dist_len;
            }
        }
        // exclusive prefix of nb (< 32) over the wave from five ballots: no cross-lane data movement
        const unsigned long long b0 = __ballot(nb & 1u), b1 = __ballot(nb & 2u), b2 = __ballot(nb & 4u), b3 = __ballot(nb & 8u), b4 = __ballot(nb & 16u);
        if (nb) {
            const int excl = __popcll(b0 & lt) + 2 * __popcll(b1 & lt) + 4 * __popcll(b2 & lt) + 8 * __popcll(b3 & lt) + 16 * __popcll(b4 & lt);
            const int o = (int)step_bits[sidx] + excl;
            const unsigned long long vv = (unsigned long long)val << (o & 31);
            atomicOr(&out[o >> 5], (uint32_t)vv);
            if (vv >> 32) atomicOr(&out[(o >> 5) + 1], (uint32_t)(vv >> 32));
        }
    }
    // end of block, then an empty stored block (3 zero bits, pad to a byte, 00 00 FF FF)
    const int endbits = (int)step_bits[nsteps];
    const uint32_t eob = tab[256];
    if (tid == 0) {
        const unsigned long long vv = (unsigned long long)(eob & 0xFFFFu) << (endbits & 31);
        atomicOr(&out[endbits >> 5], (uint32_t)vv);
        if (vv >> 32) atomicOr(&out[(endbits >> 5) + 1], (uint32_t)(vv >> 32));
    }
    __syncthreads();
    const int bitpos = endbits + (int)(eob >> 16) + 3;
    const int bo = (bitpos + 7) >> 3;
    const int size = bo + 4;
    if (tid == 0) {
        uint8_t* o8 = reinterpret_cast<uint8_t*>(out);
        o8[bo + 2] = 0xFF; o8[bo + 3] = 0xFF;
        sizes[row] = (uint32_t)size;
        unsigned long long ta = 0, tb2 = 0;
        for (int w = 0; w < PNG_ROW_WAVES; ++w) { ta += red_a[w]; tb2 += red_b[w]; }
        adler[row] = make_uint2((uint32_t)(ta % 65521ull), (uint32_t)(tb2 % 65521ull));
    }
    __syncthreads();
    uint32_t* dst = reinterpret_cast<uint32_t*>(stage + (size_t)row * stride);
    const int nwo = (size + 3) >> 2;
    for (int i = tid; i < nwo; i += NT) dst[i] = out[i];
}

struct PngHeader { uint8_t b[33]; };            // signature + IHDR chunk (host-built)

// ---------------------------------------------------------------------------------------------------------------------------------
// kernel 2: one block per row: place the row at its byte offset and leave its CRC-32 contribution in crc_part[row].  No fences, no
// atomics: the combination happens in kernel 3 behind the kernel boundary.
__global__ __launch_bounds__(256) void png_pack_kernel(const uint8_t* stage, int stride, const uint32_t* sizes, int H, uint8_t* png,
                                                       uint32_t* crc_part, unsigned long long* total_out, CrcTables tb)
{
    extern __shared__ uint32_t lds[];           // the row's bytes (stride)
    __shared__ uint32_t redx[8];
    const int t = threadIdx.x, row = blockIdx.x;
    // offsets: sum of the sizes before this row, and of all rows (both < 2^32: H <= 65535 rows of <= 27 KB); wave shuffles + one LDS hop
    uint32_t before = 0, all = 0;
    for (int i = t; i < H; i += 256) { const uint32_t s = sizes[i]; all += s; if (i < row) before += s; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { before += __shfl_xor(before, d); all += __shfl_xor(all, d); }
    if ((t & 63) == 0) { redx[t >> 6] = before; redx[4 + (t >> 6)] = all; }
    __syncthreads();
    before = redx[0] + redx[1] + redx[2] + redx[3]; all = redx[4] + redx[5] + redx[6] + redx[7];
    __syncthreads();
    const int size = (int)sizes[row];
    const uint32_t* src = reinterpret_cast<const uint32_t*>(stage + (size_t)row * stride);
    const int nws = (size + 3) >> 2;
    for (int i = t; i < nws + 1; i += 256) lds[i] = i < nws ? src[i] : 0u;
    __syncthreads();
    const uint8_t* s8 = reinterpret_cast<const uint8_t*>(lds);
    // copy to png[43 + before ...): whole destination dwords through a funnel shift, ragged ends byte-wise
    const unsigned long long D = 43ull + before;
    const unsigned long long d_first = (D + 3) >> 2, d_end = (D + (unsigned long long)size) >> 2;      // dwords [d_first, d_end) lie inside
    uint32_t* png32 = reinterpret_cast<uint32_t*>(png);
    if (d_end > d_first) {
        for (unsigned long long d = d_first + t; d < d_end; d += 256) {
            const int si = (int)(4ull * d - D);
            const uint32_t lo = lds[si >> 2], hi = lds[(si >> 2) + 1];
            png32[d] = __funnelshift_r(lo, hi, 8 * (si & 3));
        }
        const int head = (int)(4ull * d_first - D);                                                    // 0..3 bytes before the first whole dword
        if (t < head) png[D + t] = s8[t];
        const int tail0 = (int)(4ull * d_end - D);                                                     // bytes after the last whole dword
        if (t < size - tail0) png[D + tail0 + t] = s8[tail0 + t];
    } else {
        for (int i = t; i < size; i += 256) png[D + i] = s8[i];
    }
    // CRC-32 of the row's bytes: 16-byte pieces counted from the END (a piece's power is then x^(8 * 16 * index)), XOR of the raised parts
    uint32_t acc = 0;
    const int npieces = (size + 15) >> 4;
    for (int j = t; j < npieces; j += 256) {
        const int hi = size - 16 * j, lo = hi - 16 > 0 ? hi - 16 : 0;
        const uint32_t c = crc_bytes(s8 + lo, hi - lo);
        acc ^= crc_mulmod(crc_xpow8(tb, 16u * (unsigned)j), c);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc ^= __shfl_xor(acc, d);
    if ((t & 63) == 0) redx[t >> 6] = acc;
    __syncthreads();
    if (t == 0) {
        const unsigned long long after = (unsigned long long)all - before - (unsigned long long)size + 6ull;      // + final block (2) + Adler-32 (4)
        crc_part[row] = crc_mulmod(crc_xpow8(tb, (unsigned)after), redx[0] ^ redx[1] ^ redx[2] ^ redx[3]);
        if (row == 0) *total_out = all;
    }
}

// kernel 3 (one block): XOR of the rows' CRC parts + chunk prelude + trailer, Adler-32 from the rows' parts, header, length, IEND, size.
// It stays a kernel of its own on purpose: the file size it stores last may sit in host-mapped memory that the host polls before it
// starts a DMA out of `png` -- the rows' bytes must have left the XCDs' L2 caches by then, which the boundary behind the pack
// kernel guarantees and a "last block finishes" scheme inside it would not (short of a system-scope fence in every block: the
// first version's 88 us).
__global__ __launch_bounds__(256) void png_finish_kernel(const uint32_t* crc_part, const uint2* adler, const unsigned long long* total_in, int W, int H,
                                                         uint8_t* png, uint32_t* png_bytes, PngHeader hdr, CrcTables tb)
{
    __shared__ unsigned long long red[8];
    __shared__ uint32_t redx[4];
    const int t = threadIdx.x;
    const unsigned long long all = *total_in;
    const unsigned long long n = 3ull * W + 1ull;
    unsigned long long s1 = 0, s2 = 0; uint32_t x = 0;
    for (int k = t; k < H; k += 256) {
        const uint2 ab = adler[k];
        s1 += ab.x;
        s2 += (ab.y + ((n * (unsigned long long)(H - 1 - k)) % 65521ull) * ab.x) % 65521ull;
        x ^= crc_part[k];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { s1 += __shfl_xor(s1, d); s2 += __shfl_xor(s2, d); x ^= __shfl_xor(x, d); }
    if ((t & 63) == 0) { red[t >> 6] = s1; red[4 + (t >> 6)] = s2; redx[t >> 6] = x; }
    __syncthreads();
    s1 = red[0] + red[1] + red[2] + red[3]; s2 = red[4] + red[5] + red[6] + red[7];
    x = redx[0] ^ redx[1] ^ redx[2] ^ redx[3];
    const unsigned long long E = 43ull + all;                      // end of the row data
    if (t < 33) png[t] = hdr.b[t];
    __syncthreads();                                               // the header bytes happen-before thread 64's system-scope release below
    if (t == 64) {
        const uint32_t a1 = (uint32_t)((1ull + s1) % 65521ull);
        const uint32_t a2 = (uint32_t)((((n % 65521ull) * ((unsigned long long)H % 65521ull)) % 65521ull + s2) % 65521ull);
        const uint32_t ad = (a2 << 16) | a1;
        uint8_t tr[6] = {0x03, 0x00, (uint8_t)(ad >> 24), (uint8_t)(ad >> 16), (uint8_t)(ad >> 8), (uint8_t)ad};
        const uint8_t pre[6] = {'I', 'D', 'A', 'T', 0x78, 0x01};
        uint32_t crc = x;
        crc ^= crc_mulmod(crc_xpow8(tb, (unsigned)(all + 6ull)), crc_bytes(pre, 6));
        crc ^= crc_bytes(tr, 6);
        const uint32_t len = (uint32_t)(all + 8ull);               // zlib header (2) + rows + final block (2) + Adler-32 (4)
        uint8_t tail[26];
        for (int i = 0; i < 6; ++i) tail[i] = tr[i];
        tail[6] = (uint8_t)(crc >> 24); tail[7] = (uint8_t)(crc >> 16); tail[8] = (uint8_t)(crc >> 8); tail[9] = (uint8_t)crc;
        const uint8_t iend[12] = {0, 0, 0, 0, 'I', 'E', 'N', 'D', 0xAE, 0x42, 0x60, 0x82};
        for (int i = 0; i < 12; ++i) tail[10 + i] = iend[i];
        for (int i = 0; i < 22; ++i) png[E + i] = tail[i];
        png[33] = (uint8_t)(len >> 24); png[34] = (uint8_t)(len >> 16); png[35] = (uint8_t)(len >> 8); png[36] = (uint8_t)len;
        for (int i = 0; i < 6; ++i) png[37 + i] = pre[i];
        __threadfence_system();                                    // the bytes above before the size (png_bytes may be host-mapped: the host polls it)
        *png_bytes = (uint32_t)(E + 22ull);
        __threadfence_system();
    }
}

uint32_t host_crc32(const uint8_t* s, size_t n)
{
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) { c ^= s[i]; for (int b = 0; b < 8; ++b) c = (c >> 1) ^ (CRC_POLY & (0u - (c & 1u))); }
    return ~c;
}

}  // namespace

size_t png_capacity(int W, int H) { return 43 + (size_t)H * png_row_stride(W) + 6 + 4 + 12 + 8; }
// workspace: [total: 16 B][sizes: H words][crc parts: H words][adler: H uint2][stage: H * stride], each part 16-byte aligned
size_t png_workspace_bytes(int W, int H)
{
    const size_t h4 = ((size_t)H * 4 + 15) & ~(size_t)15, h8 = ((size_t)H * 8 + 15) & ~(size_t)15;
    return 16 + 2 * h4 + h8 + (size_t)H * png_row_stride(W) + 16;
}

namespace {
// the encoder's Huffman codes (png_tables.cpp), uploaded once per device
const PngTable* png_tables_for_device(int device)
{
    static std::mutex mu;
    static std::vector<std::pair<int, PngTable*>> cache;
    std::lock_guard<std::mutex> lock(mu);
    for (auto& kv : cache) if (kv.first == device) return kv.second;
    const std::vector<PngTable>& t = png_tables();
    for (const PngTable& x : t) if (x.hdr_bits == 0xFFFFFFFFu) return nullptr;
    PngTable* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), t.size() * sizeof(PngTable)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, t.data(), t.size() * sizeof(PngTable), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
    cache.emplace_back(device, d);
    return d;
}

// x^(8 n) tables (4 x 256 words), built on the host with the same arithmetic the kernels use
std::vector<uint32_t> crc_pow_tables()
{
    std::vector<uint32_t> t(1024);
    uint32_t x8 = 1u << 31;                                         // x^0
    for (int k = 0; k < 8; ++k) x8 = crc_mulmod(x8, 1u << 30);      // x^8: one byte
    uint32_t step = x8;
    for (int tbl = 0; tbl < 4; ++tbl) {
        uint32_t p = 1u << 31;
        for (int i = 0; i < 256; ++i) { t[tbl * 256 + i] = p; p = crc_mulmod(p, step); }
        step = p;                                                   // step^256: the next table's unit
    }
    return t;
}

// ... uploaded once per device
const uint32_t* crc_tables_for_device(int device)
{
    static std::mutex mu;
    static std::vector<std::pair<int, uint32_t*>> cache;
    std::lock_guard<std::mutex> lock(mu);
    for (auto& kv : cache) if (kv.first == device) return kv.second;
    const std::vector<uint32_t> t = crc_pow_tables();
    uint32_t* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), t.size() * 4) != hipSuccess) return nullptr;
    if (hipMemcpy(d, t.data(), t.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
    cache.emplace_back(device, d);
    return d;
}
}  // namespace

// crc32(A || B) from crc32(A), crc32(B), |B| with the tables and the arithmetic the kernels use (host-only: the CPU suite pins the
// combine against zlib beyond 2^24 bytes, where three tables used to end)
uint32_t png_crc32_combine_host(uint32_t crc_a, uint32_t crc_b, uint32_t len_b)
{
    static const std::vector<uint32_t> t = crc_pow_tables();
    return crc_mulmod(crc_xpow8_tab(t.data(), len_b), crc_a) ^ crc_b;
}

int launch_png_encode(const uint8_t* rgb_hwc, const float* rgb_planar, int W, int H, void* png_out, size_t capacity, uint32_t* png_bytes,
                      void* workspace, size_t ws_bytes, hipStream_t st)
{
    FAV_REQUIRE((rgb_hwc != nullptr) != (rgb_planar != nullptr), "png: exactly one of the u8 and the float source must be given");
    FAV_REQUIRE(W >= 1 && H >= 1 && W <= 9000 && H <= 65535, "png: %dx%d is outside the encoder's range (width <= 9000, height <= 65535)", W, H);
    FAV_REQUIRE(png_out && png_bytes && workspace, "png: null argument");
    FAV_REQUIRE(capacity >= png_capacity(W, H), "png: output capacity %zu < fav_png_capacity = %zu", capacity, png_capacity(W, H));
    FAV_REQUIRE(png_capacity(W, H) < ((size_t)1 << 32) - 64, "png: %dx%d needs a file of more than 4 GiB (IDAT length and the CRC combine are 32-bit)", W, H);
    FAV_REQUIRE(ws_bytes >= png_workspace_bytes(W, H), "png: workspace %zu < fav_png_workspace_bytes = %zu", ws_bytes, png_workspace_bytes(W, H));
    FAV_REQUIRE((reinterpret_cast<uintptr_t>(png_out) & 3) == 0 && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "png: output must be 4-byte, workspace 16-byte aligned");
    int device = 0; FAV_HIP(hipGetDevice(&device));
    CrcTables tb{crc_tables_for_device(device)};
    if (!tb.t) return hip_fail(hipErrorOutOfMemory, "png: CRC tables");
    const PngTable* tabs = png_tables_for_device(device);
    if (!tabs) return hip_fail(hipErrorOutOfMemory, "png: Huffman tables");
    const int stride = png_row_stride(W);
    const size_t h4 = ((size_t)H * 4 + 15) & ~(size_t)15, h8 = ((size_t)H * 8 + 15) & ~(size_t)15;
    uint8_t* ws = static_cast<uint8_t*>(workspace);
    unsigned long long* total = reinterpret_cast<unsigned long long*>(ws);
    uint32_t* sizes = reinterpret_cast<uint32_t*>(ws + 16);
    uint32_t* crc_part = reinterpret_cast<uint32_t*>(ws + 16 + h4);
    uint2* adler = reinterpret_cast<uint2*>(ws + 16 + 2 * h4);
    uint8_t* stage = ws + 16 + 2 * h4 + h8;
    const int nraw = 3 * W, n = nraw + 1;
    const size_t lds1 = ((size_t)((nraw + 6) / 4 + 1) + (size_t)((n + 3) / 4 + 1) + (size_t)((n + 1) / 2 + 1) + (size_t)((n + 63) / 64 + 1) + (size_t)stride / 4) * 4;
    const size_t lds2 = (size_t)stride + 8;
    if (lds1 > 48 * 1024 || lds2 > 48 * 1024) {
        // (rows wider than ~2600 pixels: raise the dynamic LDS limit, once per DEVICE -- the attribute is the device's, not the
        //  calling thread's; 160 KB per CU on gfx950)
        static std::mutex mu;
        static std::vector<int> done;
        std::lock_guard<std::mutex> lock(mu);
        bool have = false;
        for (int d : done) have |= d == device;
        if (!have) {
            // (160 KB minus the kernel's static arrays -- the per-wave histograms: 10.6 KB with eight waves per row; the widest row, 9000
            //  pixels, needs 137 KB)
            hipFuncAttributes fa;
            FAV_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(png_rows_kernel<true>)));
            const int dyn_t = (160 * 1024 - (int)fa.sharedSizeBytes) & ~255;
            FAV_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(png_rows_kernel<false>)));
            const int dyn_f = (160 * 1024 - (int)fa.sharedSizeBytes) & ~255;
            FAV_REQUIRE((size_t)std::min(dyn_t, dyn_f) >= lds1, "png: a row of %d pixels needs %zu bytes of LDS, %d are there", W, lds1, std::min(dyn_t, dyn_f));
            FAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(png_rows_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, dyn_t));
            FAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(png_rows_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, dyn_f));
            FAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(png_pack_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 16384));
            done.push_back(device);
        }
    }
    if (rgb_planar)
        hipLaunchKernelGGL(png_rows_kernel<true>, dim3(H), dim3(64 * PNG_ROW_WAVES), lds1, st, nullptr, rgb_planar, W, H, stage, stride, sizes, adler, tabs);
    else
        hipLaunchKernelGGL(png_rows_kernel<false>, dim3(H), dim3(64 * PNG_ROW_WAVES), lds1, st, rgb_hwc, nullptr, W, H, stage, stride, sizes, adler, tabs);
    FAV_LAUNCH_CHECK("png_rows_kernel");
    PngHeader hdr;
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    memcpy(hdr.b, sig, 8);
    uint8_t* q = hdr.b + 8;
    q[0] = 0; q[1] = 0; q[2] = 0; q[3] = 13; memcpy(q + 4, "IHDR", 4);
    q[8] = (uint8_t)(W >> 24); q[9] = (uint8_t)(W >> 16); q[10] = (uint8_t)(W >> 8); q[11] = (uint8_t)W;
    q[12] = (uint8_t)(H >> 24); q[13] = (uint8_t)(H >> 16); q[14] = (uint8_t)(H >> 8); q[15] = (uint8_t)H;
    q[16] = 8; q[17] = 2; q[18] = 0; q[19] = 0; q[20] = 0;          // 8 bits, colour type 2 (RGB), deflate, adaptive filtering, not interlaced
    const uint32_t c = host_crc32(q + 4, 17);
    q[21] = (uint8_t)(c >> 24); q[22] = (uint8_t)(c >> 16); q[23] = (uint8_t)(c >> 8); q[24] = (uint8_t)c;
    hipLaunchKernelGGL(png_pack_kernel, dim3(H), dim3(256), lds2, st, stage, stride, sizes, H, static_cast<uint8_t*>(png_out), crc_part, total, tb);
    FAV_LAUNCH_CHECK("png_pack_kernel");
    hipLaunchKernelGGL(png_finish_kernel, dim3(1), dim3(256), 0, st, crc_part, adler, total, W, H, static_cast<uint8_t*>(png_out), png_bytes, hdr, tb);
    FAV_LAUNCH_CHECK("png_finish_kernel");
    return FAV_OK;
}

}  // namespace fav
