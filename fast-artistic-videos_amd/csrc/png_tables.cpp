// png_tables.cpp -- the Huffman codes of the GPU PNG encoder (kernels_png.hip), built ONCE per process on the host.
//
// A deflate block may bring its own code (BTYPE 10) in a header of 40-90 bytes.  Building an optimal code per row on the device
// would put a sort and a tree walk into a kernel that takes 20 us; instead the encoder carries PNG_NTABLES ready-made codes, each
// the length-limited (15 bits) Huffman code of a model distribution of Sub-filtered image rows -- literals two-sided geometric
// around 0 (P(v) ~ theta^min(v, 256 - v)) for ten values of theta, plus two codes for rows that are mostly runs -- together with
// its dynamic-block header, and the row kernel picks, per row and by EXACT bit count from the row's token histogram, the cheapest
// of: these codes, the fixed code of RFC 1951 3.2.6, a stored block.  On image rows that is within a few per cent of the per-row
// optimal code at the cost of one table look-up per token.
//
// Everything here is plain RFC 1951: package-merge for the length-limited codes, canonical code assignment (3.2.2), the
// code-length alphabet with its run-length symbols 16 / 17 / 18 (3.2.7).  oracle/png_model.py reads the same tables through
// fav_png_tables_host and restates the kernels' bit layout; zlib / PIL decode what both produce (tests/test_cpu_png.py).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

#include "fav_internal.h"

namespace fav {

namespace {

// length-limited Huffman code lengths (package-merge, Larmore & Hirschberg): every symbol has a positive weight
std::vector<int> limited_lengths(const std::vector<double>& w, int maxlen)
{
    const int n = (int)w.size();
    std::vector<int> len(n, 0);
    if (n == 1) { len[0] = 1; return len; }
    struct Node { double w; std::vector<int> syms; };
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return w[a] < w[b]; });
    std::vector<Node> items(n);
    for (int i = 0; i < n; ++i) { items[i].w = w[order[i]]; items[i].syms = {order[i]}; }
    std::vector<Node> cur = items;                                  // the deepest level: the items alone
    for (int level = 1; level < maxlen; ++level) {
        std::vector<Node> pk;
        for (size_t i = 0; i + 1 < cur.size(); i += 2) {
            Node p; p.w = cur[i].w + cur[i + 1].w; p.syms = cur[i].syms;
            p.syms.insert(p.syms.end(), cur[i + 1].syms.begin(), cur[i + 1].syms.end());
            pk.push_back(std::move(p));
        }
        std::vector<Node> merged; merged.reserve(items.size() + pk.size());
        size_t a = 0, b = 0;
        while (a < items.size() || b < pk.size()) {                 // ties: items first (deterministic)
            if (b >= pk.size() || (a < items.size() && items[a].w <= pk[b].w)) merged.push_back(items[a++]);
            else merged.push_back(std::move(pk[b++]));
        }
        cur.swap(merged);
    }
    for (int i = 0; i < 2 * n - 2; ++i)
        for (int s : cur[i].syms) ++len[s];
    return len;
}

// canonical codes (RFC 1951 3.2.2), returned bit-reversed: ready to be OR-ed in LSB first
std::vector<uint32_t> canonical_reversed(const std::vector<int>& len, int maxbits)
{
    std::vector<int> bl(maxbits + 1, 0);
    for (int l : len) if (l) ++bl[l];
    std::vector<uint32_t> next(maxbits + 2, 0);
    uint32_t code = 0;
    for (int b = 1; b <= maxbits; ++b) { code = (code + (uint32_t)bl[b - 1]) << 1; next[b] = code; }
    std::vector<uint32_t> out(len.size(), 0);
    for (size_t i = 0; i < len.size(); ++i) {
        if (!len[i]) continue;
        const uint32_t c = next[len[i]]++;
        uint32_t r = 0;
        for (int b = 0; b < len[i]; ++b) if (c & (1u << b)) r |= 1u << (len[i] - 1 - b);
        out[i] = r;
    }
    return out;
}

struct BitWriter {
    std::vector<uint32_t> words; int bits = 0;
    void put(uint32_t v, int n)
    {
        for (int i = 0; i < n; ++i) {
            if ((bits >> 5) >= (int)words.size()) words.push_back(0);
            if ((v >> i) & 1u) words[bits >> 5] |= 1u << (bits & 31);
            ++bits;
        }
    }
};

void build_table(const std::vector<double>& freq277, PngTable& t)
{
    const std::vector<int> ll = limited_lengths(freq277, 15);
    const std::vector<uint32_t> lc = canonical_reversed(ll, 15);
    memset(&t, 0, sizeof t);
    for (int s = 0; s < PNG_NSYM; ++s) t.sym[s] = ((uint32_t)ll[s] << 16) | lc[s];
    t.btype = 2; t.dist_len = 1; t.dist_code = 0;                  // one distance code (code 2 = distance 3) of one bit: '0'
    // ---- the block header behind the three bits BFINAL / BTYPE: HLIT, HDIST, HCLEN, the code-length code, the lengths
    std::vector<int> seq(ll.begin(), ll.end());                     // 277 literal / length code lengths ...
    seq.push_back(0); seq.push_back(0); seq.push_back(1);           // ... + distance codes 0, 1 (unused), 2 (one bit)
    struct Cl { int sym, extra, nextra; };
    std::vector<Cl> cls;
    for (size_t i = 0; i < seq.size();) {
        size_t j = i; while (j < seq.size() && seq[j] == seq[i]) ++j;
        int run = (int)(j - i);
        if (seq[i] == 0) {
            while (run >= 11) { const int r = std::min(run, 138); cls.push_back({18, r - 11, 7}); run -= r; }
            if (run >= 3) { cls.push_back({17, run - 3, 3}); run = 0; }
            while (run-- > 0) cls.push_back({0, 0, 0});
        } else {
            cls.push_back({seq[i], 0, 0}); --run;
            while (run >= 3) { const int r = std::min(run, 6); cls.push_back({16, r - 3, 2}); run -= r; }
            while (run-- > 0) cls.push_back({seq[i], 0, 0});
        }
        i = j;
    }
    std::vector<double> cf(19, 0.0);
    for (const Cl& c : cls) cf[c.sym] += 1.0;
    std::vector<int> used; for (int s = 0; s < 19; ++s) if (cf[s] > 0) used.push_back(s);
    std::vector<double> uw; for (int s : used) uw.push_back(cf[s]);
    const std::vector<int> ul = limited_lengths(uw, 7);             // (at least two symbols are in use: a complete code)
    std::vector<int> cl_len(19, 0);
    for (size_t i = 0; i < used.size(); ++i) cl_len[used[i]] = ul[i];
    const std::vector<uint32_t> cl_code = canonical_reversed(cl_len, 7);
    static const int perm[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19; while (hclen > 4 && cl_len[perm[hclen - 1]] == 0) --hclen;
    BitWriter bw;
    bw.put(PNG_NSYM - 257, 5); bw.put(3 - 1, 5); bw.put((uint32_t)(hclen - 4), 4);
    for (int i = 0; i < hclen; ++i) bw.put((uint32_t)cl_len[perm[i]], 3);
    for (const Cl& c : cls) { bw.put(cl_code[c.sym], cl_len[c.sym]); if (c.nextra) bw.put((uint32_t)c.extra, c.nextra); }
    t.hdr_bits = (uint32_t)bw.bits;
    for (size_t i = 0; i < bw.words.size() && i < PNG_HDR_WORDS; ++i) t.hdr[i] = bw.words[i];
    if (bw.words.size() > PNG_HDR_WORDS) t.hdr_bits = 0xFFFFFFFFu;  // (never: checked by the caller)
}

std::vector<double> model(double theta, double match_mass)
{
    std::vector<double> f(PNG_NSYM, 0.0);
    double z = 0;
    for (int v = 0; v < 256; ++v) { f[v] = std::pow(theta, (double)std::min(v, 256 - v)) + 2e-5; z += f[v]; }
    for (int v = 0; v < 256; ++v) f[v] *= (1.0 - match_mass) / z;
    f[0] += 0.25 / 3841.0;                                          // the filter-type byte of every row shares a code with its neighbours
    f[256] = 1.0 / 3841.0;                                          // one end-of-block per row
    // runs are cut at 64 positions: long runs come as many tokens of length 64 (symbol 276), the rest spread over the length symbols
    for (int s = 257; s < PNG_NSYM; ++s) f[s] = match_mass * (s == 276 ? 0.5 : 0.5 / 19.0);
    return f;
}

}  // namespace

const std::vector<PngTable>& png_tables()
{
    static std::vector<PngTable> tabs;
    static std::once_flag once;
    std::call_once(once, [] {
        static const double thetas[10] = {0.30, 0.50, 0.65, 0.75, 0.82, 0.88, 0.92, 0.95, 0.97, 0.985};
        std::vector<PngTable> t(PNG_NTABLES + 1);
        for (int k = 0; k < 10; ++k) build_table(model(thetas[k], 0.02), t[k]);
        build_table(model(0.55, 0.60), t[10]);                      // mostly runs, sharp literals (flat artwork, letterbox bars)
        build_table(model(0.92, 0.60), t[11]);                      // mostly runs, broad literals (gradients with dither)
        // index PNG_NTABLES: the fixed code of RFC 1951 3.2.6 in the same form (no header)
        PngTable& fx = t[PNG_NTABLES];
        memset(&fx, 0, sizeof fx);
        std::vector<int> fl(288, 8);
        for (int s = 144; s < 256; ++s) fl[s] = 9;
        for (int s = 256; s < 280; ++s) fl[s] = 7;
        const std::vector<uint32_t> fc = canonical_reversed(fl, 9);
        for (int s = 0; s < PNG_NSYM; ++s) fx.sym[s] = ((uint32_t)fl[s] << 16) | fc[s];
        fx.btype = 1; fx.dist_len = 5; fx.dist_code = 8;            // distance code 2 in five bits, reversed: 01000
        fx.hdr_bits = 0;
        tabs.swap(t);
    });
    return tabs;
}

}  // namespace fav

// host-only view of the tables (no device needed): count = PNG_NTABLES + 1 (the last one is the fixed code), each
// sizeof(fav::PngTable) bytes: sym[277] = (length << 16) | bit-reversed code, hdr[PNG_HDR_WORDS], hdr_bits, btype, dist_len, dist_code
extern "C" int fav_png_tables_host(void* out_host, size_t capacity, int* count, int* table_bytes)
{
    const std::vector<fav::PngTable>& t = fav::png_tables();
    if (count) *count = (int)t.size();
    if (table_bytes) *table_bytes = (int)sizeof(fav::PngTable);
    for (const fav::PngTable& x : t)
        if (x.hdr_bits == 0xFFFFFFFFu) { fav::set_error("png tables: header longer than %d words", fav::PNG_HDR_WORDS); return FAV_EINVAL; }
    if (out_host) {
        if (capacity < t.size() * sizeof(fav::PngTable)) { fav::set_error("fav_png_tables_host: capacity too small"); return FAV_EINVAL; }
        memcpy(out_host, t.data(), t.size() * sizeof(fav::PngTable));
    }
    return FAV_OK;
}
