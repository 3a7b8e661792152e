// first_pack.h -- host-side weight transform for the first-layer kernel (c9s1-32 on 7 / 3 real input channels, kernels_first.hip).
// Plain C++ (no HIP): the CPU test suite compiles it on its own and checks the layout against a lane-level restatement.
//
// The 9 taps of a filter row are three blocks of three (kx = 3 b + r), and each block is a 3-tap correlation along x computed with
// the 1-D minimal filtering F(2,3) (Winograd; Lavin & Gray 2016): for the output pair (2t, 2t+1) and the four inputs
// d_j = x[2t + 3b + j - pad] (j = 0..3)
//     m0 = (d0 - d2) g0,  m1 = (d1 + d2) (g0 + g1 + g2) / 2,  m2 = (d2 - d1) (g0 - g1 + g2) / 2,  m3 = (d1 - d3) g2
//     y(2t) = m0 + m1 + m2,   y(2t+1) = m1 - m2 - m3                         4 multiplies per 2 outputs instead of 6
// summed over (input channel c, filter row ky, block b): four GEMMs  M_p[tile][cout] = sum_k V_p[tile][k] U_p[k][cout],
// k = (c, ky, b), K = CR * 27 (189 for the video model's 7 channels).  U_p is computed here in double and rounded once.
//
// A 32x32x2 MFMA multiplies TWO k per instruction (one per half-wave), and any two k may share one as long as the second
// half-wave's input sits at a constant LDS offset from the first's.  Pairing (CR odd: 7 or 3; last = CR - 1):
//     type 0  channels (2i, 2i+1), same (ky, b)              i < CR/2, 27 each      second half: + one channel's planes
//     type 1  channel last, rows (2q, 2q+1), same b           q < 4, b < 3           second half: + one halo row
//     type 2  channel last, ky = 8, blocks 0 | 1                                     second half: + three columns
//     type 3  channel last, ky = 8, block 2 alone                                    second half: same pixel, zero weight
// = (CR/2) * 27 + 14 instructions per position and 32 tiles (95 for CR = 7: dense, against 2 x 287 / 2 for the direct form).
//
// out[((p * NJ + j) * 2 + half) * 32 + n]       p = position 0..3, j = pair index in the order above, n = output channel (< 32)
#pragma once
#include <cstddef>
#include <vector>

namespace fav {

inline int conv_first_pairs(int cr) { return (cr / 2) * 27 + 14; }

// the (c, ky, b) of pair j, half h; c = -1: no tap (zero weight)
inline void conv_first_combo(int cr, int j, int h, int* c, int* ky, int* b)
{
    const int ncc = (cr / 2) * 27, last = cr - 1;
    if (j < ncc) { *c = 2 * (j / 27) + h; *ky = (j % 27) / 3; *b = j % 3; return; }
    j -= ncc;
    if (j < 12) { *c = last; *ky = 2 * (j / 3) + h; *b = j % 3; return; }
    if (j == 12) { *c = last; *ky = 8; *b = h; return; }
    *c = h == 0 ? last : -1; *ky = 8; *b = 2;
}

// w: [cout][cin][9][9] with cout <= 32, cin = cr (7 or 3)
inline void conv_first_pack(const float* w, int cin, int cout, std::vector<float>& out)
{
    const int NJ = conv_first_pairs(cin);
    out.assign((size_t)4 * NJ * 64, 0.f);
    for (int n = 0; n < cout && n < 32; ++n)
        for (int j = 0; j < NJ; ++j)
            for (int h = 0; h < 2; ++h) {
                int c, ky, b;
                conv_first_combo(cin, j, h, &c, &ky, &b);
                if (c < 0) continue;
                const float* g = w + (((size_t)n * cin + c) * 9 + ky) * 9 + 3 * b;
                const double g0 = g[0], g1 = g[1], g2 = g[2];
                const double u[4] = {g0, 0.5 * (g0 + g1 + g2), 0.5 * (g0 - g1 + g2), g2};
                for (int p = 0; p < 4; ++p) out[(((size_t)p * NJ + j) * 2 + h) * 32 + n] = (float)u[p];
            }
}

}  // namespace fav
