// first2d_pack.h -- host-side weight transform for the 2-D minimal-filtering form of the first layer (c9s1-32 on 7 / 3 real input
// channels, conv_first2d_kernel in kernels_first.hip).  Plain C++ (no HIP): the CPU test suite compiles it on its own and checks the
// layout against a lane-level restatement of the kernel.
//
// A 9x9 correlation is the sum of nine 3x3 correlations: filter rows ky = 3 a + r, columns kx = 3 b + s (a, b, r, s = 0..2) read
// the input at offset (3 a, 3 b).  Each of them is computed with Winograd F(2x2, 3x3) (Lavin & Gray 2016; matrices as in
// wino_pack.h): for the 2x2 output tile (2 ty .. 2 ty + 1, 2 tx .. 2 tx + 1) and the 4x4 patch d of channel c at (2 ty + 3 a, 2 tx + 3 b)
//     M[i][j] += (B^T d B)[i][j] * (G g_{c,a,b} G^T)[i][j]          summed over all (c, a, b) IN THE TRANSFORMED DOMAIN
//     Y = A^T M A                                                    once per tile
// = 16 multiplies per (c, a, b) and 4 outputs against 24 for the 1-D form of first_pack.h and 36 for the direct one.  Sixteen GEMMs
// (positions) M_p[tile][cout] over k = (c, a, b), K = 9 CR (63 for the video model's 7 channels).
//
// v_mfma_f32_16x16x4_f32 multiplies FOUR k per instruction, one per group of 16 lanes; any four may share one as long as every
// lane group knows its own operand address (a per-lane offset per instruction, computed once).  K is cut into quads in the order
//     k = (c, a, b) -> index 9 c + 3 a + b,  quad q = index / 4, lane group = index % 4          (index 9 CR .. : zero weights)
//
// out[((((p * NQ + q) * 2 + nt) * 4 + g) * 16 + n]     p = position 4 i + j, q = quad, nt = half of the 32 output channels,
//                                                      g = lane group (k inside the quad), n = output channel nt * 16 + n
// -- the 64 lanes of one B-operand read are 64 consecutive words.
#pragma once
#include <cstddef>
#include <vector>

namespace fav {

inline int conv_first2d_quads(int cr) { return (9 * cr + 3) / 4; }

// the (c, a, b) of quad q, lane group g; c = -1: no tap (zero weight)
inline void conv_first2d_combo(int cr, int q, int g, int* c, int* a, int* b)
{
    const int idx = 4 * q + g;
    if (idx >= 9 * cr) { *c = -1; *a = 0; *b = 0; return; }
    *c = idx / 9; *a = (idx % 9) / 3; *b = idx % 3;
}

// w: [cout][cin][9][9] with cout <= 32, cin = cr (7 or 3)
inline void conv_first2d_pack(const float* w, int cin, int cout, std::vector<float>& out)
{
    static const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    const int NQ = conv_first2d_quads(cin);
    out.assign((size_t)16 * NQ * 2 * 4 * 16, 0.f);
    for (int n = 0; n < cout && n < 32; ++n)
        for (int q = 0; q < NQ; ++q)
            for (int g = 0; g < 4; ++g) {
                int c, a, b;
                conv_first2d_combo(cin, q, g, &c, &a, &b);
                if (c < 0) continue;
                double t[3][3];
                for (int r = 0; r < 3; ++r)
                    for (int s = 0; s < 3; ++s) t[r][s] = (double)w[(((size_t)n * cin + c) * 9 + 3 * a + r) * 9 + 3 * b + s];
                double Gg[4][3];
                for (int i = 0; i < 4; ++i)
                    for (int s = 0; s < 3; ++s) Gg[i][s] = G[i][0] * t[0][s] + G[i][1] * t[1][s] + G[i][2] * t[2][s];
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j) {
                        const double u = Gg[i][0] * G[j][0] + Gg[i][1] * G[j][1] + Gg[i][2] * G[j][2];
                        out[(((((size_t)(4 * i + j) * NQ + q) * 2 + (n >> 4)) * 4 + g) * 16 + (n & 15))] = (float)u;
                    }
            }
}

// layers with more than 32 filters: computed in groups of 32 output channels, each with the packed block conv_first2d_pack() makes
// of its (up to) 32 filters -- the blocks follow each other; coutp = the padded filter count (a multiple of 32)
inline void conv_first2d_pack_groups(const float* w, int cin, int cout, int coutp, std::vector<float>& out)
{
    out.clear();
    std::vector<float> one;
    for (int g = 0; g < coutp / 32; ++g) {
        const int n = cout - 32 * g < 0 ? 0 : (cout - 32 * g > 32 ? 32 : cout - 32 * g);
        conv_first2d_pack(w + (size_t)g * 32 * cin * 81, cin, n, one);
        out.insert(out.end(), one.begin(), one.end());
    }
}

}  // namespace fav
