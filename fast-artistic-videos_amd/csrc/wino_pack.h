// wino_pack.h -- host-side weight transform for the Winograd F(2x2, 3x3) kernel (kernels_wino.hip).  Plain C++ (no HIP), so the
// CPU test suite can compile it on its own and check the packed layout against a lane-level restatement of the kernel.
//
// Minimal filtering (Lavin & Gray 2016, the correlation form nn.SpatialConvolution computes, models_video.lua:20,32):
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        d = 4x4 input patch, g = 3x3 filter, Y = 2x2 outputs, summed over input channels
//   B^T = | 1  0 -1  0 |   G = | 1    0    0  |   A^T = | 1 1  1  0 |
//         | 0  1  1  0 |       | 1/2  1/2  1/2|         | 0 1 -1 -1 |
//         | 0 -1  1  0 |       | 1/2 -1/2  1/2|
//         | 0  1  0 -1 |       | 0    0    1  |
// U[i][j] = (G g G^T)[i][j] is computed here in double and rounded ONCE to fp32.  Positions (i, j) with j == 3 are stored NEGATED:
// the kernel builds that column's input transform with the opposite sign (c3 - c1 instead of c1 - c3, so that both wave parities
// use one instruction form), and (-V)(-U) = VU.
//
// Packed order = the order the kernel's waves consume it, so that every weight load is one contiguous 1 KiB wave access:
//   out[((((kg * 8 + wave) * 2 + q) * 4 + nt) * 64 + lane) * 4 + s]
//     kg   group of 8 input channels            wave  0..7: Winograd row i = wave >> 1, columns j = 2 * (wave & 1) + q
//     nt   tile of 32 output channels           lane  n = lane & 31 (output channel nt * 32 + n), h = lane >> 5
//     s    MFMA step inside the group: the 32x32x2 instruction of step s multiplies input channels kg*8 + s (h = 0) and
//          kg*8 + 4 + s (h = 1)
#pragma once
#include <cstddef>
#include <vector>

namespace fav {

inline size_t conv_wino_packed_floats(int cin, int cout) { return (size_t)(cin / 8) * 8 * 2 * (cout / 32) * 64 * 4; }

// w: [cout][cin][3][3] (nn.SpatialConvolution weight order), cin % 8 == 0, cout % 32 == 0
inline void conv_wino_pack(const float* w, int cin, int cout, std::vector<float>& out)
{
    static const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    const int NT = cout / 32;
    out.assign(conv_wino_packed_floats(cin, cout), 0.f);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const float* g = w + ((size_t)co * cin + ci) * 9;
            double Gg[4][3];
            for (int i = 0; i < 4; ++i)
                for (int b = 0; b < 3; ++b) Gg[i][b] = G[i][0] * (double)g[b] + G[i][1] * (double)g[3 + b] + G[i][2] * (double)g[6 + b];
            const int kg = ci >> 3, h = (ci >> 2) & 1, s = ci & 3, nt = co >> 5, n = co & 31;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const double u = Gg[i][0] * G[j][0] + Gg[i][1] * G[j][1] + Gg[i][2] * G[j][2];
                    const int wave = i * 2 + (j >> 1), q = j & 1, lane = h * 32 + n;
                    out[((((size_t)(kg * 8 + wave) * 2 + q) * NT + nt) * 64 + lane) * 4 + s] = (float)(j == 3 ? -u : u);
                }
        }
}

}  // namespace fav
