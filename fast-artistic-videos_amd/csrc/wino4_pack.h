// wino4_pack.h -- host-side weight transform and the shared constants of the Winograd F(4x4, 3x3) kernel (kernels_wino4.hip).
// Plain C++ (no HIP), so the CPU test suite can compile it on its own and check the packed layout against a lane-level restatement
// of the kernel (tests/test_cpu_wino4.py).
//
// Minimal filtering F(4x4, 3x3) for the correlation nn.SpatialConvolution computes (models_video.lua:20,32):
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A       d = 6x6 input patch, g = 3x3 filter, Y = 4x4 outputs:
//   36 multiplies per (input channel, output channel) and 4x4 outputs -- 1.78x fewer than F(2x2, 3x3) (16 per 2x2), 4x fewer than direct.
//
// Interpolation points 0, +-3/4, +-3/2, infinity instead of the textbook 0, +-1, +-2: every entry of B^T and A^T is still a dyadic
// rational (exact in fp32), the largest are 45/16 and 27/8 instead of 5 and 8, and the fp32 error of a 128 -> 128 layer measured
// 9.0e-7 rms / 9.0e-6 max against 1.9e-6 / 3.2e-5 for the textbook points (F(2x2, 3x3): 3.1e-7 / 1.7e-6; a plain fp32 accumulation
// chain: 4.3e-7 / 4.1e-6) -- scripts/wino4_points.py, profiles/r04_wino4_points.txt.
//
//   B^T = | 81/64    0    -45/16    0     1   0 |     G = |  64/81     0       0   |    A^T = | 1    1      1      1     1    0 |
//         |   0   -27/16   -9/4    3/4    1   0 |         | -128/243 -32/81  -8/27 |          | 0   3/4   -3/4    3/2  -3/2   0 |
//         |   0    27/16   -9/4   -3/4    1   0 |         | -128/243  32/81  -8/27 |          | 0   9/16   9/16   9/4   9/4   0 |
//         |   0   -27/32   -9/16   3/2    1   0 |         |   32/243  16/81   8/27 |          | 0  27/64 -27/64  27/8 -27/8   1 |
//         |   0    27/32   -9/16  -3/2    1   0 |         |   32/243 -16/81   8/27 |
//         |   0    81/64     0    -45/16  0   1 |         |     0       0       1  |
// U = G g G^T is computed here in double and rounded ONCE to fp32.
//
// Packed order = the order the kernel's waves consume it (every weight load is one contiguous 1 KiB wave access):
//   out[(((((s * 36 + p) * NW + w) * 2 + nt) * 64 + lane) * 4 + j]
//     s     slice of 16 input channels         p = 6 i + j'  transform position (row i, column j')
//     w     pair of kernel waves 2 w, 2 w + 1 that owns output channels 32 w .. 32 w + 31 (NW = cout / 32)
//     nt    half of those (16 channels: the N of v_mfma_f32_16x16x4_f32) = kernel wave 2 w + nt: the kernel (8 waves of 16 channels)
//           addresses block (w, nt) as wave * 1 KiB
//     lane  n = lane & 15 (output channel 32 w + 16 nt + n), kq = lane >> 4
//     j     MFMA step inside the slice: the instruction of step j multiplies input channels 16 s + 4 kq + j, kq = 0..3
#pragma once
#include <cstddef>
#include <vector>

#if defined(__HIPCC__)
#define FAV_W4_HD __host__ __device__
#else
#define FAV_W4_HD
#endif

namespace fav {

// the three transforms as the kernels apply them (one 6-vector in, 6 / 6 / 4 out); T = float on the device, double in tests
struct Wino4 {
    template <typename T>
    FAV_W4_HD static inline void bt(const T d[6], T v[6])          // B^T d : 14 operations
    {
        const T e1 = d[4] - T(2.25) * d[2], o1 = T(0.75) * d[3] - T(1.6875) * d[1];
        const T e2 = d[4] - T(0.5625) * d[2], o2 = T(1.5) * d[3] - T(0.84375) * d[1];
        v[0] = T(1.265625) * d[0] + (d[4] - T(2.8125) * d[2]);
        v[1] = e1 + o1; v[2] = e1 - o1; v[3] = e2 + o2; v[4] = e2 - o2;
        v[5] = T(1.265625) * d[1] + (d[5] - T(2.8125) * d[3]);
    }
    template <typename T>
    FAV_W4_HD static inline void at(const T m[6], T y[4])          // A^T m : 12 operations
    {
        const T s1 = m[1] + m[2], d1 = m[1] - m[2], s2 = m[3] + m[4], d2 = m[3] - m[4];
        y[0] = (m[0] + s1) + s2;
        y[1] = T(0.75) * d1 + T(1.5) * d2;
        y[2] = T(0.5625) * s1 + T(2.25) * s2;
        y[3] = (T(0.421875) * d1 + T(3.375) * d2) + m[5];
    }
};

inline size_t conv_wino4_packed_floats(int cin, int cout) { return (size_t)(cin / 16) * 36 * (cout / 32) * 2 * 64 * 4; }

// w: [cout][cin][3][3] (nn.SpatialConvolution weight order), cin % 16 == 0, cout % 32 == 0
inline void conv_wino4_pack(const float* w, int cin, int cout, std::vector<float>& out)
{
    static const double G[6][3] = {{64.0 / 81.0, 0.0, 0.0},
                                   {-128.0 / 243.0, -32.0 / 81.0, -8.0 / 27.0}, {-128.0 / 243.0, 32.0 / 81.0, -8.0 / 27.0},
                                   {32.0 / 243.0, 16.0 / 81.0, 8.0 / 27.0}, {32.0 / 243.0, -16.0 / 81.0, 8.0 / 27.0},
                                   {0.0, 0.0, 1.0}};
    const int NW = cout / 32;
    out.assign(conv_wino4_packed_floats(cin, cout), 0.f);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const float* g = w + ((size_t)co * cin + ci) * 9;
            double Gg[6][3];
            for (int i = 0; i < 6; ++i)
                for (int b = 0; b < 3; ++b) Gg[i][b] = G[i][0] * (double)g[b] + G[i][1] * (double)g[3 + b] + G[i][2] * (double)g[6 + b];
            const int s = ci >> 4, kq = (ci >> 2) & 3, j = ci & 3, wv = co >> 5, nt = (co >> 4) & 1, n = co & 15;
            for (int i = 0; i < 6; ++i)
                for (int jj = 0; jj < 6; ++jj) {
                    const double u = Gg[i][0] * G[jj][0] + Gg[i][1] * G[jj][1] + Gg[i][2] * G[jj][2];
                    const int p = 6 * i + jj, lane = kq * 16 + n;
                    out[((((((size_t)s * 36 + p) * NW + wv) * 2 + nt) * 64 + lane) * 4) + j] = (float)u;
                }
        }
}

// layers with more than 128 filters (cout % 128 == 0): the kernel computes them in groups of 128 output channels, a work unit =
// (pixel unit, group), each group with the packed block conv_wino4_pack() makes of its 128 filters -- the blocks follow each other
inline void conv_wino4_pack_groups(const float* w, int cin, int cout, std::vector<float>& out)
{
    out.clear();
    std::vector<float> one;
    for (int g = 0; g < cout / 128; ++g) {
        conv_wino4_pack(w + (size_t)g * 128 * cin * 9, cin, 128, one);
        out.insert(out.end(), one.begin(), one.end());
    }
}

}  // namespace fav
