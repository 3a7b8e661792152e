// up2_pack.h -- host-side weight merge for a 3x3 convolution (stride 1, zero padding 1) whose input is a x2 nearest-upsampled tensor
// (U2 + c3s1-64, models_video.lua:123-128; kernels_up2.hip).  Plain C++ (no HIP): the CPU test suite compiles it on its own.
//
// On the upsampled image every physical pixel appears 2 x 2 times, so the three rows (columns) of a 3x3 window cover only TWO
// physical rows (columns).  For the output pixel (2u + py, 2v + px), py, px in {0, 1}:
//     rows 2u+py-1 .. 2u+py+1  ->  physical rows u+py-1 (a = 0) and u+py (a = 1)
//         py = 0: ky {0} -> a = 0, ky {1, 2} -> a = 1          py = 1: ky {0, 1} -> a = 0, ky {2} -> a = 1        (same for columns)
//     out[2u+py][2v+px] = sum_{a,b in {0,1}} sum_ci  x[u+py-1+a][v+px-1+b][ci] * Wm[py][px][a][b][ci]
//     Wm[py][px][a][b] = sum_{ky in S(py,a)} sum_{kx in S(px,b)} w[ky][kx]              (summed here in double, rounded once)
// i.e. four 2x2 convolutions on the PHYSICAL image, 4 multiply-adds per output instead of 9; a physical row / column outside
// the image is zero exactly when both upsampled rows / columns it stands for are padding.
//
// Packed order = the order the kernel's waves consume it (wave = phase * 2 + nt: phase = py * 2 + px, nt = tile of 32 output channels):
//   out[((((s * 4 + tp) * 4 + kg) * 8 + wave) * 64 + lane) * 4 + st]      s = 32-channel slice, tp = a * 2 + b, kg = group of 8 channels,
//   lane = h * 32 + n (output channel nt * 32 + n), MFMA step st multiplies input channels s*32 + kg*8 + st (h = 0) and + 4 + st (h = 1)
#pragma once
#include <cstddef>
#include <vector>

namespace fav {

inline size_t conv_up2_packed_floats(int cin) { return (size_t)(cin / 32) * 4 * 4 * 8 * 64 * 4; }

// w: [64][cin][3][3], cin % 32 == 0
inline void conv_up2_pack(const float* w, int cin, std::vector<float>& out)
{
    static const int lo[2][2] = {{0, 1}, {0, 2}}, hi[2][2] = {{0, 2}, {1, 2}};     // S(p, a) = [lo[p][a], hi[p][a]]
    out.assign(conv_up2_packed_floats(cin), 0.f);
    for (int co = 0; co < 64; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const float* g = w + ((size_t)co * cin + ci) * 9;
            const int s = ci >> 5, kg = (ci >> 3) & 3, h = (ci >> 2) & 1, st = ci & 3, nt = co >> 5, n = co & 31;
            for (int py = 0; py < 2; ++py)
                for (int px = 0; px < 2; ++px)
                    for (int a = 0; a < 2; ++a)
                        for (int b = 0; b < 2; ++b) {
                            double sum = 0.0;
                            for (int ky = lo[py][a]; ky <= hi[py][a]; ++ky)
                                for (int kx = lo[px][b]; kx <= hi[px][b]; ++kx) sum += (double)g[ky * 3 + kx];
                            const int wave = (py * 2 + px) * 2 + nt, tp = a * 2 + b, lane = h * 32 + n;
                            out[((((size_t)(s * 4 + tp) * 4 + kg) * 8 + wave) * 64 + lane) * 4 + st] = (float)sum;
                        }
        }
}

// ---- the same layer through F(2x2,3x3) minimal filtering (conv3_up2w_kernel).  On the upsampled image the 4x4 input patch of the
// output block (2u..2u+1, 2v..2v+1) has the rows (u-1, u, u, u+1): B^T d gives  l0 = x[u-1] - x[u],  l1 = 2 x[u],  l2 = 0,
// l3 = x[u] - x[u+1]  (same along the columns) -- the line i = 2 vanishes, 9 of the 16 transform positions remain: 9 multiplies
// per 4 outputs (2.25 per output against 4 for the phase merge above and 9 for the direct form).  With the factor 2 of l1 moved
// into the weights, G g G^T becomes a plain tap sum again:
//     U[i][j] = sum_{a in S_i} sum_{b in S_j} g[a][b],   S_0 = {0}, S_1 = {0, 1, 2}, S_3 = {2}            (i, j in {0, 1, 3})
//     M[i][j] = sum_ci V[i][j] U[i][j],   V = the row / column differences (x[u-1] - x[u], x[u], x[u] - x[u+1]) of the 3x3 neighbourhood
//     Y[0][0] = M00 + M01 + M10 + M11     Y[0][1] = (M01 - M03) + (M11 - M13)
//     Y[1][0] = (M10 + M11) - (M30 + M31) Y[1][1] = (M11 - M13) - (M31 - M33)
// Packed for the kernel's waves (each wave = one row of 32 physical pixels, all 64 output channels, all 9 positions):
//   out[(((kg * 9 + p) * 2 + nt) * 64 + lane) * 4 + st]    kg = group of 8 input channels, p = 3 * ii + jj (ii, jj = 0, 1, 2 for i, j = 0, 1, 3),
//   nt = tile of 32 output channels, lane = h * 32 + n, step st multiplies input channels kg*8 + st (h = 0) and kg*8 + 4 + st (h = 1)
inline size_t conv_up2w_packed_floats(int cin) { return (size_t)(cin / 8) * 9 * 2 * 64 * 4; }

inline void conv_up2w_pack(const float* w, int cin, std::vector<float>& out)
{
    static const int lo[3] = {0, 0, 2}, hi[3] = {0, 2, 2};       // S_0, S_1, S_3
    out.assign(conv_up2w_packed_floats(cin), 0.f);
    for (int co = 0; co < 64; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const float* g = w + ((size_t)co * cin + ci) * 9;
            const int kg = ci >> 3, h = (ci >> 2) & 1, st = ci & 3, nt = co >> 5, n = co & 31;
            for (int ii = 0; ii < 3; ++ii)
                for (int jj = 0; jj < 3; ++jj) {
                    double sum = 0.0;
                    for (int a = lo[ii]; a <= hi[ii]; ++a)
                        for (int b = lo[jj]; b <= hi[jj]; ++b) sum += (double)g[a * 3 + b];
                    out[((((size_t)kg * 9 + ii * 3 + jj) * 2 + nt) * 64 + h * 32 + n) * 4 + st] = (float)sum;
                }
        }
}

// layers with more than 64 filters (cout % 64 == 0): computed in groups of 64 output channels by the nine-position kernel, each
// group with the packed block conv_up2w_pack() makes of its 64 filters -- the blocks follow each other
inline void conv_up2w_pack_groups(const float* w, int cin, int cout, std::vector<float>& out)
{
    out.clear();
    std::vector<float> one;
    for (int g = 0; g < cout / 64; ++g) {
        conv_up2w_pack(w + (size_t)g * 64 * cin * 9, cin, one);
        out.insert(out.end(), one.begin(), one.end());
    }
}

}  // namespace fav
