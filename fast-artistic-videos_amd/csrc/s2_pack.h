// s2_pack.h -- host-side weight packing for the 3x3 stride-2 layers (d64: 32 -> 64, d128: 64 -> 128; models_video.lua:88-92) in the
// fragment order of conv3s2w_kernel (kernels_s2.hip).  Plain C++ (no HIP): the CPU test suite compiles it on its own.
//
// The kernel walks K in chunks of 16 input channels; inside a chunk in two groups of 8 channels (kg), inside a group over the nine
// taps, and one v_mfma_f32_32x32x2_f32 multiplies two input channels (k = half-wave h) of one tap into 32 output channels.  A wave
// (tile nt of 32 output channels) reads its operand of (chunk, kg, tap) as ONE 16-byte load per lane = four MFMA steps st:
//   out[((((chunk * 2 + kg) * 9 + tap) * ntc + nt) * 64 + lane) * 4 + st]     lane = h * 32 + n
//     = w[nt * 32 + n][chunk * 16 + kg * 8 + 4 h + st][tap / 3][tap % 3]
#pragma once
#include <cstddef>
#include <vector>

namespace fav {

inline size_t conv_s2w_packed_floats(int cin, int cout) { return (size_t)(cin / 16) * 2 * 9 * (cout / 32) * 64 * 4; }

// w: [cout][cin][3][3], cin % 16 == 0, cout % 32 == 0
inline void conv_s2w_pack(const float* w, int cin, int cout, std::vector<float>& out)
{
    const int ntc = cout / 32;
    out.assign(conv_s2w_packed_floats(cin, cout), 0.f);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const float* g = w + ((size_t)co * cin + ci) * 9;
            const int chunk = ci >> 4, kg = (ci >> 3) & 1, h = (ci >> 2) & 1, st = ci & 3, nt = co >> 5, n = co & 31;
            for (int tap = 0; tap < 9; ++tap)
                out[(((((size_t)chunk * 2 + kg) * 9 + tap) * ntc + nt) * 64 + h * 32 + n) * 4 + st] = g[tap];
        }
}

// layers with more than 128 filters (cout % 128 == 0): computed in groups of 128 output channels, each with the packed block
// conv_s2w_pack() makes of its 128 filters -- the blocks follow each other
inline void conv_s2w_pack_groups(const float* w, int cin, int cout, std::vector<float>& out)
{
    if (cout <= 128) { conv_s2w_pack(w, cin, cout, out); return; }
    out.clear();
    std::vector<float> one;
    for (int g = 0; g < cout / 128; ++g) {
        conv_s2w_pack(w + (size_t)g * 128 * cin * 9, cin, 128, one);
        out.insert(out.end(), one.begin(), one.end());
    }
}

}  // namespace fav
