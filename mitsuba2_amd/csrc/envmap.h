// envmap.h -- host-side construction of the `envmap` emitter's sampling hierarchy (Hierarchical2D<Float, 0>,
// include/mitsuba/core/distr_2d.h:200-312) and texel table (src/emitters/envmap.cpp:66-125).
#pragma once
#include <cstdint>
#include <vector>

namespace mtsamd {

struct EnvmapHost {
    std::vector<float> texels;             // RGBA per pixel (alpha = 1)
    std::vector<float> warp;               // all hierarchy levels, concatenated
    std::vector<uint32_t> lv_offset, lv_width;
    float patch_size[2], inv_patch_size[2];
    uint32_t max_patch_index[2];
};

// rgb: height * width * 3 linear RGB.  Returns false if the image is smaller than 2 x 2.
bool build_envmap(const float *rgb, int width, int height, EnvmapHost &out);

} // namespace mtsamd
