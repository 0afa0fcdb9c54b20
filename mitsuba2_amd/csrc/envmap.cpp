// envmap.cpp -- see envmap.h.
#include "envmap.h"

#include <cmath>

namespace mtsamd {
namespace {
inline uint32_t level_index(uint32_t width, uint32_t x, uint32_t y) {          // distr_2d.h:576-580
    return ((x & 1u) | (((x & ~1u) | (y & 1u)) << 1)) + ((y & ~1u) * width);
}
inline uint32_t log2i_ceil(uint32_t v) { uint32_t r = 0; while ((1u << r) < v) ++r; return r; }
}

bool build_envmap(const float *rgb, int width, int height, EnvmapHost &out) {
    if (width < 2 || height < 2) return false;
    const uint32_t w = (uint32_t) width, h = (uint32_t) height;
    out.texels.resize(4 * (size_t) w * h);
    std::vector<float> lum((size_t) w * h);
    for (uint32_t y = 0; y < h; ++y) {
        const float sin_theta = std::sin((float) y / (float) (h - 1) * 3.14159265358979323846f);
        for (uint32_t x = 0; x < w; ++x) {
            const float *p = rgb + 3 * ((size_t) y * w + x);
            float *o = out.texels.data() + 4 * ((size_t) y * w + x);
            o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = 1.0f;
            lum[(size_t) y * w + x] = (p[0] * 0.212671f + p[1] * 0.715160f + p[2] * 0.072169f) * sin_theta;      // luminance * sin(theta)
        }
    }
    const uint32_t npx = w - 1, npy = h - 1;
    out.patch_size[0] = 1.0f / (float) npx; out.patch_size[1] = 1.0f / (float) npy;
    out.inv_patch_size[0] = (float) npx; out.inv_patch_size[1] = (float) npy;
    out.max_patch_index[0] = npx - 1; out.max_patch_index[1] = npy - 1;
    const uint32_t max_level = log2i_ceil(npx > npy ? npx : npy);
    std::vector<uint32_t> size;
    out.lv_offset.assign(1, 0u); out.lv_width.assign(1, w); size.assign(1, w * h);
    uint32_t lx = npx, ly = npy;
    for (int level = (int) max_level; level >= 0; --level) {
        lx += lx & 1u; ly += ly & 1u;
        out.lv_offset.push_back(out.lv_offset.back() + size.back());
        out.lv_width.push_back(lx); size.push_back(lx * ly);
        lx >>= 1; ly >>= 1;
    }
    out.warp.assign((size_t) out.lv_offset.back() + size.back(), 0.0f);
    float *l0 = out.warp.data(), *l1 = out.warp.data() + out.lv_offset[1];
    const float *in = lum.data();
    double sum = 0.0;
    for (uint32_t y = 0; y < npy; ++y) {
        for (uint32_t x = 0; x < npx; ++x) {
            const float avg = (in[0] + in[1] + in[w] + in[w + 1]) * 0.25f;
            sum += (double) avg;
            l1[level_index(out.lv_width[1], x, y)] = avg;
            ++in;
        }
        ++in;
    }
    const float scale = (float) ((double) (npx * npy) / sum);
    for (uint32_t i = 0; i < size[0]; ++i) l0[i] = lum[i] * scale;
    for (uint32_t i = 0; i < size[1]; ++i) l1[i] *= scale;
    lx = npx; ly = npy;
    for (uint32_t level = 2; level <= max_level + 1; ++level) {
        const float *a = out.warp.data() + out.lv_offset[level - 1];
        float *b = out.warp.data() + out.lv_offset[level];
        lx = (lx + 1u) >> 1; ly = (ly + 1u) >> 1;
        for (uint32_t y = 0; y < ly; ++y)
            for (uint32_t x = 0; x < lx; ++x) {
                const float *d0 = a + level_index(out.lv_width[level - 1], x * 2, y * 2);
                b[level_index(out.lv_width[level], x, y)] = d0[0] + d0[1] + d0[2] + d0[3];
            }
    }
    return true;
}

} // namespace mtsamd
