// spectral_upsampling.h -- RGB -> spectrum coefficient model of the spectral variant (host side), see the .cpp.
#pragma once
#include <cstdint>
#include <vector>

namespace mtsamd {

struct Rgb2Spec {
    uint32_t res = 0;
    std::vector<float> scale;      // res brightness levels
    std::vector<float> data;       // 3 * res^3 * 3 coefficients
};

void rgb2spec_build(uint32_t res, Rgb2Spec &m, unsigned threads);
bool rgb2spec_save(const char *path, const Rgb2Spec &m);
bool rgb2spec_load(const char *path, Rgb2Spec &m);
void rgb2spec_fetch(const Rgb2Spec &m, const float rgb[3], float out[3]);
void srgb_model_fetch(const Rgb2Spec &m, const float rgb[3], float out[3]);
// srgb_model_mean (include/mitsuba/render/srgb.h:25-35): mean of the model over 16 equidistant wavelengths in [360, 830] nm
float srgb_model_mean(const float coeff[3]);

} // namespace mtsamd
