// spectral_upsampling.cpp -- RGB -> spectrum upsampling model for the spectral variant (host side).
//
// The reference upsamples every RGB reflectance / emission colour to a smooth spectrum
//     S(lambda) = 1/2 + x / (2 sqrt(1 + x^2)),  x = c0 lambda^2 + c1 lambda + c2        (Jakob & Hanika 2019)
// whose coefficients come from a 3 x res^3 table ("data/srgb.coeff", res = 64) that its build generates with
// ext/rgb2spec/rgb2spec_opt.cpp and reads back with rgb2spec_fetch (ext/rgb2spec/rgb2spec.c:81-121,
// src/librender/srgb.cpp:14-40).  The generated file is absent from the reference tree, so this file restates the
// published optimisation (Gauss-Newton in CIE Lab, Simpson 3/8 quadrature of the CIE observer x D65 over 283
// samples, continuation along the brightness axis) and the table lookup; tests/ compare the table against the one
// produced by the reference's own tool compiled from its sources.
#include "spectral_upsampling.h"
#include "cie_data.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>

namespace mtsamd {
namespace {

constexpr int kFine = (95 - 1) * 3 + 1;
constexpr double kLambdaMin = 360.0, kLambdaMax = 830.0;
constexpr double kD65Norm = 10566.864005283874576;       // D65 normalised to unit luminance (ext/rgb2spec/details/cie1931.h:101)

const double kXyzToSrgb[3][3] = { { 3.240479, -1.537150, -0.498535 }, { -0.969256, 1.875991, 0.041556 }, { 0.055648, -0.204043, 1.057311 } };
const double kSrgbToXyz[3][3] = { { 0.412453, 0.357580, 0.180423 }, { 0.212671, 0.715160, 0.072169 }, { 0.019334, 0.119193, 0.950227 } };

struct Quadrature {
    double lambda[kFine], rgb[3][kFine], white[3];
    Quadrature() {
        std::memset(rgb, 0, sizeof(rgb));
        white[0] = white[1] = white[2] = 0.0;
        const double h = (kLambdaMax - kLambdaMin) / (kFine - 1);
        auto interp = [](const double *tbl, double x) {
            x = (x - kLambdaMin) * (94.0 / (kLambdaMax - kLambdaMin));
            int o = std::min(std::max((int) x, 0), 93);
            double w = x - o;
            return (1.0 - w) * tbl[o] + w * tbl[o + 1];
        };
        for (int i = 0; i < kFine; ++i) {
            const double l = kLambdaMin + i * h;
            const double xyz[3] = { interp(kCie_x, l), interp(kCie_y, l), interp(kCie_z, l) };
            const double I = interp(kCie_d65, l) / kD65Norm;
            double weight = 3.0 / 8.0 * h;                 // Simpson's 3/8 rule
            if (i == 0 || i == kFine - 1) { }
            else if ((i - 1) % 3 == 2) weight *= 2.0;
            else weight *= 3.0;
            lambda[i] = l;
            for (int k = 0; k < 3; ++k)
                for (int j = 0; j < 3; ++j) rgb[k][i] += kXyzToSrgb[k][j] * xyz[j] * I * weight;
            for (int k = 0; k < 3; ++k) white[k] += xyz[k] * I * weight;
        }
    }
};

const Quadrature &quadrature() { static Quadrature q; return q; }

void to_lab(const Quadrature &q, double p[3]) {
    double xyz[3] = { 0, 0, 0 };
    for (int k = 0; k < 3; ++k)
        for (int j = 0; j < 3; ++j) xyz[k] += p[j] * kSrgbToXyz[k][j];
    auto f = [](double t) {
        const double delta = 6.0 / 29.0;
        return t > delta * delta * delta ? std::cbrt(t) : t / (delta * delta * 3.0) + 4.0 / 29.0;
    };
    const double fx = f(xyz[0] / q.white[0]), fy = f(xyz[1] / q.white[1]), fz = f(xyz[2] / q.white[2]);
    p[0] = 116.0 * fy - 16.0; p[1] = 500.0 * (fx - fy); p[2] = 200.0 * (fy - fz);
}

void residual(const Quadrature &q, const double c[3], const double rgb[3], double r[3]) {
    double out[3] = { 0, 0, 0 };
    for (int i = 0; i < kFine; ++i) {
        const double l = (q.lambda[i] - kLambdaMin) / (kLambdaMax - kLambdaMin);
        const double x = (c[0] * l + c[1]) * l + c[2];
        const double s = 0.5 * x / std::sqrt(1.0 + x * x) + 0.5;
        for (int j = 0; j < 3; ++j) out[j] += q.rgb[j][i] * s;
    }
    to_lab(q, out);
    double target[3] = { rgb[0], rgb[1], rgb[2] };
    to_lab(q, target);
    for (int j = 0; j < 3; ++j) r[j] = target[j] - out[j];
}

// solve J x = r (3x3, partial pivoting); false if singular
bool solve3(double J[3][3], const double r[3], double x[3]) {
    double a[3][4];
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) a[i][j] = J[i][j]; a[i][3] = r[i]; }
    for (int col = 0; col < 3; ++col) {
        int piv = col;
        for (int i = col + 1; i < 3; ++i) if (std::fabs(a[i][col]) > std::fabs(a[piv][col])) piv = i;
        if (std::fabs(a[piv][col]) < 1e-15) return false;
        if (piv != col) for (int j = 0; j < 4; ++j) std::swap(a[piv][j], a[col][j]);
        for (int i = col + 1; i < 3; ++i) {
            const double f = a[i][col] / a[col][col];
            for (int j = col; j < 4; ++j) a[i][j] -= f * a[col][j];
        }
    }
    for (int i = 2; i >= 0; --i) {
        double s = a[i][3];
        for (int j = i + 1; j < 3; ++j) s -= a[i][j] * x[j];
        x[i] = s / a[i][i];
    }
    return true;
}

void gauss_newton(const Quadrature &q, const double rgb[3], double c[3]) {
    const double eps = 1e-4;
    for (int it = 0; it < 15; ++it) {
        double r[3], J[3][3];
        residual(q, c, rgb, r);
        for (int i = 0; i < 3; ++i) {
            double lo[3] = { c[0], c[1], c[2] }, hi[3] = { c[0], c[1], c[2] }, r0[3], r1[3];
            lo[i] -= eps; hi[i] += eps;
            residual(q, lo, rgb, r0); residual(q, hi, rgb, r1);
            for (int j = 0; j < 3; ++j) J[j][i] = (r1[j] - r0[j]) / (2 * eps);
        }
        double x[3];
        if (!solve3(J, r, x)) return;
        double rr = 0.0;
        for (int j = 0; j < 3; ++j) { c[j] -= x[j]; rr += r[j] * r[j]; }
        const double mx = std::max(std::max(c[0], c[1]), c[2]);
        if (mx > 200) for (int j = 0; j < 3; ++j) c[j] *= 200 / mx;
        if (rr < 1e-6) break;
    }
}

inline double smoothstep(double x) { return x * x * (3.0 - 2.0 * x); }

} // namespace

void rgb2spec_build(uint32_t res, Rgb2Spec &m, unsigned threads) {
    const Quadrature &q = quadrature();
    m.res = res;
    m.scale.resize(res);
    for (uint32_t k = 0; k < res; ++k) m.scale[k] = (float) smoothstep(smoothstep(k / double(res - 1)));
    m.data.assign((size_t) 9 * res * res * res, 0.0f);
    auto store = [&](int l, int k, int j, int i, const double c[3]) {
        // polynomial over [0,1] -> polynomial over wavelengths in nm
        const double c0 = 360.0, c1 = 1.0 / (830.0 - 360.0), A = c[0], B = c[1], C = c[2];
        const size_t idx = (((size_t) l * res + k) * res + j) * res + i;
        m.data[3 * idx + 0] = (float) (A * c1 * c1);
        m.data[3 * idx + 1] = (float) (B * c1 - 2 * A * c0 * c1 * c1);
        m.data[3 * idx + 2] = (float) (C - B * c0 * c1 + A * (c0 * c1) * (c0 * c1));
    };
    std::atomic<uint32_t> next{ 0 };
    auto worker = [&]() {
        for (uint32_t job = next++; job < 3 * res; job = next++) {
            const int l = (int) (job / res), j = (int) (job % res);
            const double y = j / double(res - 1);
            for (uint32_t i = 0; i < res; ++i) {
                const double x = i / double(res - 1);
                const int start = (int) res / 5;
                double c[3] = { 0, 0, 0 }, rgb[3];
                for (int k = start; k < (int) res; ++k) {          // continuation towards brighter colours
                    const double b = (double) m.scale[k];
                    rgb[l] = b; rgb[(l + 1) % 3] = x * b; rgb[(l + 2) % 3] = y * b;
                    gauss_newton(q, rgb, c);
                    store(l, k, j, (int) i, c);
                }
                c[0] = c[1] = c[2] = 0.0;
                for (int k = start; k >= 0; --k) {                 // ... and towards darker ones
                    const double b = (double) m.scale[k];
                    rgb[l] = b; rgb[(l + 1) % 3] = x * b; rgb[(l + 2) % 3] = y * b;
                    gauss_newton(q, rgb, c);
                    store(l, k, j, (int) i, c);
                }
            }
        }
    };
    threads = std::max(1u, threads);
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads; ++t) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
}

bool rgb2spec_save(const char *path, const Rgb2Spec &m) {
    FILE *f = std::fopen(path, "wb");
    if (!f) return false;
    bool ok = std::fwrite("SPEC", 4, 1, f) == 1 && std::fwrite(&m.res, sizeof(uint32_t), 1, f) == 1 &&
              std::fwrite(m.scale.data(), sizeof(float) * m.res, 1, f) == 1 &&
              std::fwrite(m.data.data(), sizeof(float) * m.data.size(), 1, f) == 1;
    std::fclose(f);
    return ok;
}

bool rgb2spec_load(const char *path, Rgb2Spec &m) {
    FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    char hdr[4];
    bool ok = std::fread(hdr, 4, 1, f) == 1 && std::memcmp(hdr, "SPEC", 4) == 0 && std::fread(&m.res, sizeof(uint32_t), 1, f) == 1 &&
              m.res >= 2 && m.res <= 256;
    if (ok) {
        m.scale.resize(m.res);
        m.data.resize((size_t) 9 * m.res * m.res * m.res);
        ok = std::fread(m.scale.data(), sizeof(float) * m.res, 1, f) == 1 && std::fread(m.data.data(), sizeof(float) * m.data.size(), 1, f) == 1;
    }
    std::fclose(f);
    return ok;
}

// trilinear lookup (the published rgb2spec_fetch: largest component selects the table, the other two / brightness index it)
void rgb2spec_fetch(const Rgb2Spec &m, const float rgb_[3], float out[3]) {
    const int res = (int) m.res;
    float rgb[3];
    for (int j = 0; j < 3; ++j) rgb[j] = std::max(std::min(rgb_[j], 1.0f), 0.0f);
    int i = 0;
    for (int j = 1; j < 3; ++j) if (rgb[j] >= rgb[i]) i = j;
    const float z = rgb[i], scale = (res - 1) / z, x = rgb[(i + 1) % 3] * scale, y = rgb[(i + 2) % 3] * scale;
    const uint32_t xi = std::min((uint32_t) x, (uint32_t) (res - 2)), yi = std::min((uint32_t) y, (uint32_t) (res - 2));
    // last interval whose left end is <= z
    int left = 0, last = res - 2, size = last;
    while (size > 0) {
        int half = size >> 1, middle = left + half + 1;
        if (m.scale[middle] <= z) { left = middle; size -= half + 1; } else size = half;
    }
    const uint32_t zi = (uint32_t) std::min(left, last);
    uint32_t offset = (((i * res + zi) * res + yi) * res + xi) * 3;
    const uint32_t dx = 3, dy = 3 * res, dz = 3 * res * res;
    const float x1 = x - xi, x0 = 1.0f - x1, y1 = y - yi, y0 = 1.0f - y1,
                z1 = (z - m.scale[zi]) / (m.scale[zi + 1] - m.scale[zi]), z0 = 1.0f - z1;
    const float *d = m.data.data();
    for (int j = 0; j < 3; ++j) {
        out[j] = ((d[offset] * x0 + d[offset + dx] * x1) * y0 + (d[offset + dy] * x0 + d[offset + dy + dx] * x1) * y1) * z0 +
                 ((d[offset + dz] * x0 + d[offset + dz + dx] * x1) * y0 + (d[offset + dz + dy] * x0 + d[offset + dz + dy + dx] * x1) * y1) * z1;
        ++offset;
    }
}

// srgb_model_fetch (src/librender/srgb.cpp:14-40): pure black / white map to the -inf / +inf sentinels
void srgb_model_fetch(const Rgb2Spec &m, const float rgb[3], float out[3]) {
    if (rgb[0] == 0.0f && rgb[1] == 0.0f && rgb[2] == 0.0f) { out[0] = out[1] = 0.0f; out[2] = -INFINITY; return; }
    if (rgb[0] == 1.0f && rgb[1] == 1.0f && rgb[2] == 1.0f) { out[0] = out[1] = 0.0f; out[2] = INFINITY; return; }
    rgb2spec_fetch(m, rgb, out);
}

float srgb_model_mean(const float c[3]) {
    float sum = 0.0f;
    for (int i = 0; i < 16; ++i) {
        const float l = 360.0f + (float) i * ((830.0f - 360.0f) / 15.0f);
        const float v = std::fma(std::fma(c[0], l, c[1]), l, c[2]);
        float r;
        if (std::isinf(c[2])) r = std::fma(std::copysign(1.0f, c[2]), 0.5f, 0.5f);
        else r = std::fmax(0.0f, std::fma(0.5f * v, 1.0f / std::sqrt(std::fma(v, v, 1.0f)), 0.5f));
        sum += r;
    }
    return sum * (1.0f / 16.0f);
}

} // namespace mtsamd
