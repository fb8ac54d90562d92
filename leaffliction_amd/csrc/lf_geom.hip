// libleafhip — geometric augmentation kernels with Pillow (libImaging) semantics.
//
// Coordinates and the bicubic polynomial are evaluated in IEEE double in the same
// operation order as libImaging/Geometry.c; this file is compiled with
// -ffp-contract=off so that no multiply-add is fused (a fused FMA rounds once where C
// rounds twice and would break bit-exactness).  The resampler is pure integer.
#include "lf_common.h"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ int pil_floor(double v) { return v < 0.0 ? (int)floor(v) : (int)v; }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Geometry.c BICUBIC(v, v1, v2, v3, v4, d)
__device__ __forceinline__ double bicubic(double v1, double v2, double v3, double v4, double d) {
    const double p1 = v2;
    const double p2 = -v1 + v3;
    const double p3 = 2 * (v1 - v2) + v3 - v4;
    const double p4 = -v1 + v2 - v3 + v4;
    return p1 + d * (p2 + d * (p3 + d * p4));
}

// One thread per output pixel (3 channels).  grid = (ceil(h*w/256), n).
__global__ __launch_bounds__(kBlock) void warp_bicubic_kernel(const uint8_t* __restrict__ in,
                                                              uint8_t* __restrict__ out,
                                                              const double* __restrict__ coeffs,
                                                              int perspective, int h, int w) {
    const unsigned n = blockIdx.y;
    const double* a = coeffs + (size_t)n * 8;
    const double a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4], a5 = a[5], a6 = a[6],
                 a7 = a[7];
    const uint8_t* src = in + (size_t)n * h * w * 3;
    uint8_t* dst = out + (size_t)n * h * w * 3;
    const int total = h * w;
    for (int t = blockIdx.x * kBlock + threadIdx.x; t < total; t += gridDim.x * kBlock) {
        const int oy = t / w, ox = t - oy * w;
        const double xi = ox + 0.5, yi = oy + 0.5;
        double xin, yin;
        if (perspective) {
            xin = (a0 * xi + a1 * yi + a2) / (a6 * xi + a7 * yi + 1);
            yin = (a3 * xi + a4 * yi + a5) / (a6 * xi + a7 * yi + 1);
        } else {
            xin = a0 * xi + a1 * yi + a2;
            yin = a3 * xi + a4 * yi + a5;
        }
        uint8_t* o = dst + (size_t)t * 3;
        if (xin < 0.0 || xin >= w || yin < 0.0 || yin >= h) {
            o[0] = 0;
            o[1] = 0;
            o[2] = 0;
            continue;
        }
        xin -= 0.5;
        yin -= 0.5;
        int x = pil_floor(xin), y = pil_floor(yin);
        const double dx = xin - x, dy = yin - y;
        x--;
        y--;
        const int x0 = clampi(x, 0, w - 1) * 3, x1 = clampi(x + 1, 0, w - 1) * 3,
                  x2 = clampi(x + 2, 0, w - 1) * 3, x3 = clampi(x + 3, 0, w - 1) * 3;
        const uint8_t* r0 = src + (size_t)clampi(y, 0, h - 1) * w * 3;
        const bool ok1 = (y + 1 >= 0 && y + 1 < h), ok2 = (y + 2 >= 0 && y + 2 < h),
                   ok3 = (y + 3 >= 0 && y + 3 < h);
        const uint8_t* r1 = src + (size_t)(ok1 ? y + 1 : 0) * w * 3;
        const uint8_t* r2 = src + (size_t)(ok2 ? y + 2 : 0) * w * 3;
        const uint8_t* r3 = src + (size_t)(ok3 ? y + 3 : 0) * w * 3;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const double v1 = bicubic(r0[x0 + b], r0[x1 + b], r0[x2 + b], r0[x3 + b], dx);
            double v2 = v1, v3, v4;
            if (ok1) v2 = bicubic(r1[x0 + b], r1[x1 + b], r1[x2 + b], r1[x3 + b], dx);
            v3 = v2;
            if (ok2) v3 = bicubic(r2[x0 + b], r2[x1 + b], r2[x2 + b], r2[x3 + b], dx);
            v4 = v3;
            if (ok3) v4 = bicubic(r3[x0 + b], r3[x1 + b], r3[x2 + b], r3[x3 + b], dx);
            const double v = bicubic(v1, v2, v3, v4, dy);
            o[b] = v <= 0.0 ? 0 : (v >= 255.0 ? 255 : (uint8_t)v);
        }
    }
}

// Geometry.c affine_fixed: xx = a2 + a1*y + a0*x in wrapping int32 16.16.
// Ragged batch: image n writes ohw[n] = (oh, ow) pixels at byte offset out_off[n].
__global__ __launch_bounds__(kBlock) void affine_nearest_kernel(const uint8_t* __restrict__ in,
                                                                uint8_t* __restrict__ out,
                                                                const int32_t* __restrict__ fix6,
                                                                const int32_t* __restrict__ ohw,
                                                                const int64_t* __restrict__ out_off,
                                                                int h, int w, unsigned fill) {
    const unsigned n = blockIdx.y;
    const int32_t* a = fix6 + (size_t)n * 6;
    const unsigned a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4], a5 = a[5];
    const int oh = ohw[2 * n], ow = ohw[2 * n + 1];
    const uint8_t* src = in + (size_t)n * h * w * 3;
    uint8_t* dst = out + out_off[n];
    const int total = oh * ow;
    for (int t = blockIdx.x * kBlock + threadIdx.x; t < total; t += gridDim.x * kBlock) {
        const unsigned oy = t / ow, ox = t - oy * ow;
        // unsigned arithmetic == two's-complement wrap of the C int accumulation
        const int xx = (int)(a2 + a1 * oy + a0 * ox);
        const int yy = (int)(a5 + a4 * oy + a3 * ox);
        const int xin = xx >> 16, yin = yy >> 16;
        uint8_t* o = dst + (size_t)t * 3;
        if (xin >= 0 && xin < w && yin >= 0 && yin < h) {
            const uint8_t* s = src + ((size_t)yin * w + xin) * 3;
            o[0] = s[0];
            o[1] = s[1];
            o[2] = s[2];
        } else {
            o[0] = (uint8_t)fill;
            o[1] = (uint8_t)fill;
            o[2] = (uint8_t)fill;
        }
    }
}

// Resample.c ImagingResampleHorizontal_8bpc / Vertical_8bpc: ss = 1<<21 + sum px*k; clip8(ss>>22).
constexpr int kPrec = 22;

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= kPrec;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal: in [n][h][w][3] -> tmp [n][h][ow][3]; one thread per (y, ox)
__global__ __launch_bounds__(kBlock) void resample_h_kernel(const uint8_t* __restrict__ in,
                                                            uint8_t* __restrict__ tmp, int h, int w,
                                                            int ow, const int32_t* __restrict__ bounds,
                                                            const int32_t* __restrict__ kk, int ks,
                                                            int per_image) {
    const unsigned n = blockIdx.y;
    const int32_t* bnd = bounds + (per_image ? (size_t)n * ow * 2 : 0);
    const int32_t* kx = kk + (per_image ? (size_t)n * ow * ks : 0);
    const uint8_t* src = in + (size_t)n * h * w * 3;
    uint8_t* dst = tmp + (size_t)n * h * ow * 3;
    const int total = h * ow;
    for (int t = blockIdx.x * kBlock + threadIdx.x; t < total; t += gridDim.x * kBlock) {
        const int y = t / ow, ox = t - y * ow;
        // clamp the host-provided window so a bad table can never read out of bounds
        const int xmin = clampi(bnd[2 * ox], 0, w);
        const int cnt = min(bnd[2 * ox + 1], min(ks, w - xmin));
        const int32_t* k = kx + (size_t)ox * ks;
        const uint8_t* p = src + ((size_t)y * w + xmin) * 3;
        int s0 = 1 << (kPrec - 1), s1 = s0, s2 = s0;
        for (int i = 0; i < cnt; ++i) {
            const int c = k[i];
            s0 += p[3 * i] * c;
            s1 += p[3 * i + 1] * c;
            s2 += p[3 * i + 2] * c;
        }
        uint8_t* o = dst + (size_t)t * 3;
        o[0] = clip8(s0);
        o[1] = clip8(s1);
        o[2] = clip8(s2);
    }
}

// vertical: tmp [n][h][ow][3] -> out [n][oh][ow][3]; one thread per (oy, byte column)
__global__ __launch_bounds__(kBlock) void resample_v_kernel(const uint8_t* __restrict__ tmp,
                                                            uint8_t* __restrict__ out, int h, int oh,
                                                            int ow, const int32_t* __restrict__ bounds,
                                                            const int32_t* __restrict__ kk, int ks,
                                                            int per_image) {
    const unsigned n = blockIdx.y;
    const int32_t* bnd = bounds + (per_image ? (size_t)n * oh * 2 : 0);
    const int32_t* ky = kk + (per_image ? (size_t)n * oh * ks : 0);
    const int rowb = ow * 3;
    const uint8_t* src = tmp + (size_t)n * h * rowb;
    uint8_t* dst = out + (size_t)n * oh * rowb;
    const int total = oh * rowb;
    for (int t = blockIdx.x * kBlock + threadIdx.x; t < total; t += gridDim.x * kBlock) {
        const int oy = t / rowb, xb = t - oy * rowb;
        const int ymin = clampi(bnd[2 * oy], 0, h);
        const int cnt = min(bnd[2 * oy + 1], min(ks, h - ymin));
        const int32_t* k = ky + (size_t)oy * ks;
        const uint8_t* p = src + (size_t)ymin * rowb + xb;
        int s = 1 << (kPrec - 1);
        for (int i = 0; i < cnt; ++i) s += p[(size_t)i * rowb] * k[i];
        dst[t] = clip8(s);
    }
}

}  // namespace

extern "C" {

int lf_warp_bicubic_u8(const uint8_t* in, uint8_t* out, const double* coeffs, int perspective,
                       int n, int h, int w, lf_stream_t stream) {
    LF_REQUIRE(in && out && coeffs, "lf_warp_bicubic: null buffer");
    LF_REQUIRE(n > 0 && h > 0 && w > 0, "lf_warp_bicubic: bad dims n=%d h=%d w=%d", n, h, w);
    LF_REQUIRE((size_t)h * w < (1u << 30), "lf_warp_bicubic: image too large");
    LF_REQUIRE(in != out, "lf_warp_bicubic: in-place warp is not supported");
    dim3 grid(lf::stream_grid((size_t)h * w, kBlock, 1024), n);
    warp_bicubic_kernel<<<grid, kBlock, 0, lf::as_stream(stream)>>>(in, out, coeffs, perspective, h,
                                                                    w);
    return lf::check_launch("lf_warp_bicubic");
}

int lf_affine_nearest_fixed_u8(const uint8_t* in, uint8_t* out, const int32_t* fix6,
                               const int32_t* ohw, const int64_t* out_off, int n, int h, int w,
                               int max_out_pixels, int fill, lf_stream_t stream) {
    LF_REQUIRE(in && out && fix6 && ohw && out_off, "lf_affine_nearest_fixed: null buffer");
    LF_REQUIRE(n > 0 && h > 0 && w > 0 && max_out_pixels > 0,
               "lf_affine_nearest_fixed: bad dims n=%d h=%d w=%d max_out_pixels=%d", n, h, w,
               max_out_pixels);
    LF_REQUIRE(h < 32768 && w < 32768, "lf_affine_nearest_fixed: 16.16 fixed point needs sizes < 32768");
    LF_REQUIRE(fill >= 0 && fill <= 255, "lf_affine_nearest_fixed: fill must be 0..255");
    dim3 grid(lf::stream_grid((size_t)max_out_pixels, kBlock, 1024), n);
    affine_nearest_kernel<<<grid, kBlock, 0, lf::as_stream(stream)>>>(in, out, fix6, ohw, out_off, h,
                                                                      w, (unsigned)fill);
    return lf::check_launch("lf_affine_nearest_fixed");
}

int lf_resample_u8(const uint8_t* in, uint8_t* tmp, uint8_t* out, int n, int h, int w, int oh,
                   int ow, const int32_t* xbounds, const int32_t* xk, int kx,
                   const int32_t* ybounds, const int32_t* yk, int ky, int per_image_coeffs,
                   lf_stream_t stream) {
    LF_REQUIRE(in && tmp && out && xbounds && xk && ybounds && yk, "lf_resample: null buffer");
    LF_REQUIRE(n > 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && kx > 0 && ky > 0,
               "lf_resample: bad dims n=%d h=%d w=%d oh=%d ow=%d kx=%d ky=%d", n, h, w, oh, ow, kx,
               ky);
    hipStream_t s = lf::as_stream(stream);
    resample_h_kernel<<<dim3(lf::stream_grid((size_t)h * ow, kBlock, 1024), n), kBlock, 0, s>>>(
        in, tmp, h, w, ow, xbounds, xk, kx, per_image_coeffs);
    resample_v_kernel<<<dim3(lf::stream_grid((size_t)oh * ow * 3, kBlock, 1024), n), kBlock, 0,
                        s>>>(tmp, out, h, oh, ow, ybounds, yk, ky, per_image_coeffs);
    return lf::check_launch("lf_resample");
}

}  // extern "C"
