// libleafhip — the saliency ("blur") filter of srcs/transform/filters/blur.py:18-79 on batches
// of uint8 images resident in HBM.
//
// The filter is a chain of small per-pixel / 3x3-neighbourhood passes over byte and float
// planes (gray, Canny edges, plus-shaped morphology, Sobel magnitude, brown-region mask, colour
// difference against a 15x15 Gaussian, three min-max normalisations, a 5x5 Gaussian, the leaf
// mask).  Every pass is HBM/L2-bound integer or float32 work; nothing here is GEMM-shaped.
// OpenCV semantics (parity unpinned, see oracle/cv_ops.py) are restated step by step, so this
// translation unit is compiled with -ffp-contract=off; the one fused multiply-add OpenCV itself
// uses (convertTo inside cv2.normalize) is written as an explicit fmaf.
#include <float.h>
#include <math.h>

#include <algorithm>
#include <vector>

#include "lf_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kPxPerThread = 8;  // passes that end in a per-image min / max: fewer, fatter workgroups
constexpr int kHystThreads = 1024;
constexpr unsigned kInfBits = 0x7f800000u;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// BORDER_REFLECT_101 for p in [-1, len]
__device__ __forceinline__ int reflect101i(int p, int len) {
    if (len == 1) return 0;
    if (p < 0) return -p;
    if (p >= len) return 2 * len - 2 - p;
    return p;
}

struct Sob {
    int dx, dy;
};

// 3x3 Sobel from the three (already border-mapped) rows / columns.
__device__ __forceinline__ Sob sobel_at(const uint8_t* g, int w, int y0, int y1, int y2, int x0, int x1,
                                        int x2) {
    const uint8_t* r0 = g + (size_t)y0 * w;
    const uint8_t* r1 = g + (size_t)y1 * w;
    const uint8_t* r2 = g + (size_t)y2 * w;
    const int a = r0[x0], b = r0[x1], c = r0[x2];
    const int d = r1[x0], f = r1[x2];
    const int k = r2[x0], l = r2[x1], m = r2[x2];
    Sob s;
    s.dx = (c + 2 * f + m) - (a + 2 * d + k);
    s.dy = (k + 2 * l + m) - (a + 2 * b + c);
    return s;
}

// Per-image min / max of non-negative floats through their bit patterns (monotone as uint):
// wave shuffle, then the four waves through LDS, one atomic pair per workgroup.
struct MinMax {
    unsigned lo = kInfBits, hi = 0u;
    __device__ __forceinline__ void take(float v) {
        lo = min(lo, __float_as_uint(v));
        hi = max(hi, __float_as_uint(v));
    }
};

__device__ __forceinline__ void minmax_publish(MinMax r, unsigned* mn, unsigned* mx) {
    __shared__ unsigned wlo[kBlock / 64], whi[kBlock / 64];
    unsigned lo = r.lo, hi = r.hi;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, (unsigned)__shfl_xor((int)lo, off, 64));
        hi = max(hi, (unsigned)__shfl_xor((int)hi, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        wlo[threadIdx.x >> 6] = lo;
        whi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 1; k < kBlock / 64; ++k) {
            lo = min(lo, wlo[k]);
            hi = max(hi, whi[k]);
        }
        atomicMin(mn, lo);
        atomicMax(mx, hi);
    }
}

// cv2.normalize(.., 0, 255, NORM_MINMAX): scale / shift in double, applied as float32 fma.
__device__ __forceinline__ void norm_coeffs(unsigned mn_bits, unsigned mx_bits, float& a, float& b) {
    const double smin = (double)__uint_as_float(mn_bits), smax = (double)__uint_as_float(mx_bits);
    const double span = smax - smin;
    const double scale = 255.0 * (span > DBL_EPSILON ? 1.0 / span : 0.0);
    const double shift = 0.0 - smin * scale;
    a = (float)scale;
    b = (float)shift;
}

__device__ __forceinline__ uint8_t trunc_u8(float v) {  // numpy float32 -> uint8 astype, in range
    const int i = (int)v;
    return (uint8_t)(i < 0 ? 0 : (i > 255 ? 255 : i));
}

__global__ void minmax_init_kernel(unsigned* mm, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n * 6) mm[i] = (i & 1) ? 0u : kInfBits;  // (min, max) x {gradient, colour diff, saliency}
}

// Sobel of the gray plane: dx^2 + dy^2 and (dx, dy) with BORDER_REPLICATE for Canny
// (canny.cpp), sqrt(dx^2 + dy^2) with BORDER_REFLECT_101 for cv2.Sobel + cv2.magnitude.
__global__ __launch_bounds__(kBlock) void sal_sobel_kernel(const uint8_t* __restrict__ gray,
                                                           int32_t* __restrict__ mag2,
                                                           uint32_t* __restrict__ dxdy,
                                                           float* __restrict__ gmag,
                                                           unsigned* __restrict__ mm, int h, int w) {
    const unsigned n = blockIdx.y;
    const int hw = h * w;
    const uint8_t* g = gray + (size_t)n * hw;
    MinMax mmx;
    for (int k = 0; k < kPxPerThread; ++k) {
        const int p = (blockIdx.x * kPxPerThread + k) * kBlock + threadIdx.x;
        if (p >= hw) break;
        const int y = p / w, x = p - y * w;
        const Sob s = sobel_at(g, w, clampi(y - 1, 0, h - 1), y, clampi(y + 1, 0, h - 1),
                               clampi(x - 1, 0, w - 1), x, clampi(x + 1, 0, w - 1));
        mag2[(size_t)n * hw + p] = __mul24(s.dx, s.dx) + __mul24(s.dy, s.dy);  // |d| <= 1020
        dxdy[(size_t)n * hw + p] = ((unsigned)s.dx & 0xffffu) | ((unsigned)s.dy << 16);
        Sob r = s;
        if (x == 0 || y == 0 || x == w - 1 || y == h - 1)
            r = sobel_at(g, w, reflect101i(y - 1, h), y, reflect101i(y + 1, h), reflect101i(x - 1, w), x,
                         reflect101i(x + 1, w));
        const float gm = __fsqrt_rn((float)(__mul24(r.dx, r.dx) + __mul24(r.dy, r.dy)));  // < 2^24: exact
        gmag[(size_t)n * hw + p] = gm;
        mmx.take(gm);
    }
    minmax_publish(mmx, mm + n * 6 + 0, mm + n * 6 + 1);
}

// Non-maximum suppression + double threshold: map = 1 (no edge), 0 (weak), 2 (strong).
__global__ __launch_bounds__(kBlock) void canny_nms_kernel(const int32_t* __restrict__ mag2,
                                                           const uint32_t* __restrict__ dxdy,
                                                           uint8_t* __restrict__ map, int h, int w,
                                                           int low, int high) {
    const unsigned n = blockIdx.y;
    const int hw = h * w;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= hw) return;
    const int32_t* mg = mag2 + (size_t)n * hw;
    const int y = p / w, x = p - y * w;
    auto at = [&](int yy, int xx) -> int {  // the magnitude buffer has a zero frame
        return (yy < 0 || yy >= h || xx < 0 || xx >= w) ? 0 : mg[yy * w + xx];
    };
    const int m = mg[p];
    uint8_t out = 1;
    if (m > low) {
        const unsigned pk = dxdy[(size_t)n * hw + p];
        const int xs = (int)(short)(pk & 0xffffu), ys = (int)(short)(pk >> 16);
        const int ax = xs < 0 ? -xs : xs;
        const int ay = (ys < 0 ? -ys : ys) << 15;
        const int tg22x = __mul24(ax, 13573);  // tan(22.5 deg) in 15-bit fixed point
        bool keep;
        if (ay < tg22x) {
            keep = m > at(y, x - 1) && m >= at(y, x + 1);
        } else {
            const int tg67x = tg22x + (ax << 16);
            if (ay > tg67x) {
                keep = m > at(y - 1, x) && m >= at(y + 1, x);
            } else {
                const int s = (xs ^ ys) < 0 ? 1 : -1;
                keep = m > at(y - 1, x - s) && m > at(y + 1, x + s);
            }
        }
        if (keep) out = m > high ? 2 : 0;
    }
    map[(size_t)n * hw + p] = out;
}

// Hysteresis: one workgroup per image sweeps the map until no weak pixel next to a strong one
// is left; the map lives in LDS when it fits.  Ends with the edge image (0 / 255) in place.
__global__ __launch_bounds__(kHystThreads) void canny_hysteresis_kernel(uint8_t* __restrict__ map,
                                                                        int h, int w, int in_lds) {
    extern __shared__ uint8_t lds_map[];
    __shared__ int changed;
    const int hw = h * w;
    uint8_t* gm = map + (size_t)blockIdx.x * hw;
    uint8_t* m = gm;
    if (in_lds) {
        for (int p = threadIdx.x; p < hw; p += kHystThreads) lds_map[p] = gm[p];
        m = lds_map;
    }
    do {
        __syncthreads();
        if (threadIdx.x == 0) changed = 0;
        __syncthreads();
        bool any = false;
        for (int p = threadIdx.x; p < hw; p += kHystThreads) {
            if (m[p] != 0) continue;
            const int y = p / w, x = p - y * w;
            bool strong = false;
            for (int dy = -1; dy <= 1; ++dy) {
                const int yy = y + dy;
                if (yy < 0 || yy >= h) continue;
                for (int dx = -1; dx <= 1; ++dx) {
                    const int xx = x + dx;
                    if (xx < 0 || xx >= w) continue;
                    strong = strong || m[yy * w + xx] == 2;
                }
            }
            if (strong) {
                m[p] = 2;
                any = true;
            }
        }
        if (any) changed = 1;
        __syncthreads();
    } while (changed);
    for (int p = threadIdx.x; p < hw; p += kHystThreads) gm[p] = m[p] == 2 ? 255 : 0;
}

// dilate / erode with the 3x3 MORPH_ELLIPSE element (a plus); outside pixels never win.
template <bool ERODE>
__global__ __launch_bounds__(kBlock) void morph_cross_kernel(const uint8_t* __restrict__ in,
                                                             uint8_t* __restrict__ out, int h, int w) {
    const unsigned n = blockIdx.y;
    const int hw = h * w;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= hw) return;
    const uint8_t* s = in + (size_t)n * hw;
    const int y = p / w, x = p - y * w;
    unsigned v = s[p];
    auto take = [&](unsigned t) { v = ERODE ? min(v, t) : max(v, t); };
    if (y > 0) take(s[p - w]);
    if (y < h - 1) take(s[p + w]);
    if (x > 0) take(s[p - 1]);
    if (x < w - 1) take(s[p + 1]);
    out[(size_t)n * hw + p] = (uint8_t)v;
}

// The planes this filter dilates / erodes hold 0 or 255 only, so max / min are OR / AND and four
// pixels go through as one dword (w % 4 == 0): the left / right neighbours are byte funnels
// with the adjacent dwords.
template <bool ERODE>
__global__ __launch_bounds__(kBlock) void morph_cross_bin4_kernel(const uint32_t* __restrict__ in,
                                                                  uint32_t* __restrict__ out, int h,
                                                                  int w4) {
    const unsigned n = blockIdx.y;
    const int total = h * w4;
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= total) return;
    const uint32_t* s = in + (size_t)n * total;
    const int y = t / w4, g = t - y * w4;
    const unsigned ident = ERODE ? 0xffffffffu : 0u;
    const unsigned c = s[t];
    const unsigned up = y > 0 ? s[t - w4] : ident, dn = y < h - 1 ? s[t + w4] : ident;
    const unsigned pv = g > 0 ? s[t - 1] : ident, nx = g < w4 - 1 ? s[t + 1] : ident;
    const unsigned left = (c << 8) | (pv >> 24), right = (c >> 8) | (nx << 24);
    out[(size_t)n * total + t] = ERODE ? (c & up & dn & left & right) : (c | up | dn | left | right);
}

// brown_regions of blur.py:47-53 as a 0 / 255 plane (OpenCV 8-bit RGB2HSV, H in [0,180)).
__global__ __launch_bounds__(kBlock) void brown_mask_kernel(const uint8_t* __restrict__ rgb,
                                                            const uint8_t* __restrict__ leaf,
                                                            uint8_t* __restrict__ out, size_t npx,
                                                            int hue_lo, int hue_hi, int s_min, int v_max) {
    __shared__ int sdiv[256], hdiv[256];
    for (int i = threadIdx.x; i < 256; i += kBlock) {
        sdiv[i] = i ? __double2int_rn(__ddiv_rn(1044480.0, (double)i)) : 0;
        hdiv[i] = i ? __double2int_rn(__ddiv_rn(737280.0, __dmul_rn(6.0, (double)i))) : 0;
    }
    __syncthreads();
    for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < npx; p += (size_t)gridDim.x * kBlock) {
        const int r = rgb[3 * p], g = rgb[3 * p + 1], b = rgb[3 * p + 2];
        const int v = max(r, max(g, b)), vmin = min(r, min(g, b)), diff = v - vmin;
        const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
        const int s = (__mul24(diff, sdiv[v]) + (1 << 11)) >> 12;
        int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
        hh = (__mul24(hh, hdiv[diff]) + (1 << 11)) >> 12;
        hh += hh < 0 ? 180 : 0;
        const bool brown = hh >= hue_lo && hh <= hue_hi && s >= s_min && v <= v_max && leaf[p] > 0;
        out[p] = brown ? 255 : 0;
    }
}

// mean over channels of |rgb - blurred| as numpy float32 computes it, + per-image min / max.
__global__ __launch_bounds__(kBlock) void color_diff_kernel(const uint8_t* __restrict__ rgb,
                                                            const uint8_t* __restrict__ blurred,
                                                            float* __restrict__ cdiff,
                                                            unsigned* __restrict__ mm, int hw) {
    const unsigned n = blockIdx.y;
    MinMax mmx;
    for (int k = 0; k < kPxPerThread; ++k) {
        const int p = (blockIdx.x * kPxPerThread + k) * kBlock + threadIdx.x;
        if (p >= hw) break;
        const size_t o = ((size_t)n * hw + p) * 3;
        int acc = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int d = (int)rgb[o + c] - (int)blurred[o + c];
            acc += d < 0 ? -d : d;
        }
        const float v = __fdiv_rn((float)acc, 3.0f);
        cdiff[(size_t)n * hw + p] = v;
        mmx.take(v);
    }
    minmax_publish(mmx, mm + n * 6 + 2, mm + n * 6 + 3);
}

// saliency = 0.4 edges + 0.3 uint8(norm(gradient)) + 0.6 brown + 0.2 norm(colour diff), each
// product and sum rounded to float32 in blur.py's order; + per-image min / max.
__global__ __launch_bounds__(kBlock) void saliency_kernel(const uint8_t* __restrict__ edges,
                                                          const float* __restrict__ gmag,
                                                          const uint8_t* __restrict__ brown,
                                                          const float* __restrict__ cdiff,
                                                          float* __restrict__ sal,
                                                          unsigned* __restrict__ mm, int hw) {
    const unsigned n = blockIdx.y;
    __shared__ float coef[4];
    if (threadIdx.x == 0) {
        norm_coeffs(mm[n * 6 + 0], mm[n * 6 + 1], coef[0], coef[1]);
        norm_coeffs(mm[n * 6 + 2], mm[n * 6 + 3], coef[2], coef[3]);
    }
    __syncthreads();
    const float ga = coef[0], gb = coef[1], ca = coef[2], cb = coef[3];
    MinMax mmx;
    for (int k = 0; k < kPxPerThread; ++k) {
        const int p = (blockIdx.x * kPxPerThread + k) * kBlock + threadIdx.x;
        if (p >= hw) break;
        const size_t i = (size_t)n * hw + p;
        float s = (float)edges[i] * 0.4f;
        s = s + (float)trunc_u8(__fmaf_rn(gmag[i], ga, gb)) * 0.3f;
        if (brown) s = s + (float)brown[i] * 0.6f;
        s = s + __fmaf_rn(cdiff[i], ca, cb) * 0.2f;
        sal[i] = s;
        mmx.take(s);
    }
    minmax_publish(mmx, mm + n * 6 + 4, mm + n * 6 + 5);
}

__global__ __launch_bounds__(kBlock) void saliency_norm_kernel(const float* __restrict__ sal,
                                                               const unsigned* __restrict__ mm,
                                                               uint8_t* __restrict__ out, int hw) {
    const unsigned n = blockIdx.y;
    __shared__ float coef[2];
    if (threadIdx.x == 0) norm_coeffs(mm[n * 6 + 4], mm[n * 6 + 5], coef[0], coef[1]);
    __syncthreads();
    const float a = coef[0], b = coef[1];
    for (int k = 0; k < kPxPerThread; ++k) {
        const int p = (blockIdx.x * kPxPerThread + k) * kBlock + threadIdx.x;
        if (p >= hw) break;
        out[(size_t)n * hw + p] = trunc_u8(__fmaf_rn(sal[(size_t)n * hw + p], a, b));
    }
}

// result[leaf] = blurred saliency, elsewhere 0, replicated to three channels.
__global__ __launch_bounds__(kBlock) void saliency_out_kernel(const uint8_t* __restrict__ sal,
                                                              const uint8_t* __restrict__ leaf,
                                                              uint8_t* __restrict__ out, size_t npx) {
    for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < npx; p += (size_t)gridDim.x * kBlock) {
        const uint8_t v = leaf[p] > 0 ? sal[p] : 0;
        out[3 * p] = v;
        out[3 * p + 1] = v;
        out[3 * p + 2] = v;
    }
}

// ===========================================================================
// _create_inclusive_mask (srcs/transform/filters/mask.py:727-831): the default strategy of make_mask.
// Per-pixel colour predicates (8-bit HSV and L*a*b*, uint8 channel comparisons that wrap as numpy's
// do), Canny (L1 gradient, 30 / 100) dilated by the 3x3 ellipse, the texture test against a 15x15
// Gaussian of the gray plane -> one BIT per pixel; then, one workgroup per image with the bit planes
// in LDS: open 3x3, close 9x9, close 7x7 (cv2's MORPH_ELLIPSE elements), largest 8-connected
// component (run-length union-find), close 5x5.
// ===========================================================================
__global__ __launch_bounds__(kBlock) void canny_sobel_l1_kernel(const uint8_t* __restrict__ gray,
                                                                int32_t* __restrict__ mag,
                                                                uint32_t* __restrict__ dxdy, int h, int w) {
    const unsigned n = blockIdx.y;
    const int hw = h * w;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= hw) return;
    const int y = p / w, x = p - y * w;
    const Sob s = sobel_at(gray + (size_t)n * hw, w, clampi(y - 1, 0, h - 1), y, clampi(y + 1, 0, h - 1),
                           clampi(x - 1, 0, w - 1), x, clampi(x + 1, 0, w - 1));
    mag[(size_t)n * hw + p] = (s.dx < 0 ? -s.dx : s.dx) + (s.dy < 0 ? -s.dy : s.dy);
    dxdy[(size_t)n * hw + p] = ((unsigned)s.dx & 0xffffu) | ((unsigned)s.dy << 16);
}

constexpr int kLabCbrtSize = 256 * 3 / 2 * 8;   // LAB_CBRT_TAB_SIZE_B

// One wave per 64-pixel segment of a row; bit i of the ballot is pixel x0 + i.
// bits[n][y][2 * seg + {0, 1}]: `wpr` = 2 * ceil(w / 64) words per row.
__global__ __launch_bounds__(kBlock) void inclusive_pred_kernel(
    const uint8_t* __restrict__ rgb, const uint8_t* __restrict__ gray, const uint8_t* __restrict__ blur,
    const uint8_t* __restrict__ edges, const uint16_t* __restrict__ lab_tabs, uint32_t* __restrict__ bits,
    int n_images, int h, int w, int hue_lo, int hue_hi) {
    __shared__ int sdiv[256], hdiv[256];
    __shared__ uint16_t gam[256], cbr[kLabCbrtSize];
    for (int i = threadIdx.x; i < 256; i += kBlock) {
        sdiv[i] = i ? __double2int_rn(__ddiv_rn(1044480.0, (double)i)) : 0;
        hdiv[i] = i ? __double2int_rn(__ddiv_rn(737280.0, __dmul_rn(6.0, (double)i))) : 0;
        gam[i] = lab_tabs[i];
    }
    for (int i = threadIdx.x; i < kLabCbrtSize; i += kBlock) cbr[i] = lab_tabs[256 + i];
    __syncthreads();
    const int spr = (w + 63) / 64, wpr = 2 * spr;
    const long total = (long)n_images * h * spr;
    const int lane = threadIdx.x & 63;
    for (long seg = (long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); seg < total;
         seg += (long)gridDim.x * (kBlock / 64)) {
        const int sx = (int)(seg % spr);
        const long ry = seg / spr;
        const int y = (int)(ry % h);
        const size_t n = (size_t)(ry / h);
        const int x = sx * 64 + lane;
        bool plant = false;
        if (x < w) {
            const size_t p = (n * h + y) * (size_t)w + x;
            const int r = rgb[3 * p], g = rgb[3 * p + 1], b = rgb[3 * p + 2];
            // 8-bit HSV (color_hsv: H in [0, 180))
            const int v = max(r, max(g, b)), vmin = min(r, min(g, b)), diff = v - vmin;
            const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
            const int s = (__mul24(diff, sdiv[v]) + (1 << 11)) >> 12;
            int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
            hh = (__mul24(hh, hdiv[diff]) + (1 << 11)) >> 12;
            hh += hh < 0 ? 180 : 0;
            // 8-bit L*a*b* (color_lab RGB2Lab_b)
            const int R = gam[r], G = gam[g], B = gam[b];
            const int fx = cbr[(R * 1777 + G * 1541 + B * 778 + 2048) >> 12];
            const int fy = cbr[(R * 871 + G * 2929 + B * 296 + 2048) >> 12];
            const int fz = cbr[(R * 73 + G * 448 + B * 3575 + 2048) >> 12];
            const int L = clampi((296 * fy - 1336934 + 16384) >> 15, 0, 255);
            const int la = clampi((500 * (fx - fy) + 4194304 + 16384) >> 15, 0, 255);
            const int lb = clampi((200 * (fy - fz) + 4194304 + 16384) >> 15, 0, 255);
            const bool strong_green = hh >= hue_lo && hh <= hue_hi && s >= 30 && v >= 30;
            // uint8 planes: r + 15 wraps (mask.py:759-763)
            const bool dominant = g > ((r + 15) & 255) || g > ((b + 15) & 255) ||
                                  (g > ((r + 5) & 255) && g > ((b + 5) & 255) && s >= 20);
            const bool lab_green = la <= 125 && lb >= 120 && L >= 20 && L <= 240;
            const uint8_t* e = edges + (n * h) * (size_t)w;
            const int q = y * w + x;
            const bool edge = e[q] || (x > 0 && e[q - 1]) || (x < w - 1 && e[q + 1]) || (y > 0 && e[q - w]) ||
                              (y < h - 1 && e[q + w]);
            const int tex = (int)gray[p] - (int)blur[p];
            const bool background = (s <= 25 && v >= 50 && v <= 220) ||
                                    (hh >= 120 && hh <= 160 && s >= 20 && r > g && b > g) ||
                                    (s <= 15 && (tex < 0 ? -tex : tex) < 10);
            plant = (strong_green || dominant || lab_green || edge) && !background;
        }
        const unsigned long long m = __ballot(plant);
        if (lane == 0) {
            uint32_t* o = bits + ((n * h + y) * (size_t)wpr + 2 * sx);
            o[0] = (unsigned)m;
            o[1] = (unsigned)(m >> 32);
        }
    }
}

// ---- binary morphology on bit rows in LDS
template <int K>
struct EllipseRows;   // half-width of each row of cv2.getStructuringElement(MORPH_ELLIPSE, (K, K))
template <>
struct EllipseRows<3> {
    static constexpr int dx[3] = {0, 1, 0};
};
template <>
struct EllipseRows<5> {
    static constexpr int dx[5] = {0, 2, 2, 2, 0};
};
template <>
struct EllipseRows<7> {
    static constexpr int dx[7] = {0, 2, 3, 3, 3, 2, 0};
};
template <>
struct EllipseRows<9> {
    static constexpr int dx[9] = {0, 3, 3, 4, 4, 4, 3, 3, 0};
};

// dst = dilate(src) or erode(src) by the K x K ellipse; pixels outside the image never win, i.e. an
// erosion is the dilation of the complement taken INSIDE the image.  Bits past column w stay 0.
template <int K, bool ERODE>
__device__ void morph_bits(const unsigned* __restrict__ src, unsigned* __restrict__ dst, int h, int w, int wpr) {
    constexpr int R = K / 2;
    const int used = (w + 31) / 32;
    const unsigned lastmask = (w & 31) ? ((1u << (w & 31)) - 1u) : 0xffffffffu;
    auto valid = [&](int xw) -> unsigned { return xw < used - 1 ? 0xffffffffu : (xw == used - 1 ? lastmask : 0u); };
    auto word = [&](int y, int xw) -> unsigned {
        if (xw < 0 || xw >= wpr) return 0u;
        const unsigned v = src[y * wpr + xw];
        return ERODE ? ~v & valid(xw) : v;
    };
    for (int i = threadIdx.x; i < h * wpr; i += blockDim.x) {
        const int y = i / wpr, xw = i - y * wpr;
        unsigned out = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int yy = y + k - R;
            if (yy < 0 || yy >= h) continue;
            const unsigned cur = word(yy, xw), prev = word(yy, xw - 1), next = word(yy, xw + 1);
            out |= cur;
#pragma unroll
            for (int s = 1; s <= EllipseRows<K>::dx[k]; ++s)
                out |= (cur << s) | (prev >> (32 - s)) | (cur >> s) | (next << (32 - s));
        }
        dst[i] = (ERODE ? ~out : out) & valid(xw);
    }
    __syncthreads();
}

__device__ __forceinline__ int uf_load(int* p, int x) {
    return __hip_atomic_load(p + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // past the L1: atomics update L2 only
}
__device__ int uf_find(int* p, int x) {
    for (;;) {
        const int px = uf_load(p, x);
        if (px == x) return x;
        x = px;
    }
}
__device__ void uf_union(int* p, int a, int b) {
    for (;;) {
        a = uf_find(p, a);
        b = uf_find(p, b);
        if (a == b) return;
        if (a > b) {
            const int t = a;
            a = b;
            b = t;
        }
        const int old = atomicMin(p + b, a);   // roots only ever move to a smaller index
        if (old == b) return;
        b = old;
    }
}

// next run of ones in a bit row at or after column x: [start, end] inclusive; false when none
__device__ bool next_run(const unsigned* row, int w, int x, int& start, int& end) {
    while (x < w) {
        const unsigned wd = row[x >> 5] >> (x & 31);
        if (wd == 0u) {
            x = (x | 31) + 1;
            continue;
        }
        x += __builtin_ctz(wd);
        break;
    }
    if (x >= w) return false;
    start = x;
    while (x < w) {
        const unsigned wd = ~(row[x >> 5] >> (x & 31));   // zeros above the shifted-in part end the run too
        const int room = 32 - (x & 31);
        const int ones = wd == 0u ? 32 : __builtin_ctz(wd);
        if (ones < room) {
            x += ones;
            break;
        }
        x += room;
    }
    end = (x > w ? w : x) - 1;
    return true;
}

struct MaskRun {
    unsigned short start, end;
};

__global__ __launch_bounds__(kBlock) void inclusive_morph_kernel(const uint32_t* __restrict__ bits,
                                                                 uint8_t* __restrict__ out, MaskRun* __restrict__ runs,
                                                                 int* __restrict__ parent, int* __restrict__ area,
                                                                 int h, int w, int wpr, int runs_per_image) {
    extern __shared__ unsigned lds_bits[];
    unsigned* A = lds_bits;
    unsigned* B = A + h * wpr;
    int* rowstart = reinterpret_cast<int*>(B + h * wpr);   // [h + 1]
    __shared__ unsigned long long best;
    const size_t n = blockIdx.x;
    for (int i = threadIdx.x; i < h * wpr; i += kBlock) A[i] = bits[n * h * wpr + i];
    __syncthreads();
    morph_bits<3, true>(A, B, h, w, wpr);    // MORPH_OPEN 3x3
    morph_bits<3, false>(B, A, h, w, wpr);
    morph_bits<9, false>(A, B, h, w, wpr);   // MORPH_CLOSE 9x9
    morph_bits<9, true>(B, A, h, w, wpr);
    morph_bits<7, false>(A, B, h, w, wpr);   // MORPH_CLOSE 7x7
    morph_bits<7, true>(B, A, h, w, wpr);

    // ---- largest 8-connected component of A -> B
    MaskRun* rn = runs + n * runs_per_image;
    int* par = parent + n * runs_per_image;
    int* ar = area + n * runs_per_image;
    for (int y = threadIdx.x; y < h; y += kBlock) {
        int c = 0, x = 0, s, e;
        while (next_run(A + y * wpr, w, x, s, e)) {
            ++c;
            x = e + 1;
        }
        rowstart[y + 1] = c;
    }
    if (threadIdx.x == 0) {
        rowstart[0] = 0;
        best = 0ull;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int y = 0; y < h; ++y) rowstart[y + 1] += rowstart[y];
    __syncthreads();
    const int nruns = rowstart[h];
    for (int y = threadIdx.x; y < h; y += kBlock) {
        int k = rowstart[y], x = 0, s, e;
        while (next_run(A + y * wpr, w, x, s, e)) {
            rn[k] = MaskRun{(unsigned short)s, (unsigned short)e};
            par[k] = k;
            ar[k] = 0;
            ++k;
            x = e + 1;
        }
    }
    __threadfence();
    __syncthreads();
    for (int y = 1 + threadIdx.x; y < h; y += kBlock) {   // runs of row y against runs of row y - 1
        int a = rowstart[y - 1], b = rowstart[y];
        const int a_end = rowstart[y], b_end = rowstart[y + 1];
        while (a < a_end && b < b_end) {
            const MaskRun ra = rn[a], rb = rn[b];
            if ((int)ra.start <= (int)rb.end + 1 && (int)ra.end >= (int)rb.start - 1) uf_union(par, a, b);
            if (ra.end < rb.end) ++a;
            else ++b;
        }
    }
    __threadfence();
    __syncthreads();
    for (int k = threadIdx.x; k < nruns; k += kBlock) {
        const int root = uf_find(par, k);
        atomicAdd(ar + root, (int)rn[k].end - (int)rn[k].start + 1);
    }
    __threadfence();
    __syncthreads();
    for (int k = threadIdx.x; k < nruns; k += kBlock) {   // largest area; the earliest component among equals
        const int a = __hip_atomic_load(ar + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a > 0) atomicMax(&best, ((unsigned long long)a << 32) | (unsigned)(0x7fffffff - k));
    }
    for (int i = threadIdx.x; i < h * wpr; i += kBlock) B[i] = 0u;
    __syncthreads();
    if (nruns > 0) {
        const int keep = 0x7fffffff - (int)(best & 0xffffffffull);
        for (int y = threadIdx.x; y < h; y += kBlock)
            for (int k = rowstart[y]; k < rowstart[y + 1]; ++k) {
                if (uf_find(par, k) != keep) continue;
                for (int x = rn[k].start; x <= (int)rn[k].end;) {   // this thread owns the row's words
                    const int bit = x & 31, len = min(32 - bit, (int)rn[k].end - x + 1);
                    B[y * wpr + (x >> 5)] |= (len == 32 ? 0xffffffffu : ((1u << len) - 1u)) << bit;
                    x += len;
                }
            }
    }
    __syncthreads();
    morph_bits<5, false>(B, A, h, w, wpr);   // MORPH_CLOSE 5x5
    morph_bits<5, true>(A, B, h, w, wpr);
    uint8_t* o = out + n * (size_t)h * w;
    for (int p = threadIdx.x; p < h * w; p += kBlock) {
        const int y = p / w, x = p - y * w;
        o[p] = (B[y * wpr + (x >> 5)] >> (x & 31)) & 1u ? 255 : 0;
    }
}

constexpr size_t kAlign = 256;
inline size_t up(size_t v) { return (v + kAlign - 1) & ~(kAlign - 1); }

// ===========================================================================
// Round 3: the per-image middle of both filters in ONE workgroup per image, planes resident in LDS.
//
// Both chains above spend their time in launches and in round trips of small planes through L2 / HBM (13 launches
// and 20 bytes of intermediates per pixel for the saliency filter).  A 224 x 224 gray plane is 50 KB: the gray
// plane, the Canny map and the bit planes of the brown regions of one image fit the 160 KB of a CU together, and
// everything the old kernels kept in 4-byte planes (squared gradient, (dx, dy), gradient magnitude, colour
// difference, saliency) is cheaper to RECOMPUTE from the gray plane in LDS / the two RGB images in L2 than to store.
// What stays outside: the two Gaussian blurs (the i8-MFMA kernel of lf_blur_mfma.hip) and the final masking pass.
// The arithmetic is the old kernels', expression for expression (the bit-exact tests are unchanged).
// ===========================================================================
constexpr int kFuseT = 1024;

__device__ __forceinline__ unsigned gray_px_f(int r, int g, int b) {   // cv2 RGB2GRAY, 14-bit fixed point (lf_augment.hip)
    return (unsigned)(r * 4899 + g * 9617 + b * 1868 + 8192) >> 14;
}

// Canny's non-maximum suppression + double threshold on a gray plane in LDS: map = 1 (no edge), 0 (weak),
// 2 (strong).  L1: |dx| + |dy| (cv2.Canny default), else dx^2 + dy^2 against squared thresholds.  The gradient of a
// neighbour is recomputed from the plane (8 LDS bytes) instead of being read from a 4-byte plane in memory.
template <bool L1>
__device__ __forceinline__ void canny_nms_lds(const uint8_t* gray, uint8_t* emap, int h, int w, int low, int high) {
    const int hw = h * w;
    auto sob = [&](int y, int x) -> Sob {
        return sobel_at(gray, w, clampi(y - 1, 0, h - 1), y, clampi(y + 1, 0, h - 1), clampi(x - 1, 0, w - 1), x,
                        clampi(x + 1, 0, w - 1));
    };
    auto mag = [&](const Sob s) -> int {
        return L1 ? (s.dx < 0 ? -s.dx : s.dx) + (s.dy < 0 ? -s.dy : s.dy) : __mul24(s.dx, s.dx) + __mul24(s.dy, s.dy);
    };
    auto at = [&](int yy, int xx) -> int {  // the magnitude buffer has a zero frame
        return (yy < 0 || yy >= h || xx < 0 || xx >= w) ? 0 : mag(sob(yy, xx));
    };
    const float inv_w = 1.0f / (float)w;
    for (int p = threadIdx.x; p < hw; p += kFuseT) {
        int y = (int)((float)p * inv_w);
        int x = p - __mul24(y, w);
        if (x < 0) { --y; x += w; } else if (x >= w) { ++y; x -= w; }
        const Sob s = sob(y, x);
        const int m = mag(s);
        uint8_t out = 1;
        if (m > low) {
            const int xs = s.dx, ys = s.dy;
            const int ax = xs < 0 ? -xs : xs;
            const int ay = (ys < 0 ? -ys : ys) << 15;
            const int tg22x = __mul24(ax, 13573);  // tan(22.5 deg) in 15-bit fixed point
            bool keep;
            if (ay < tg22x) {
                keep = m > at(y, x - 1) && m >= at(y, x + 1);
            } else {
                const int tg67x = tg22x + (ax << 16);
                if (ay > tg67x) {
                    keep = m > at(y - 1, x) && m >= at(y + 1, x);
                } else {
                    const int sg = (xs ^ ys) < 0 ? 1 : -1;
                    keep = m > at(y - 1, x - sg) && m > at(y + 1, x + sg);
                }
            }
            if (keep) out = m > high ? 2 : 0;
        }
        emap[p] = out;
    }
    __syncthreads();
}

// Hysteresis on the map in LDS (1 = no edge, 0 = weak, 2 = strong; 2 = edge afterwards): a weak pixel with a strong
// 8-neighbour becomes strong, until nothing changes.  On bit planes — strong bits S, weak bits W, one 32-pixel word
// per thread and sweep: S |= W & dilate3x3(S) — instead of one pixel per thread with nine byte reads: a sweep over a
// 224 x 224 map is 1,792 words, and noisy images need dozens of sweeps (the byte sweeps were most of the fused
// kernels' time: 0.4 ms per image).  Rows are `wpr` words wide (two per 64-pixel segment, as the ballots deliver them).
__device__ __forceinline__ void canny_hysteresis_lds(uint8_t* m, unsigned* sb, unsigned* wb, int h, int w, int wpr,
                                                     int* changed) {
    const int spr = wpr / 2;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int seg = wv; seg < h * spr; seg += kFuseT / 64) {
        const int y = seg / spr, sx = seg - y * spr, x = sx * 64 + lane;
        const uint8_t v = x < w ? m[y * w + x] : 1;
        const unsigned long long s = __ballot(v == 2), wk = __ballot(v == 0);
        if (lane == 0) {
            sb[y * wpr + 2 * sx] = (unsigned)s;
            sb[y * wpr + 2 * sx + 1] = (unsigned)(s >> 32);
            wb[y * wpr + 2 * sx] = (unsigned)wk;
            wb[y * wpr + 2 * sx + 1] = (unsigned)(wk >> 32);
        }
    }
    do {
        __syncthreads();
        if (threadIdx.x == 0) *changed = 0;
        __syncthreads();
        bool any = false;
        for (int i = threadIdx.x; i < h * wpr; i += kFuseT) {
            const unsigned weak = wb[i] & ~sb[i];
            if (weak == 0u) continue;
            const int y = i / wpr, xw = i - y * wpr;
            unsigned nb = 0;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) {
                const int yy = y + dy;
                if (yy < 0 || yy >= h) continue;
                const unsigned* row = sb + yy * wpr;
                const unsigned c = row[xw], pv = xw > 0 ? row[xw - 1] : 0u, nx = xw < wpr - 1 ? row[xw + 1] : 0u;
                nb |= c | (c << 1) | (pv >> 31) | (c >> 1) | (nx << 31);
            }
            const unsigned grow = weak & nb;   // (bits past column w are never weak)
            if (grow) {
                sb[i] |= grow;                 // in place: growth is monotone, the fixed point is the same
                any = true;
            }
        }
        if (any) *changed = 1;
        __syncthreads();
    } while (*changed);
    for (int seg = wv; seg < h * spr; seg += kFuseT / 64) {
        const int y = seg / spr, sx = seg - y * spr, x = sx * 64 + lane;
        if (x < w) m[y * w + x] = (sb[y * wpr + (x >> 5)] >> (x & 31) & 1u) ? 2 : 1;
    }
    __syncthreads();
}

// min / max of non-negative floats over the workgroup (bit patterns, as MinMax does), result in lohi[0..1]
__device__ __forceinline__ void minmax_block(MinMax r, unsigned* wscratch, unsigned* lohi) {
    unsigned lo = r.lo, hi = r.hi;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, (unsigned)__shfl_xor((int)lo, off, 64));
        hi = max(hi, (unsigned)__shfl_xor((int)hi, off, 64));
    }
    __syncthreads();   // wscratch may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) {
        wscratch[2 * (threadIdx.x >> 6)] = lo;
        wscratch[2 * (threadIdx.x >> 6) + 1] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < kFuseT / 64; ++k) {
            lo = min(lo, wscratch[2 * k]);
            hi = max(hi, wscratch[2 * k + 1]);
        }
        lohi[0] = lo;
        lohi[1] = hi;
    }
    __syncthreads();
}

// One workgroup per image: gray plane, Canny (L2 gradient, 50 / 150), brown regions (closed, dilated twice), the
// three normalisations and the weighted sum of blur.py:30-66 -> the normalised saliency plane (uint8) in `nsal`.
// rgb / blurred: [n][h][w][3]; leaf: [n][h][w]; w % 4 == 0.
__global__ __launch_bounds__(kFuseT) void saliency_fused_kernel(const uint8_t* __restrict__ rgb,
                                                                const uint8_t* __restrict__ blurred,
                                                                const uint8_t* __restrict__ leaf,
                                                                uint8_t* __restrict__ nsal, int h, int w, int use_brown,
                                                                int hue_lo, int hue_hi, int s_min, int v_max) {
    extern __shared__ __attribute__((aligned(16))) uint8_t fl[];
    __shared__ int sdiv[256], hdiv[256];
    __shared__ unsigned wscratch[2 * (kFuseT / 64)];
    __shared__ unsigned mm[6];
    __shared__ float coef[6];
    __shared__ int changed;
    const int hw = h * w, plane = (hw + 15) & ~15;
    const int spr = (w + 63) / 64, wpr = 2 * spr;      // 64-pixel segments / 32-bit words per bit row
    uint8_t* gray = fl;
    uint8_t* emap = fl + plane;
    unsigned* ba = reinterpret_cast<unsigned*>(fl + 2 * plane);
    unsigned* bb = ba + h * wpr;
    unsigned* hs = bb + h * wpr;   // hysteresis: strong / weak bit planes (bb is free until the brown morphology)
    const size_t n = blockIdx.x;
    const uint8_t* src = rgb + n * (size_t)hw * 3;
    const uint8_t* blr = blurred + n * (size_t)hw * 3;
    const uint8_t* lf_ = leaf + n * (size_t)hw;
    for (int i = threadIdx.x; i < 256; i += kFuseT) {
        sdiv[i] = i ? __double2int_rn(__ddiv_rn(1044480.0, (double)i)) : 0;
        hdiv[i] = i ? __double2int_rn(__ddiv_rn(737280.0, __dmul_rn(6.0, (double)i))) : 0;
    }
    __syncthreads();
    // ---- pass 1: gray plane; brown_regions of blur.py:47-53 as one bit per pixel (a wave = 64 pixels of a row)
    {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        for (int seg = wv; seg < h * spr; seg += kFuseT / 64) {
            const int y = seg / spr, sx = seg - y * spr, x = sx * 64 + lane;
            bool brown = false;
            if (x < w) {
                const int p = y * w + x;
                const int r = src[3 * p], g = src[3 * p + 1], b = src[3 * p + 2];
                gray[p] = (uint8_t)gray_px_f(r, g, b);
                if (use_brown) {
                    const int v = max(r, max(g, b)), vmin = min(r, min(g, b)), diff = v - vmin;
                    const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
                    const int s = (__mul24(diff, sdiv[v]) + (1 << 11)) >> 12;
                    int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
                    hh = (__mul24(hh, hdiv[diff]) + (1 << 11)) >> 12;
                    hh += hh < 0 ? 180 : 0;
                    brown = hh >= hue_lo && hh <= hue_hi && s >= s_min && v <= v_max && lf_[p] > 0;
                }
            }
            const unsigned long long m = __ballot(brown);
            if (lane == 0) {
                ba[y * wpr + 2 * sx] = (unsigned)m;
                ba[y * wpr + 2 * sx + 1] = (unsigned)(m >> 32);
            }
        }
    }
    __syncthreads();
    // ---- Canny(gray, 50, 150, L2gradient=True): thresholds compared squared; edges = (emap == 2)
    canny_nms_lds<false>(gray, emap, h, w, 50 * 50, 150 * 150);
    canny_hysteresis_lds(emap, hs, bb, h, w, wpr, &changed);
    // ---- brown: MORPH_CLOSE with the 3x3 ellipse (a plus), then dilate twice
    if (use_brown) {
        morph_bits<3, false>(ba, bb, h, w, wpr);
        morph_bits<3, true>(bb, ba, h, w, wpr);
        morph_bits<3, false>(ba, bb, h, w, wpr);
        morph_bits<3, false>(bb, ba, h, w, wpr);
    }
    const float inv_w = 1.0f / (float)w;
    auto rowcol = [&](int p, int& y, int& x) {
        y = (int)((float)p * inv_w);
        x = p - __mul24(y, w);
        if (x < 0) { --y; x += w; } else if (x >= w) { ++y; x -= w; }
    };
    // cv2.Sobel + cv2.magnitude (BORDER_REFLECT_101), float32 sqrt of an exact integer
    auto gmag_at = [&](int y, int x) -> float {
        const Sob r = sobel_at(gray, w, reflect101i(y - 1, h), y, reflect101i(y + 1, h), reflect101i(x - 1, w), x,
                               reflect101i(x + 1, w));
        return __fsqrt_rn((float)(__mul24(r.dx, r.dx) + __mul24(r.dy, r.dy)));
    };
    // ---- pass 2: min / max of the gradient magnitude
    {
        MinMax r;
        for (int p = threadIdx.x; p < hw; p += kFuseT) {
            int y, x;
            rowcol(p, y, x);
            r.take(gmag_at(y, x));
        }
        minmax_block(r, wscratch, mm + 0);
    }
    // mean over channels of |rgb - blurred| as numpy float32 computes it: four pixels (three dwords of each image)
    auto cdiff4 = [&](int q, float* out) {
        const uint32_t* a = reinterpret_cast<const uint32_t*>(src) + 3 * q;
        const uint32_t* b = reinterpret_cast<const uint32_t*>(blr) + 3 * q;
        const unsigned av[3] = {a[0], a[1], a[2]}, bv[3] = {b[0], b[1], b[2]};
        int acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int d = (int)((av[j >> 2] >> (8 * (j & 3))) & 0xffu) - (int)((bv[j >> 2] >> (8 * (j & 3))) & 0xffu);
            acc[j / 3] += d < 0 ? -d : d;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = __fdiv_rn((float)acc[k], 3.0f);
    };
    // ---- pass 3: min / max of the colour difference
    {
        MinMax r;
        for (int q = threadIdx.x; q < hw / 4; q += kFuseT) {
            float v[4];
            cdiff4(q, v);
#pragma unroll
            for (int k = 0; k < 4; ++k) r.take(v[k]);
        }
        minmax_block(r, wscratch, mm + 2);
    }
    if (threadIdx.x == 0) {
        norm_coeffs(mm[0], mm[1], coef[0], coef[1]);
        norm_coeffs(mm[2], mm[3], coef[2], coef[3]);
    }
    __syncthreads();
    // saliency = 0.4 dilate(edges) + 0.3 uint8(norm(gradient)) + 0.6 brown + 0.2 norm(colour diff): each product and
    // sum rounded to float32 in blur.py's order (saliency_kernel above)
    const float ga = coef[0], gb = coef[1], ca = coef[2], cb = coef[3];
    auto sal4 = [&](int q, float* out) {
        float cd[4];
        cdiff4(q, cd);
        int y, x;
        rowcol(4 * q, y, x);   // w % 4 == 0: the four pixels share a row
#pragma unroll
        for (int k = 0; k < 4; ++k, ++x) {
            const int p = 4 * q + k;
            const bool e = emap[p] == 2 || (x > 0 && emap[p - 1] == 2) || (x < w - 1 && emap[p + 1] == 2) ||
                           (y > 0 && emap[p - w] == 2) || (y < h - 1 && emap[p + w] == 2);
            float sv = (e ? 255.0f : 0.0f) * 0.4f;
            sv = sv + (float)trunc_u8(__fmaf_rn(gmag_at(y, x), ga, gb)) * 0.3f;
            if (use_brown) sv = sv + ((ba[y * wpr + (x >> 5)] >> (x & 31) & 1u) ? 255.0f : 0.0f) * 0.6f;
            sv = sv + __fmaf_rn(cd[k], ca, cb) * 0.2f;
            out[k] = sv;
        }
    };
    // ---- pass 4: min / max of the saliency
    {
        MinMax r;
        for (int q = threadIdx.x; q < hw / 4; q += kFuseT) {
            float v[4];
            sal4(q, v);
#pragma unroll
            for (int k = 0; k < 4; ++k) r.take(v[k]);
        }
        minmax_block(r, wscratch, mm + 4);
    }
    if (threadIdx.x == 0) norm_coeffs(mm[4], mm[5], coef[4], coef[5]);
    __syncthreads();
    // ---- pass 5: the normalised saliency plane
    const float na = coef[4], nb = coef[5];
    uint32_t* dst = reinterpret_cast<uint32_t*>(nsal + n * (size_t)hw);
    for (int q = threadIdx.x; q < hw / 4; q += kFuseT) {
        float v[4];
        sal4(q, v);
        unsigned o = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) o |= (unsigned)trunc_u8(__fmaf_rn(v[k], na, nb)) << (8 * k);
        dst[q] = o;
    }
}

// gray plane (from memory) -> Canny (L1 gradient, 30 / 100) in LDS -> the per-pixel predicates of
// _create_inclusive_mask (inclusive_pred_kernel above, expression for expression) -> one bit per pixel.
__global__ __launch_bounds__(kFuseT) void inclusive_fused_kernel(
    const uint8_t* __restrict__ rgb, const uint8_t* __restrict__ gray_g, const uint8_t* __restrict__ blur,
    const uint16_t* __restrict__ lab_tabs, uint32_t* __restrict__ bits, int h, int w, int hue_lo, int hue_hi) {
    extern __shared__ __attribute__((aligned(16))) uint8_t fl[];
    __shared__ int sdiv[256], hdiv[256];
    __shared__ uint16_t gam[256], cbr[kLabCbrtSize];
    __shared__ int changed;
    const int hw = h * w, plane = (hw + 15) & ~15;
    uint8_t* gray = fl;
    uint8_t* emap = fl + plane;
    const size_t n = blockIdx.x;
    for (int i = threadIdx.x; i < 256; i += kFuseT) {
        sdiv[i] = i ? __double2int_rn(__ddiv_rn(1044480.0, (double)i)) : 0;
        hdiv[i] = i ? __double2int_rn(__ddiv_rn(737280.0, __dmul_rn(6.0, (double)i))) : 0;
        gam[i] = lab_tabs[i];
    }
    for (int i = threadIdx.x; i < kLabCbrtSize; i += kFuseT) cbr[i] = lab_tabs[256 + i];
    {
        const uint32_t* g4 = reinterpret_cast<const uint32_t*>(gray_g + n * (size_t)hw);   // hw % 4 == 0
        for (int q = threadIdx.x; q < hw / 4; q += kFuseT) reinterpret_cast<uint32_t*>(gray)[q] = g4[q];
    }
    __syncthreads();
    const int spr = (w + 63) / 64, wpr = 2 * spr;
    canny_nms_lds<true>(gray, emap, h, w, 30, 100);   // cv2.Canny(gray, 30, 100)
    {
        unsigned* hs = reinterpret_cast<unsigned*>(fl + 2 * plane);
        canny_hysteresis_lds(emap, hs, hs + h * wpr, h, w, wpr, &changed);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint8_t* src = rgb + n * (size_t)hw * 3;
    const uint8_t* bl = blur + n * (size_t)hw;
    for (int seg = wv; seg < h * spr; seg += kFuseT / 64) {
        const int y = seg / spr, sx = seg - y * spr, x = sx * 64 + lane;
        bool plant = false;
        if (x < w) {
            const int q = y * w + x;
            const int r = src[3 * q], g = src[3 * q + 1], b = src[3 * q + 2];
            const int v = max(r, max(g, b)), vmin = min(r, min(g, b)), diff = v - vmin;
            const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
            const int s = (__mul24(diff, sdiv[v]) + (1 << 11)) >> 12;
            int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
            hh = (__mul24(hh, hdiv[diff]) + (1 << 11)) >> 12;
            hh += hh < 0 ? 180 : 0;
            const int R = gam[r], G = gam[g], B = gam[b];
            const int fx = cbr[(R * 1777 + G * 1541 + B * 778 + 2048) >> 12];
            const int fy = cbr[(R * 871 + G * 2929 + B * 296 + 2048) >> 12];
            const int fz = cbr[(R * 73 + G * 448 + B * 3575 + 2048) >> 12];
            const int L = clampi((296 * fy - 1336934 + 16384) >> 15, 0, 255);
            const int la = clampi((500 * (fx - fy) + 4194304 + 16384) >> 15, 0, 255);
            const int lb = clampi((200 * (fy - fz) + 4194304 + 16384) >> 15, 0, 255);
            const bool strong_green = hh >= hue_lo && hh <= hue_hi && s >= 30 && v >= 30;
            const bool dominant = g > ((r + 15) & 255) || g > ((b + 15) & 255) ||
                                  (g > ((r + 5) & 255) && g > ((b + 5) & 255) && s >= 20);
            const bool lab_green = la <= 125 && lb >= 120 && L >= 20 && L <= 240;
            const bool edge = emap[q] == 2 || (x > 0 && emap[q - 1] == 2) || (x < w - 1 && emap[q + 1] == 2) ||
                              (y > 0 && emap[q - w] == 2) || (y < h - 1 && emap[q + w] == 2);
            const int tex = (int)gray[q] - (int)bl[q];
            const bool background = (s <= 25 && v >= 50 && v <= 220) ||
                                    (hh >= 120 && hh <= 160 && s >= 20 && r > g && b > g) ||
                                    (s <= 15 && (tex < 0 ? -tex : tex) < 10);
            plant = (strong_green || dominant || lab_green || edge) && !background;
        }
        const unsigned long long m = __ballot(plant);
        if (lane == 0) {
            uint32_t* o = bits + ((n * h + y) * (size_t)wpr + 2 * sx);
            o[0] = (unsigned)m;
            o[1] = (unsigned)(m >> 32);
        }
    }
}

// dynamic LDS of the two kernels above, or 0 when an image does not fit a CU (the multi-launch chains then run)
static size_t fused_lds_bytes(int h, int w, bool with_bits) {
    if (w % 4 != 0 || h < 1) return 0;
    const size_t plane = ((size_t)h * w + 15) & ~(size_t)15;
    const size_t wpr = 2 * (size_t)((w + 63) / 64);
    const size_t need = 2 * plane + (with_bits ? 3 : 2) * (size_t)h * wpr * 4;   // + strong / weak (/ brown) bit planes
    return need <= (size_t)140 * 1024 ? need : 0;
}

}  // namespace

extern "C" {

size_t lf_blur_saliency_workspace(int n, int h, int w) {
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    const size_t px = (size_t)n * h * w;
    // 4 byte planes, the blurred RGB copy, 3 four-byte planes, the min/max table
    return 4 * up(px) + up(3 * px) + 3 * up(4 * px) + up((size_t)n * 6 * sizeof(unsigned));
}

int lf_blur_saliency_u8(const uint8_t* rgb, const uint8_t* leaf_mask, uint8_t* out, int n, int h, int w,
                        int use_brown, int hue_lo, int hue_hi, int s_min, int v_max,
                        const uint16_t* kq15, const uint16_t* kq5, void* workspace, size_t ws_bytes,
                        lf_stream_t stream) {
    LF_REQUIRE(rgb && leaf_mask && out && kq15 && kq5 && workspace, "lf_blur_saliency: null buffer");
    LF_REQUIRE(n > 0 && h > 0 && w > 0, "lf_blur_saliency: bad dims n=%d h=%d w=%d", n, h, w);
    LF_REQUIRE(n <= 65535, "lf_blur_saliency: batch too large for grid.y");
    LF_REQUIRE((size_t)h * w < ((size_t)1 << 30), "lf_blur_saliency: image too large");
    LF_REQUIRE(ws_bytes >= lf_blur_saliency_workspace(n, h, w),
               "lf_blur_saliency: workspace too small (%zu < %zu)", ws_bytes,
               lf_blur_saliency_workspace(n, h, w));
    LF_REQUIRE((reinterpret_cast<size_t>(workspace) & 15) == 0, "lf_blur_saliency: workspace must be 16-byte aligned");
    hipStream_t s = lf::as_stream(stream);
    const int hw = h * w;
    const size_t px = (size_t)n * hw;
    uint8_t* base = static_cast<uint8_t*>(workspace);
    uint8_t* pa = base;                      // gray -> brown -> closed -> normalised saliency
    uint8_t* pb = pa + up(px);               // Canny map / edges -> morphology scratch -> blurred saliency
    uint8_t* pc = pb + up(px);               // dilated edges
    uint8_t* pd = pc + up(px);               // dilated brown regions
    uint8_t* blurred = pd + up(px);
    int32_t* mag2 = reinterpret_cast<int32_t*>(blurred + up(3 * px));
    uint32_t* dxdy = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(mag2) + up(4 * px));
    float* gmag = reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(dxdy) + up(4 * px));
    unsigned* mm = reinterpret_cast<unsigned*>(reinterpret_cast<uint8_t*>(gmag) + up(4 * px));
    float* cdiff = reinterpret_cast<float*>(mag2);  // mag2 is dead after the NMS pass
    float* sal = reinterpret_cast<float*>(dxdy);    // so is (dx, dy)

    const dim3 grid_px((hw + kBlock - 1) / kBlock, n);
    const dim3 grid_fat((hw + kBlock * kPxPerThread - 1) / (kBlock * kPxPerThread), n);
    int rc;
    // images that fit a CU (two byte planes + the brown bit planes in LDS; 224 x 224 does): the 15 x 15 blur, ONE
    // workgroup per image for everything up to the normalised saliency plane, the 5 x 5 blur, the masking pass
    if (const size_t fl = fused_lds_bytes(h, w, true)) {
        static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(saliency_fused_kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024) == hipSuccess;
        if (ok) {
            rc = lf_gauss_blur_u8(rgb, blurred, n, h, w, 3, kq15, 15, stream);
            if (rc != LF_OK) return rc;
            saliency_fused_kernel<<<n, kFuseT, fl, s>>>(rgb, blurred, leaf_mask, pa, h, w, use_brown, hue_lo, hue_hi,
                                                        s_min, v_max);
            rc = lf_gauss_blur_u8(pa, pb, n, h, w, 1, kq5, 5, stream);
            if (rc != LF_OK) return rc;
            saliency_out_kernel<<<lf::stream_grid(px / 4 + 1, kBlock, lf::kFullGrid), kBlock, 0, s>>>(pb, leaf_mask,
                                                                                                   out, px);
            return lf::check_launch("lf_blur_saliency");
        }
    }
    minmax_init_kernel<<<(n * 6 + 255) / 256, 256, 0, s>>>(mm, n);
    rc = lf_rgb2gray_u8(rgb, pa, px, stream);
    if (rc != LF_OK) return rc;
    sal_sobel_kernel<<<grid_fat, kBlock, 0, s>>>(pa, mag2, dxdy, gmag, mm, h, w);
    // cv2.Canny(gray, 50, 150, L2gradient=True): thresholds are compared squared
    canny_nms_kernel<<<grid_px, kBlock, 0, s>>>(mag2, dxdy, pb, h, w, 50 * 50, 150 * 150);
    {
        static const size_t lds_cap = []() {
            const size_t want = 156 * 1024;
            return hipFuncSetAttribute(reinterpret_cast<const void*>(canny_hysteresis_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)want) == hipSuccess
                       ? want
                       : (size_t)60 * 1024;
        }();
        const int in_lds = (size_t)hw <= lds_cap;
        canny_hysteresis_kernel<<<n, kHystThreads, in_lds ? (size_t)((hw + 15) & ~15) : 0, s>>>(pb, h, w,
                                                                                                 in_lds);
    }
    // binary planes, 16-byte aligned, so rows of w % 4 == 0 pixels go four at a time
    const bool quad = w % 4 == 0;
    const dim3 grid_q((hw / 4 + kBlock - 1) / kBlock, n);
    auto morph = [&](const uint8_t* src, uint8_t* dst, bool erode) {
        if (quad) {
            const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src);
            uint32_t* d4 = reinterpret_cast<uint32_t*>(dst);
            if (erode)
                morph_cross_bin4_kernel<true><<<grid_q, kBlock, 0, s>>>(s4, d4, h, w / 4);
            else
                morph_cross_bin4_kernel<false><<<grid_q, kBlock, 0, s>>>(s4, d4, h, w / 4);
        } else if (erode) {
            morph_cross_kernel<true><<<grid_px, kBlock, 0, s>>>(src, dst, h, w);
        } else {
            morph_cross_kernel<false><<<grid_px, kBlock, 0, s>>>(src, dst, h, w);
        }
    };
    morph(pb, pc, false);
    if (use_brown) {
        brown_mask_kernel<<<lf::stream_grid(px / 4 + 1, kBlock, lf::kFullGrid), kBlock, 0, s>>>(
            rgb, leaf_mask, pa, px, hue_lo, hue_hi, s_min, v_max);
        morph(pa, pb, false);  // MORPH_CLOSE
        morph(pb, pa, true);
        morph(pa, pb, false);  // dilate, iterations=2
        morph(pb, pd, false);
    }
    rc = lf_gauss_blur_u8(rgb, blurred, n, h, w, 3, kq15, 15, stream);
    if (rc != LF_OK) return rc;
    color_diff_kernel<<<grid_fat, kBlock, 0, s>>>(rgb, blurred, cdiff, mm, hw);
    saliency_kernel<<<grid_fat, kBlock, 0, s>>>(pc, gmag, use_brown ? pd : nullptr, cdiff, sal, mm, hw);
    saliency_norm_kernel<<<grid_fat, kBlock, 0, s>>>(sal, mm, pa, hw);
    rc = lf_gauss_blur_u8(pa, pb, n, h, w, 1, kq5, 5, stream);
    if (rc != LF_OK) return rc;
    saliency_out_kernel<<<lf::stream_grid(px / 4 + 1, kBlock, lf::kFullGrid), kBlock, 0, s>>>(pb, leaf_mask, out,
                                                                                           px);
    return lf::check_launch("lf_blur_saliency");
}

static void lab_tables_host(uint16_t* out) {   // color_lab.cpp initLabTabs: sRGBGammaTab_b, LabCbrtTab_b
    for (int i = 0; i < 256; ++i) {
        const float x = (float)i / 255.0f;
        const double xd = (double)x;
        const float lin = (float)(xd <= 0.04045 ? xd / 12.92 : pow((xd + 0.055) / 1.055, 2.4));
        const long v = lrint((double)(2040.0f * lin));
        out[i] = (uint16_t)(v < 0 ? 0 : (v > 65535 ? 65535 : v));
    }
    const float scale = 1.0f / (255.0f * 8.0f);
    const float lthresh = 216.0f / 24389.0f, lscale = 841.0f / 108.0f, lbias = 16.0f / 116.0f;
    for (int i = 0; i < kLabCbrtSize; ++i) {
        const float y = scale * (float)i;
        float f;
        if (y < lthresh) {
            const float prod = y * lscale;   // two roundings, as numpy's float32 arithmetic in the oracle
            f = prod + lbias;
        } else {
            f = (float)cbrt((double)y);
        }
        const long v = lrint((double)(32768.0f * f));
        out[256 + i] = (uint16_t)(v < 0 ? 0 : (v > 65535 ? 65535 : v));
    }
}

static size_t mask_runs_per_image(int h, int w) { return (size_t)h * (w / 2 + 1); }

size_t lf_inclusive_mask_workspace(int n, int h, int w) {
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    const size_t px = (size_t)n * h * w;
    const size_t wpr = 2 * (size_t)((w + 63) / 64);
    const size_t runs = (size_t)n * mask_runs_per_image(h, w);
    // gray, blurred gray, Canny map; |dx| + |dy| and (dx, dy); the bit planes; runs, parents, areas; the two tables
    return 3 * up(px) + 2 * up(4 * px) + up((size_t)n * h * wpr * 4) + up(runs * sizeof(MaskRun)) + 2 * up(runs * 4) +
           up((256 + kLabCbrtSize) * sizeof(uint16_t));
}

int lf_inclusive_mask_u8(const uint8_t* rgb, uint8_t* mask, int n, int h, int w, int green_lo, int green_hi,
                         const uint16_t* kq15, void* workspace, size_t ws_bytes, lf_stream_t stream) {
    LF_REQUIRE(rgb && mask && kq15 && workspace, "lf_inclusive_mask: null buffer");
    LF_REQUIRE(n > 0 && h > 0 && w > 0, "lf_inclusive_mask: bad dims n=%d h=%d w=%d", n, h, w);
    LF_REQUIRE(n <= 65535, "lf_inclusive_mask: batch too large for grid.y");
    LF_REQUIRE(w <= 65535 && h <= 65535, "lf_inclusive_mask: image too large (%d x %d)", h, w);
    LF_REQUIRE(ws_bytes >= lf_inclusive_mask_workspace(n, h, w), "lf_inclusive_mask: workspace too small (%zu < %zu)",
               ws_bytes, lf_inclusive_mask_workspace(n, h, w));
    LF_REQUIRE((reinterpret_cast<size_t>(workspace) & 15) == 0, "lf_inclusive_mask: workspace must be 16-byte aligned");
    const int wpr = 2 * ((w + 63) / 64);
    const size_t lds = (size_t)2 * h * wpr * 4 + (size_t)(h + 1) * 4;
    static const size_t lds_cap = []() {
        const size_t want = 150 * 1024;
        return hipFuncSetAttribute(reinterpret_cast<const void*>(inclusive_morph_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)want) == hipSuccess
                   ? want
                   : (size_t)60 * 1024;
    }();
    LF_REQUIRE(lds <= lds_cap, "lf_inclusive_mask: a %d x %d image needs %zu bytes of LDS for its bit planes (limit %zu)",
               h, w, lds, lds_cap);
    hipStream_t s = lf::as_stream(stream);
    const int hw = h * w;
    const size_t px = (size_t)n * hw;
    const size_t runs = (size_t)n * mask_runs_per_image(h, w);
    uint8_t* base = static_cast<uint8_t*>(workspace);
    uint8_t* gray = base;
    uint8_t* blur = gray + up(px);
    uint8_t* map = blur + up(px);
    int32_t* mag = reinterpret_cast<int32_t*>(map + up(px));
    uint32_t* dxdy = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(mag) + up(4 * px));
    uint32_t* bits = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(dxdy) + up(4 * px));
    MaskRun* rn = reinterpret_cast<MaskRun*>(reinterpret_cast<uint8_t*>(bits) + up((size_t)n * h * wpr * 4));
    int* parent = reinterpret_cast<int*>(reinterpret_cast<uint8_t*>(rn) + up(runs * sizeof(MaskRun)));
    int* area = reinterpret_cast<int*>(reinterpret_cast<uint8_t*>(parent) + up(runs * 4));
    uint16_t* tabs = reinterpret_cast<uint16_t*>(reinterpret_cast<uint8_t*>(area) + up(runs * 4));

    static const std::vector<uint16_t> host_tabs = []() {
        std::vector<uint16_t> t(256 + kLabCbrtSize);
        lab_tables_host(t.data());
        return t;
    }();
    if (hipMemcpyAsync(tabs, host_tabs.data(), host_tabs.size() * sizeof(uint16_t), hipMemcpyHostToDevice, s) != hipSuccess) {
        lf::set_error("lf_inclusive_mask: table upload failed");
        return LF_ERR_LAUNCH;
    }
    int rc = lf_rgb2gray_u8(rgb, gray, px, stream);
    if (rc != LF_OK) return rc;
    rc = lf_gauss_blur_u8(gray, blur, n, h, w, 1, kq15, 15, stream);
    if (rc != LF_OK) return rc;
    if (const size_t fl = fused_lds_bytes(h, w, false)) {
        // the image fits a CU: Canny and the predicates in one workgroup per image, gray plane and map in LDS
        static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(inclusive_fused_kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024) == hipSuccess;
        if (ok) {
            inclusive_fused_kernel<<<n, kFuseT, fl, s>>>(rgb, gray, blur, tabs, bits, h, w, std::max(0, green_lo - 10),
                                                         std::min(179, green_hi + 15));
            inclusive_morph_kernel<<<n, kBlock, lds, s>>>(bits, mask, rn, parent, area, h, w, wpr,
                                                          (int)mask_runs_per_image(h, w));
            return lf::check_launch("lf_inclusive_mask");
        }
    }
    const dim3 grid_px((hw + kBlock - 1) / kBlock, n);
    canny_sobel_l1_kernel<<<grid_px, kBlock, 0, s>>>(gray, mag, dxdy, h, w);
    canny_nms_kernel<<<grid_px, kBlock, 0, s>>>(mag, dxdy, map, h, w, 30, 100);   // cv2.Canny(gray, 30, 100)
    {
        static const size_t hyst_cap = []() {
            const size_t want = 156 * 1024;
            return hipFuncSetAttribute(reinterpret_cast<const void*>(canny_hysteresis_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)want) == hipSuccess
                       ? want
                       : (size_t)60 * 1024;
        }();
        const int in_lds = (size_t)hw <= hyst_cap;
        canny_hysteresis_kernel<<<n, kHystThreads, in_lds ? (size_t)((hw + 15) & ~15) : 0, s>>>(map, h, w, in_lds);
    }
    const long segs = (long)n * h * ((w + 63) / 64);
    const unsigned pgrid = (unsigned)std::min<long>((segs + 3) / 4, 2048);
    inclusive_pred_kernel<<<pgrid, kBlock, 0, s>>>(rgb, gray, blur, map, tabs, bits, n, h, w,
                                                   std::max(0, green_lo - 10), std::min(179, green_hi + 15));
    inclusive_morph_kernel<<<n, kBlock, lds, s>>>(bits, mask, rn, parent, area, h, w, wpr,
                                                  (int)mask_runs_per_image(h, w));
    return lf::check_launch("lf_inclusive_mask");
}

}  // extern "C"
