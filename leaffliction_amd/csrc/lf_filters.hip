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

constexpr size_t kAlign = 256;
inline size_t up(size_t v) { return (v + kAlign - 1) & ~(kAlign - 1); }

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
    minmax_init_kernel<<<(n * 6 + 255) / 256, 256, 0, s>>>(mm, n);
    int rc = lf_rgb2gray_u8(rgb, pa, px, stream);
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

}  // extern "C"
