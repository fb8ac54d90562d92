// libleafhip — augmentation / input-side kernels (uint8, HBM-bound, bit-exact).
//
// All kernels are written for gfx950 (MI355X): 64-lane waves, 256-thread
// workgroups, 12- or 16-byte per-lane global accesses so that one wave
// instruction moves 768 B / 1 KiB of contiguous memory, LDS for per-image tables
// and histograms.  No kernel here is GEMM-shaped; the bound is HBM (or VALU for the
// 15x15 blur) — see DESIGN.md §kernels.
//
// This translation unit is compiled with -ffp-contract=off: the double-precision
// arithmetic below restates Python / Pillow expressions whose mul and add round
// separately.
#include "lf_common.h"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ unsigned byte_of(unsigned w, int i) { return (w >> (8 * i)) & 0xffu; }

// ---------------------------------------------------------------------------
// pack: u8 HWC -> f32 NCHW, x/255 (+ optional per-channel normalisation)
// ---------------------------------------------------------------------------
// One thread = 4 pixels = 12 input bytes (3 dwords) -> one float4 per plane.
template <bool NORM, bool NT>
__global__ __launch_bounds__(kBlock) void pack_kernel(const uint8_t* __restrict__ in,
                                                      float* __restrict__ out, unsigned hw4,
                                                      float m0, float m1, float m2, float d0,
                                                      float d1, float d2) {
    const unsigned n = blockIdx.y;
    const size_t hw = (size_t)hw4 * 4;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(in + (size_t)n * hw * 3);
    float* dst = out + (size_t)n * hw * 3;
    for (unsigned g = blockIdx.x * kBlock + threadIdx.x; g < hw4; g += gridDim.x * kBlock) {
        const unsigned w0 = lf::ldg<NT>(src + 3 * g), w1 = lf::ldg<NT>(src + 3 * g + 1),
                       w2 = lf::ldg<NT>(src + 3 * g + 2);
        float r[4] = {(float)byte_of(w0, 0), (float)byte_of(w0, 3), (float)byte_of(w1, 2),
                      (float)byte_of(w2, 1)};
        float gch[4] = {(float)byte_of(w0, 1), (float)byte_of(w1, 0), (float)byte_of(w1, 3),
                        (float)byte_of(w2, 2)};
        float b[4] = {(float)byte_of(w0, 2), (float)byte_of(w1, 1), (float)byte_of(w2, 0),
                      (float)byte_of(w2, 3)};
        lf::f32x4 o0, o1, o2;
        float* pr = reinterpret_cast<float*>(&o0);
        float* pg = reinterpret_cast<float*>(&o1);
        float* pb = reinterpret_cast<float*>(&o2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float vr = __fdiv_rn(r[i], 255.0f), vg = __fdiv_rn(gch[i], 255.0f),
                  vb = __fdiv_rn(b[i], 255.0f);
            if (NORM) {
                vr = __fdiv_rn(__fsub_rn(vr, m0), d0);
                vg = __fdiv_rn(__fsub_rn(vg, m1), d1);
                vb = __fdiv_rn(__fsub_rn(vb, m2), d2);
            }
            pr[i] = vr;
            pg[i] = vg;
            pb[i] = vb;
        }
        lf::stg<NT>(reinterpret_cast<lf::f32x4*>(dst) + g, o0);
        lf::stg<NT>(reinterpret_cast<lf::f32x4*>(dst + hw) + g, o1);
        lf::stg<NT>(reinterpret_cast<lf::f32x4*>(dst + 2 * hw) + g, o2);
    }
}

// Generic fallback (H*W not a multiple of 4): one thread per pixel.
template <bool NORM>
__global__ __launch_bounds__(kBlock) void pack_scalar_kernel(const uint8_t* __restrict__ in,
                                                             float* __restrict__ out, size_t hw,
                                                             float m0, float m1, float m2,
                                                             float d0, float d1, float d2) {
    const unsigned n = blockIdx.y;
    const uint8_t* src = in + (size_t)n * hw * 3;
    float* dst = out + (size_t)n * hw * 3;
    const float m[3] = {m0, m1, m2}, d[3] = {d0, d1, d2};
    for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < hw;
         p += (size_t)gridDim.x * kBlock) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = __fdiv_rn((float)src[3 * p + c], 255.0f);
            if (NORM) v = __fdiv_rn(__fsub_rn(v, m[c]), d[c]);
            dst[(size_t)c * hw + p] = v;
        }
    }
}

// ---------------------------------------------------------------------------
// hist: per image, per channel 256-bin histogram (int32, exact)
// ---------------------------------------------------------------------------
// grid = (splits, n).  Each wave owns a private 3x256 LDS histogram so that the
// 64 lanes of one ds_add only collide with each other; the four copies are summed
// and flushed with one global atomic per bin per workgroup.
constexpr int kHistCopies = kBlock / 64;

template <bool NT>
__global__ __launch_bounds__(kBlock) void hist_kernel(const uint8_t* __restrict__ in,
                                                      int32_t* __restrict__ hist, size_t nbytes) {
    __shared__ unsigned lh[kHistCopies][768];
    const unsigned n = blockIdx.y;
    for (int i = threadIdx.x; i < kHistCopies * 768; i += kBlock) (&lh[0][0])[i] = 0;
    __syncthreads();
    unsigned* my = lh[threadIdx.x >> 6];
    const uint8_t* base = in + (size_t)n * nbytes;
    // 16-byte aligned middle part [a0, a1) of this image's bytes; head/tail are scalar.
    const size_t addr = reinterpret_cast<size_t>(base);
    size_t head = (16 - (addr & 15)) & 15;
    if (head > nbytes) head = nbytes;
    const size_t nchunks = (nbytes - head) / 16;
    const lf::u32x4* mid = reinterpret_cast<const lf::u32x4*>(base + head);
    const unsigned hc = (unsigned)(head % 3);
    auto tally = [&](const lf::u32x4 v, size_t q) {
        // channel of byte 0 of this chunk: (head + 16 q) % 3 = (hc + q) % 3
        const unsigned r = (hc + (unsigned)(q % 3)) % 3;
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            unsigned c = r + (j % 3);
            c = c >= 3 ? c - 3 : c;
            atomicAdd(&my[c * 256 + byte_of(w[j >> 2], j & 3)], 1u);
        }
    };
    // four 16-byte loads in flight per thread: one load per trip leaves the kernel
    // latency-bound (32 KB in flight per CU ~ 4.2 TB/s)
    const size_t stride = (size_t)gridDim.x * kBlock;
    size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x;
    for (; q + 3 * stride < nchunks; q += 4 * stride) {
        const lf::u32x4 v0 = lf::ldg<NT>(mid + q), v1 = lf::ldg<NT>(mid + q + stride),
                        v2 = lf::ldg<NT>(mid + q + 2 * stride), v3 = lf::ldg<NT>(mid + q + 3 * stride);
        tally(v0, q);
        tally(v1, q + stride);
        tally(v2, q + 2 * stride);
        tally(v3, q + 3 * stride);
    }
    for (; q < nchunks; q += stride) tally(lf::ldg<NT>(mid + q), q);
    if (blockIdx.x == 0) {
        const size_t tail0 = head + nchunks * 16;
        for (size_t i = threadIdx.x; i < head; i += kBlock)
            atomicAdd(&my[(i % 3) * 256 + base[i]], 1u);
        for (size_t i = tail0 + threadIdx.x; i < nbytes; i += kBlock)
            atomicAdd(&my[(i % 3) * 256 + base[i]], 1u);
    }
    __syncthreads();
    int32_t* gh = hist + (size_t)n * 768;
    for (int i = threadIdx.x; i < 768; i += kBlock) {
        unsigned s = 0;
#pragma unroll
        for (int k = 0; k < kHistCopies; ++k) s += lh[k][i];
        if (s) atomicAdd(&gh[i], (int32_t)s);
    }
}

// One workgroup per image, ONE table per workgroup with 16 lane slots per bin: tab[bin][lane % 16].
// The word a lane touches sits in bank 16*(bin & 1) + lane % 16, so the 64 lanes of a ds_add spread
// over all 32 banks whatever the bytes are (two lanes per bank on average; equal bytes in
// neighbouring lanes — flat leaf regions — land in different slots instead of serialising on one
// word), where the per-wave 768-word histogram above takes ~3-4 LDS cycles per bank on uniform bytes
// and more on flat images.  The image's 768 sums go out with plain stores (no global atomics).
constexpr int kHistSlots = 16;

template <bool NT>
__global__ __launch_bounds__(kBlock) void hist_slot_kernel(const uint8_t* __restrict__ in,
                                                           int32_t* __restrict__ hist, size_t nbytes, int n) {
    __shared__ __attribute__((aligned(16))) unsigned tab[768 * kHistSlots];
    const unsigned slot = threadIdx.x & (kHistSlots - 1);
    for (int img = blockIdx.x; img < n; img += gridDim.x) {
        for (int i = threadIdx.x; i < 768 * kHistSlots / 4; i += kBlock)
            reinterpret_cast<lf::u32x4*>(tab)[i] = lf::u32x4{0u, 0u, 0u, 0u};
        __syncthreads();
        const uint8_t* base = in + (size_t)img * nbytes;
        const size_t addr = reinterpret_cast<size_t>(base);
        size_t head = (16 - (addr & 15)) & 15;
        if (head > nbytes) head = nbytes;
        const size_t nchunks = (nbytes - head) / 16;
        const lf::u32x4* mid = reinterpret_cast<const lf::u32x4*>(base + head);
        const unsigned hc = (unsigned)(head % 3);
        auto tally = [&](const lf::u32x4 v, size_t q) {
            const unsigned r = (hc + (unsigned)(q % 3)) % 3;   // channel of byte 0 of this chunk
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                unsigned c = r + (j % 3);
                c = c >= 3 ? c - 3 : c;
                atomicAdd(&tab[(c * 256 + byte_of(w[j >> 2], j & 3)) * kHistSlots + slot], 1u);
            }
        };
        size_t q = threadIdx.x;
        for (; q + 3 * kBlock < nchunks; q += 4 * kBlock) {
            const lf::u32x4 v0 = lf::ldg<NT>(mid + q), v1 = lf::ldg<NT>(mid + q + kBlock),
                            v2 = lf::ldg<NT>(mid + q + 2 * kBlock), v3 = lf::ldg<NT>(mid + q + 3 * kBlock);
            tally(v0, q);
            tally(v1, q + kBlock);
            tally(v2, q + 2 * kBlock);
            tally(v3, q + 3 * kBlock);
        }
        for (; q < nchunks; q += kBlock) tally(lf::ldg<NT>(mid + q), q);
        const size_t tail0 = head + nchunks * 16;
        for (size_t i = threadIdx.x; i < head; i += kBlock)
            atomicAdd(&tab[((i % 3) * 256 + base[i]) * kHistSlots + slot], 1u);
        for (size_t i = tail0 + threadIdx.x; i < nbytes; i += kBlock)
            atomicAdd(&tab[((i % 3) * 256 + base[i]) * kHistSlots + slot], 1u);
        __syncthreads();
        int32_t* gh = hist + (size_t)img * 768;
        for (int b = threadIdx.x; b < 768; b += kBlock) {
            unsigned sum = 0;
#pragma unroll
            for (int k = 0; k < kHistSlots / 4; ++k) {
                const lf::u32x4 v = reinterpret_cast<const lf::u32x4*>(tab)[b * (kHistSlots / 4) + k];
                sum += (v.x + v.y) + (v.z + v.w);
            }
            gh[b] = (int32_t)sum;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// autocontrast LUT (Pillow ImageOps.autocontrast arithmetic, IEEE double)
// ---------------------------------------------------------------------------
// Python's float floor division x // y for y > 0 (CPython floatobject.c _float_div_mod).
__device__ double py_floordiv(double vx, double wx) {
    double mod = fmod(vx, wx);
    double div = __ddiv_rn(__dsub_rn(vx, mod), wx);
    if (mod != 0.0) {
        if ((wx < 0) != (mod < 0)) {
            div = __dsub_rn(div, 1.0);
        }
    }
    if (div != 0.0) {
        double f = floor(div);
        if (__dsub_rn(div, f) > 0.5) f = __dadd_rn(f, 1.0);
        return f;
    }
    return copysign(0.0, __ddiv_rn(vx, wx));
}

// One thread per (image, channel); 256-entry scans are tiny next to the image passes.
__global__ void autocontrast_lut_kernel(const int32_t* __restrict__ hist,
                                        const double* __restrict__ cutoff,
                                        uint8_t* __restrict__ lut, int total) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int n = t / 3;
    const int32_t* hsrc = hist + (size_t)t * 256;
    uint8_t* dst = lut + (size_t)t * 256;
    int h[256];
    long long npx = 0;
    for (int i = 0; i < 256; ++i) {
        h[i] = hsrc[i];
        npx += h[i];
    }
    const double co = cutoff[n];
    if (co != 0.0) {
        // cut = int(n * cutoff // 100)
        long long cut = (long long)py_floordiv(__dmul_rn((double)npx, co), 100.0);
        for (int lo = 0; lo < 256; ++lo) {
            if (cut > h[lo]) {
                cut -= h[lo];
                h[lo] = 0;
            } else {
                h[lo] -= (int)cut;
                cut = 0;
            }
            if (cut <= 0) break;
        }
        cut = (long long)py_floordiv(__dmul_rn((double)npx, co), 100.0);
        for (int hi = 255; hi >= 0; --hi) {
            if (cut > h[hi]) {
                cut -= h[hi];
                h[hi] = 0;
            } else {
                h[hi] -= (int)cut;
                cut = 0;
            }
            if (cut <= 0) break;
        }
    }
    int lo = 0, hi = 255;
    for (lo = 0; lo < 256; ++lo)
        if (h[lo]) break;
    if (lo == 256) lo = 255;  // Python: loop variable keeps its last value
    for (hi = 255; hi >= 0; --hi)
        if (h[hi]) break;
    if (hi < 0) hi = 0;
    if (hi <= lo) {
        for (int i = 0; i < 256; ++i) dst[i] = (uint8_t)i;
    } else {
        const double scale = __ddiv_rn(255.0, (double)(hi - lo));
        const double offset = __dmul_rn(-(double)lo, scale);
        for (int i = 0; i < 256; ++i) {
            int v = (int)__dadd_rn(__dmul_rn((double)i, scale), offset);
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            dst[i] = (uint8_t)v;
        }
    }
}

// ---------------------------------------------------------------------------
// LUT apply (Image.point)
// ---------------------------------------------------------------------------
template <bool NT>
__global__ __launch_bounds__(kBlock) void lut_apply_kernel(const uint8_t* __restrict__ in,
                                                           const uint8_t* __restrict__ lut,
                                                           uint8_t* __restrict__ out,
                                                           size_t nbytes) {
    __shared__ uint8_t sl[768];
    const unsigned n = blockIdx.y;
    for (int i = threadIdx.x; i < 768; i += kBlock) sl[i] = lut[(size_t)n * 768 + i];
    __syncthreads();
    const uint8_t* src = in + (size_t)n * nbytes;
    uint8_t* dst = out + (size_t)n * nbytes;
    const size_t nchunks = nbytes / 16;  // caller guarantees 16-byte aligned images
    const lf::u32x4* s4 = reinterpret_cast<const lf::u32x4*>(src);
    lf::u32x4* d4 = reinterpret_cast<lf::u32x4*>(dst);
    for (size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x; q < nchunks;
         q += (size_t)gridDim.x * kBlock) {
        const lf::u32x4 v = lf::ldg<NT>(s4 + q);
        const unsigned r = (unsigned)(q % 3);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
        unsigned o[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            unsigned c = r + (j % 3);
            c = c >= 3 ? c - 3 : c;
            o[j >> 2] |= (unsigned)sl[c * 256 + byte_of(w[j >> 2], j & 3)] << (8 * (j & 3));
        }
        lf::u32x4 ov;
        ov.x = o[0]; ov.y = o[1]; ov.z = o[2]; ov.w = o[3];
        lf::stg<NT>(d4 + q, ov);
    }
    if (blockIdx.x == 0) {
        for (size_t i = nchunks * 16 + threadIdx.x; i < nbytes; i += kBlock)
            dst[i] = sl[(i % 3) * 256 + src[i]];
    }
}

__global__ __launch_bounds__(kBlock) void lut_apply_scalar_kernel(const uint8_t* __restrict__ in,
                                                                  const uint8_t* __restrict__ lut,
                                                                  uint8_t* __restrict__ out,
                                                                  size_t nbytes) {
    __shared__ uint8_t sl[768];
    const unsigned n = blockIdx.y;
    for (int i = threadIdx.x; i < 768; i += kBlock) sl[i] = lut[(size_t)n * 768 + i];
    __syncthreads();
    const uint8_t* src = in + (size_t)n * nbytes;
    uint8_t* dst = out + (size_t)n * nbytes;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nbytes;
         i += (size_t)gridDim.x * kBlock)
        dst[i] = sl[(i % 3) * 256 + src[i]];
}

// ---------------------------------------------------------------------------
// flip
// ---------------------------------------------------------------------------
// One thread = 4 pixels (12 bytes).  w % 4 == 0.
__global__ __launch_bounds__(kBlock) void flip_kernel(const uint8_t* __restrict__ in,
                                                      uint8_t* __restrict__ out,
                                                      const int32_t* __restrict__ mode, int h,
                                                      int w4) {
    const unsigned n = blockIdx.y;
    const int m = mode[n];
    const size_t img = (size_t)h * w4 * 3;  // dwords per image
    const uint32_t* src = reinterpret_cast<const uint32_t*>(in) + (size_t)n * img;
    uint32_t* dst = reinterpret_cast<uint32_t*>(out) + (size_t)n * img;
    const unsigned total = (unsigned)h * w4;
    for (unsigned t = blockIdx.x * kBlock + threadIdx.x; t < total; t += gridDim.x * kBlock) {
        const unsigned y = t / w4, g = t - y * w4;
        const uint32_t* s = src + (m == 0 ? ((size_t)y * w4 + (w4 - 1 - g)) * 3
                                          : ((size_t)(h - 1 - y) * w4 + g) * 3);
        uint32_t* d = dst + (size_t)t * 3;
        const unsigned w0 = s[0], w1 = s[1], w2 = s[2];
        unsigned o0 = w0, o1 = w1, o2 = w2;
        if (m == 0) {  // FLIP_LEFT_RIGHT: reverse the pixel order inside the group
            o0 = byte_of(w2, 1) | byte_of(w2, 2) << 8 | byte_of(w2, 3) << 16 | byte_of(w1, 2) << 24;
            o1 = byte_of(w1, 3) | byte_of(w2, 0) << 8 | byte_of(w0, 3) << 16 | byte_of(w1, 0) << 24;
            o2 = byte_of(w1, 1) | byte_of(w0, 0) << 8 | byte_of(w0, 1) << 16 | byte_of(w0, 2) << 24;
        }
        d[0] = o0;
        d[1] = o1;
        d[2] = o2;
    }
}

__global__ __launch_bounds__(kBlock) void flip_scalar_kernel(const uint8_t* __restrict__ in,
                                                             uint8_t* __restrict__ out,
                                                             const int32_t* __restrict__ mode,
                                                             int h, int w) {
    const unsigned n = blockIdx.y;
    const int m = mode[n];
    const size_t img = (size_t)h * w * 3;
    const uint8_t* src = in + (size_t)n * img;
    uint8_t* dst = out + (size_t)n * img;
    const size_t total = (size_t)h * w;
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total;
         t += (size_t)gridDim.x * kBlock) {
        const size_t y = t / w, x = t - y * w;
        const size_t s = m == 0 ? (y * w + (w - 1 - x)) : ((h - 1 - y) * w + x);
        dst[3 * t] = src[3 * s];
        dst[3 * t + 1] = src[3 * s + 1];
        dst[3 * t + 2] = src[3 * s + 2];
    }
}

// ---------------------------------------------------------------------------
// noise add with uint8 wrap-around
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void noise_add_kernel(const uint8_t* __restrict__ in,
                                                           const double* __restrict__ noise,
                                                           uint8_t* __restrict__ out, size_t n) {
    // numpy float64 -> uint8 astype on x86-64: truncate toward zero, keep the low 8 bits.
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
         i += (size_t)gridDim.x * kBlock) {
        const long long t = (long long)noise[i];
        out[i] = (uint8_t)(in[i] + (uint8_t)t);
    }
}

// out = in + add (mod 256) on whole dwords: the same wrap-around add when the noise has already been
// cast to uint8 on the host (numpy's own astype, image_augmenter.py:121-123)
__global__ __launch_bounds__(kBlock) void add_wrap_u8_kernel(const uint32_t* __restrict__ a,
                                                             const uint32_t* __restrict__ b,
                                                             uint32_t* __restrict__ out, size_t nwords) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nwords; i += (size_t)gridDim.x * kBlock) {
        const uint32_t x = a[i], y = b[i];
        // per-byte add without carries across bytes
        out[i] = ((x & 0x7f7f7f7fu) + (y & 0x7f7f7f7fu)) ^ ((x ^ y) & 0x80808080u);
    }
}

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t* r) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

// One thread = 4 bytes: one Philox block -> two Box-Muller pairs.
template <bool NT>
__global__ __launch_bounds__(kBlock) void noise_philox_kernel(const uint8_t* __restrict__ in,
                                                              uint8_t* __restrict__ out,
                                                              size_t nwords, size_t nbytes,
                                                              uint64_t seed, float sigma) {
    for (size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x; q < nwords;
         q += (size_t)gridDim.x * kBlock) {
        uint32_t r[4];
        philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), 0u, 0u, (uint32_t)seed,
                      (uint32_t)(seed >> 32), r);
        float z[4];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float u1 = ((float)(r[2 * k] >> 8) + 0.5f) * (1.0f / 16777216.0f);
            const float u2 = ((float)(r[2 * k + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
            const float rad = sigma * __fsqrt_rn(-2.0f * __logf(u1));
            float s, c;
            __sincosf(6.283185307179586f * u2, &s, &c);
            z[2 * k] = rad * c;
            z[2 * k + 1] = rad * s;
        }
        const size_t b = q * 4;
        if (b + 4 <= nbytes) {
            const unsigned v = lf::ldg<NT>(reinterpret_cast<const uint32_t*>(in) + q);
            unsigned o = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o |= ((byte_of(v, j) + (unsigned)(int)z[j]) & 0xffu) << (8 * j);
            lf::stg<NT>(reinterpret_cast<uint32_t*>(out) + q, o);
        } else {
            for (size_t i = b; i < nbytes; ++i) out[i] = (uint8_t)(in[i] + (uint8_t)(int)z[i - b]);
        }
    }
}

// The distortion's first two passes in one (round 3): out = in + noise (mod 256) AND the per-image, per-channel
// histogram of `out`, which autocontrast needs next — one workgroup per image with the slotted table of
// hist_slot_kernel, so the noisy image is not read back from memory just to be counted (5 image passes -> 4).
// ADD: the noise is a uint8 plane (cast on the host by numpy, lf_add_wrap_u8); otherwise Philox4x32-10 + Box-Muller
// keyed by (seed, dword index in the BATCH), the very values lf_noise_philox_add_u8 draws.  nbytes % 16 == 0.
template <bool ADD>
__global__ __launch_bounds__(kBlock) void noise_hist_kernel(const uint8_t* __restrict__ in,
                                                            const uint8_t* __restrict__ add,
                                                            uint8_t* __restrict__ out, int32_t* __restrict__ hist,
                                                            size_t nbytes, int n, uint64_t seed, float sigma) {
    __shared__ __attribute__((aligned(16))) unsigned tab[768 * kHistSlots];
    const unsigned slot = threadIdx.x & (kHistSlots - 1);
    const size_t nchunks = nbytes / 16;
    for (int img = blockIdx.x; img < n; img += gridDim.x) {
        for (int i = threadIdx.x; i < 768 * kHistSlots / 4; i += kBlock)
            reinterpret_cast<lf::u32x4*>(tab)[i] = lf::u32x4{0u, 0u, 0u, 0u};
        __syncthreads();
        const lf::u32x4* src = reinterpret_cast<const lf::u32x4*>(in + (size_t)img * nbytes);
        const lf::u32x4* nz = ADD ? reinterpret_cast<const lf::u32x4*>(add + (size_t)img * nbytes) : nullptr;
        lf::u32x4* dst = reinterpret_cast<lf::u32x4*>(out + (size_t)img * nbytes);
        const size_t word0 = (size_t)img * (nbytes / 4);
        for (size_t q = threadIdx.x; q < nchunks; q += kBlock) {
            const lf::u32x4 v = src[q];
            const unsigned x[4] = {v.x, v.y, v.z, v.w};
            unsigned y[4];
            if (ADD) {
                const lf::u32x4 a = nz[q];
                const unsigned b[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)   // per-byte add without carries across bytes
                    y[k] = ((x[k] & 0x7f7f7f7fu) + (b[k] & 0x7f7f7f7fu)) ^ ((x[k] ^ b[k]) & 0x80808080u);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const size_t wq = word0 + 4 * q + k;
                    uint32_t r[4];
                    philox4x32_10((uint32_t)wq, (uint32_t)(wq >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
                    float z[4];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float u1 = ((float)(r[2 * j] >> 8) + 0.5f) * (1.0f / 16777216.0f);
                        const float u2 = ((float)(r[2 * j + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
                        const float rad = sigma * __fsqrt_rn(-2.0f * __logf(u1));
                        float sn, cs;
                        __sincosf(6.283185307179586f * u2, &sn, &cs);
                        z[2 * j] = rad * cs;
                        z[2 * j + 1] = rad * sn;
                    }
                    unsigned o = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o |= ((byte_of(x[k], j) + (unsigned)(int)z[j]) & 0xffu) << (8 * j);
                    y[k] = o;
                }
            }
            dst[q] = lf::u32x4{y[0], y[1], y[2], y[3]};
            const unsigned r0 = (unsigned)((q * 16) % 3);   // channel of byte 0 of this chunk (images start on a pixel)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                unsigned c = r0 + (j % 3);
                c = c >= 3 ? c - 3 : c;
                atomicAdd(&tab[(c * 256 + byte_of(y[j >> 2], j & 3)) * kHistSlots + slot], 1u);
            }
        }
        __syncthreads();
        int32_t* gh = hist + (size_t)img * 768;
        for (int b = threadIdx.x; b < 768; b += kBlock) {
            unsigned sum = 0;
#pragma unroll
            for (int k = 0; k < kHistSlots / 4; ++k) {
                const lf::u32x4 v = reinterpret_cast<const lf::u32x4*>(tab)[b * (kHistSlots / 4) + k];
                sum += (v.x + v.y) + (v.z + v.w);
            }
            gh[b] = (int32_t)sum;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// mask-and-composite
// ---------------------------------------------------------------------------
// One thread = 4 pixels: 12 image bytes + 4 mask bytes.
template <bool NT>
__global__ __launch_bounds__(kBlock) void composite_kernel(const uint8_t* __restrict__ img,
                                                           const uint8_t* __restrict__ mask,
                                                           uint8_t* __restrict__ out, size_t npx4,
                                                           unsigned fill) {
    const uint32_t* s = reinterpret_cast<const uint32_t*>(img);
    const uint32_t* m = reinterpret_cast<const uint32_t*>(mask);
    uint32_t* d = reinterpret_cast<uint32_t*>(out);
    const unsigned f4 = fill * 0x01010101u;
    for (size_t g = (size_t)blockIdx.x * kBlock + threadIdx.x; g < npx4;
         g += (size_t)gridDim.x * kBlock) {
        const unsigned mk = lf::ldg<NT>(m + g);
        const unsigned w0 = lf::ldg<NT>(s + 3 * g), w1 = lf::ldg<NT>(s + 3 * g + 1),
                       w2 = lf::ldg<NT>(s + 3 * g + 2);
        // per-pixel keep masks expanded to the byte lanes each pixel occupies
        const unsigned k0 = byte_of(mk, 0) > 127, k1 = byte_of(mk, 1) > 127,
                       k2 = byte_of(mk, 2) > 127, k3 = byte_of(mk, 3) > 127;
        const unsigned m0 = (k0 ? 0x00ffffffu : 0u) | (k1 ? 0xff000000u : 0u);
        const unsigned m1 = (k1 ? 0x0000ffffu : 0u) | (k2 ? 0xffff0000u : 0u);
        const unsigned m2 = (k2 ? 0x000000ffu : 0u) | (k3 ? 0xffffff00u : 0u);
        lf::stg<NT>(d + 3 * g, (w0 & m0) | (f4 & ~m0));
        lf::stg<NT>(d + 3 * g + 1, (w1 & m1) | (f4 & ~m1));
        lf::stg<NT>(d + 3 * g + 2, (w2 & m2) | (f4 & ~m2));
    }
}

__global__ __launch_bounds__(kBlock) void composite_scalar_kernel(const uint8_t* __restrict__ img,
                                                                  const uint8_t* __restrict__ mask,
                                                                  uint8_t* __restrict__ out,
                                                                  size_t start, size_t npx,
                                                                  unsigned fill) {
    for (size_t p = start + (size_t)blockIdx.x * kBlock + threadIdx.x; p < npx;
         p += (size_t)gridDim.x * kBlock) {
        const bool keep = mask[p] > 127;
#pragma unroll
        for (int c = 0; c < 3; ++c) out[3 * p + c] = keep ? img[3 * p + c] : (uint8_t)fill;
    }
}

// ---------------------------------------------------------------------------
// OpenCV 8-bit colour conversions
// ---------------------------------------------------------------------------
// OpenCV RGB2HSV_b (imgproc color_hsv): hsv_shift = 12, tables
// sdiv[i] = cvRound((255<<12)/i), hdiv180[i] = cvRound((180<<12)/(6 i)), [0] = 0.
__device__ __forceinline__ void build_hsv_tables(int* sdiv, int* hdiv) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        if (i == 0) {
            sdiv[0] = 0;
            hdiv[0] = 0;
        } else {
            sdiv[i] = __double2int_rn(__ddiv_rn(1044480.0, (double)i));
            hdiv[i] = __double2int_rn(__ddiv_rn(737280.0, __dmul_rn(6.0, (double)i)));
        }
    }
}

__device__ __forceinline__ void rgb2hsv_px(int r, int g, int b, const int* sdiv, const int* hdiv,
                                           int& h, int& s, int& v) {
    v = max(r, max(g, b));
    const int vmin = min(r, min(g, b));
    const int diff = v - vmin;
    const int vr = v == r ? -1 : 0;
    const int vg = v == g ? -1 : 0;
    // operands fit 24 bits (diff <= 255, sdiv < 2^20, |h| <= 1275, hdiv < 2^17): v_mad_i32_i24
    s = (__mul24(diff, sdiv[v]) + (1 << 11)) >> 12;
    h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    h = (__mul24(h, hdiv[diff]) + (1 << 11)) >> 12;
    h += h < 0 ? 180 : 0;
}

__device__ __forceinline__ void unpack4(unsigned w0, unsigned w1, unsigned w2, int* r, int* g,
                                        int* b) {
    r[0] = byte_of(w0, 0); g[0] = byte_of(w0, 1); b[0] = byte_of(w0, 2);
    r[1] = byte_of(w0, 3); g[1] = byte_of(w1, 0); b[1] = byte_of(w1, 1);
    r[2] = byte_of(w1, 2); g[2] = byte_of(w1, 3); b[2] = byte_of(w2, 0);
    r[3] = byte_of(w2, 1); g[3] = byte_of(w2, 2); b[3] = byte_of(w2, 3);
}

template <bool NT>
__global__ __launch_bounds__(kBlock) void rgb2hsv_kernel(const uint8_t* __restrict__ rgb,
                                                         uint8_t* __restrict__ hsv, size_t npx) {
    __shared__ int sdiv[256], hdiv[256];
    build_hsv_tables(sdiv, hdiv);
    __syncthreads();
    const size_t npx4 = npx / 4;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(rgb);
    uint32_t* d = reinterpret_cast<uint32_t*>(hsv);
    for (size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x; q < npx4;
         q += (size_t)gridDim.x * kBlock) {
        int r[4], g[4], b[4], hh[4], ss[4], vv[4];
        unpack4(lf::ldg<NT>(s + 3 * q), lf::ldg<NT>(s + 3 * q + 1), lf::ldg<NT>(s + 3 * q + 2), r, g, b);
#pragma unroll
        for (int i = 0; i < 4; ++i) rgb2hsv_px(r[i], g[i], b[i], sdiv, hdiv, hh[i], ss[i], vv[i]);
        lf::stg<NT>(d + 3 * q, (uint32_t)(hh[0] | ss[0] << 8 | vv[0] << 16 | hh[1] << 24));
        lf::stg<NT>(d + 3 * q + 1, (uint32_t)(ss[1] | vv[1] << 8 | hh[2] << 16 | ss[2] << 24));
        lf::stg<NT>(d + 3 * q + 2, (uint32_t)(vv[2] | hh[3] << 8 | ss[3] << 16 | vv[3] << 24));
    }
    if (blockIdx.x == 0) {
        for (size_t p = npx4 * 4 + threadIdx.x; p < npx; p += kBlock) {
            int h, sv, v;
            rgb2hsv_px(rgb[3 * p], rgb[3 * p + 1], rgb[3 * p + 2], sdiv, hdiv, h, sv, v);
            hsv[3 * p] = (uint8_t)h;
            hsv[3 * p + 1] = (uint8_t)sv;
            hsv[3 * p + 2] = (uint8_t)v;
        }
    }
}

__device__ __forceinline__ unsigned gray_px(int r, int g, int b) {
    return (unsigned)(r * 4899 + g * 9617 + b * 1868 + 8192) >> 14;
}

template <bool NT>
__global__ __launch_bounds__(kBlock) void rgb2gray_kernel(const uint8_t* __restrict__ rgb,
                                                          uint8_t* __restrict__ gray, size_t npx) {
    const size_t npx4 = npx / 4;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(rgb);
    uint32_t* d = reinterpret_cast<uint32_t*>(gray);
    for (size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x; q < npx4;
         q += (size_t)gridDim.x * kBlock) {
        int r[4], g[4], b[4];
        unpack4(lf::ldg<NT>(s + 3 * q), lf::ldg<NT>(s + 3 * q + 1), lf::ldg<NT>(s + 3 * q + 2), r, g, b);
        lf::stg<NT>(d + q, (uint32_t)(gray_px(r[0], g[0], b[0]) | gray_px(r[1], g[1], b[1]) << 8 |
                               gray_px(r[2], g[2], b[2]) << 16 | gray_px(r[3], g[3], b[3]) << 24));
    }
    if (blockIdx.x == 0) {
        for (size_t p = npx4 * 4 + threadIdx.x; p < npx; p += kBlock)
            gray[p] = (uint8_t)gray_px(rgb[3 * p], rgb[3 * p + 1], rgb[3 * p + 2]);
    }
}

// ---------------------------------------------------------------------------
// HSV colour-region statistics (hist.py)
// ---------------------------------------------------------------------------
// grid = (splits, n).  Per pixel: fused RGB->HSV, leaf predicate, 13 region predicates
// (wave ballot + popcount), and H/S/V 256-bin histograms over the leaf pixels.
__global__ __launch_bounds__(kBlock) void hsv_stats_kernel(const uint8_t* __restrict__ rgb,
                                                           int32_t* __restrict__ counts,
                                                           int32_t* __restrict__ hsv_hist,
                                                           size_t npx) {
    __shared__ int sdiv[256], hdiv[256];
    __shared__ unsigned lh[768];
    __shared__ unsigned lc[LF_HSV_NCOUNTS];
    build_hsv_tables(sdiv, hdiv);
    for (int i = threadIdx.x; i < 768; i += kBlock) lh[i] = 0;
    if (threadIdx.x < LF_HSV_NCOUNTS) lc[threadIdx.x] = 0;
    __syncthreads();
    const unsigned n = blockIdx.y;
    const uint8_t* src = rgb + (size_t)n * npx * 3;
    unsigned cnt[LF_HSV_NCOUNTS];
#pragma unroll
    for (int i = 0; i < LF_HSV_NCOUNTS; ++i) cnt[i] = 0;
    for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < npx;
         p += (size_t)gridDim.x * kBlock) {
        int h, s, v;
        rgb2hsv_px(src[3 * p], src[3 * p + 1], src[3 * p + 2], sdiv, hdiv, h, s, v);
        const bool leaf = (s > 10) && (v > 15) && (v < 245);
        if (leaf) {
            atomicAdd(&lh[h], 1u);
            atomicAdd(&lh[256 + s], 1u);
            atomicAdd(&lh[512 + v], 1u);
            cnt[0] += 1;
            cnt[1] += (h >= 35) && (h <= 85) && (s >= 40) && (v >= 30);           // Vert Sain
            cnt[2] += (h >= 20) && (h <= 40) && (s >= 25) && (v >= 30);           // Vert Jaunatre
            cnt[3] += (h >= 15) && (h <= 35) && (s >= 50) && (v >= 50);           // Jaune
            cnt[4] += ((h <= 25) || (h >= 160)) && (s >= 30) && (v >= 20);        // Brun/Orange
            cnt[5] += (((h >= 160) && (h <= 180)) || (h <= 10)) && (s >= 40) && (v >= 30);  // Rouge
            cnt[6] += (v <= 50) && (s >= 20);                                     // Zones Sombres
            cnt[7] += (v >= 200) && (s <= 30);                                    // Zones Claires
            cnt[8] += (h >= 120) && (h <= 160) && (s >= 20);                      // Violet/Pourpre
            cnt[9] += (h >= 35) && (h <= 85);                                     // hue: Vert
            cnt[10] += (h >= 15) && (h <= 35);                                    // hue: Jaune/Orange
            cnt[11] += (h <= 15) || (h >= 160);                                   // hue: Rouge
            cnt[12] += (h >= 120) && (h <= 160);                                  // hue: Violet
            cnt[13] += ((h > 85) && (h < 120));  // hue: Autres ((h>35)&(h<15) is empty)
        }
    }
#pragma unroll
    for (int i = 0; i < LF_HSV_NCOUNTS; ++i) {
        unsigned v = cnt[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&lc[i], v);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 768; i += kBlock)
        if (lh[i]) atomicAdd(&hsv_hist[(size_t)n * 768 + i], (int32_t)lh[i]);
    if (threadIdx.x < LF_HSV_NCOUNTS && lc[threadIdx.x])
        atomicAdd(&counts[(size_t)n * LF_HSV_NCOUNTS + threadIdx.x], (int32_t)lc[threadIdx.x]);
}

// ---------------------------------------------------------------------------
// Gaussian blur, uint8, OpenCV fixed-point semantics, both passes fused via LDS
// ---------------------------------------------------------------------------
// Tile = kBT x kBT output pixels.  Stage (T+2r)x(T+2r) input with BORDER_REFLECT_101,
// horizontal pass -> Q8.8 uint16 rows in LDS, vertical pass -> (acc + 2^15) >> 16.
constexpr int kBT = 32;
constexpr int kMaxR = 15;
using lf::BlurTaps;  // kernel argument: the Q8.8 taps live in scalar registers

__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

template <int CH>
__global__ __launch_bounds__(kBlock) void gauss_blur_kernel(const uint8_t* __restrict__ in,
                                                            uint8_t* __restrict__ out, int h,
                                                            int w, BlurTaps taps, int ksize) {
    const uint16_t* kq = taps.k;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int r = ksize / 2;
    const int pw = kBT + 2 * r;  // patch width/height in pixels
    uint8_t* patch = smem;                                                     // [pw][pw*CH]
    uint16_t* mid = reinterpret_cast<uint16_t*>(smem + ((pw * pw * CH + 15) & ~15));  // [pw][kBT*CH]
    __shared__ unsigned kc[2 * kMaxR + 1];
    if (threadIdx.x < ksize) kc[threadIdx.x] = kq[threadIdx.x];
    const unsigned n = blockIdx.z;
    const int x0 = blockIdx.x * kBT, y0 = blockIdx.y * kBT;
    const uint8_t* src = in + (size_t)n * h * w * CH;
    uint8_t* dst = out + (size_t)n * h * w * CH;
    for (int i = threadIdx.x; i < pw * pw; i += kBlock) {
        const int py = i / pw, px = i - py * pw;
        const int sy = reflect101(y0 + py - r, h), sx = reflect101(x0 + px - r, w);
#pragma unroll
        for (int c = 0; c < CH; ++c) patch[i * CH + c] = src[((size_t)sy * w + sx) * CH + c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < pw * kBT * CH; i += kBlock) {
        const int py = i / (kBT * CH), rem = i - py * (kBT * CH);
        const uint8_t* p = patch + py * pw * CH + rem;  // window start (x - r) for this channel
        unsigned acc = 0;
        for (int k = 0; k < ksize; ++k) acc += kc[k] * p[k * CH];
        mid[i] = (uint16_t)acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kBT * kBT * CH; i += kBlock) {
        const int ty = i / (kBT * CH), rem = i - ty * (kBT * CH);
        const int tx = rem / CH;
        if (y0 + ty >= h || x0 + tx >= w) continue;
        const uint16_t* p = mid + ty * (kBT * CH) + rem;
        unsigned acc = 0;
        for (int k = 0; k < ksize; ++k) acc += kc[k] * p[k * kBT * CH];
        dst[((size_t)(y0 + ty) * w + x0) * CH + rem] = (uint8_t)((acc + (1u << 15)) >> 16);
    }
}

// ---------------------------------------------------------------------------
// batch gather: dst[i] = src[index[i]] for rows of row_bytes bytes (device-resident dataset)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(const uint8_t* __restrict__ src,
                                                             const int32_t* __restrict__ index,
                                                             uint8_t* __restrict__ dst,
                                                             size_t row_bytes, int vec16) {
    const unsigned i = blockIdx.y;
    const uint8_t* s = src + (size_t)index[i] * row_bytes;
    uint8_t* d = dst + (size_t)i * row_bytes;
    if (vec16) {  // rows are multiples of 16 bytes and both bases 16-byte aligned
        const uint4* s4 = reinterpret_cast<const uint4*>(s);
        uint4* d4 = reinterpret_cast<uint4*>(d);
        const size_t n16 = row_bytes / 16;
        for (size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x; q < n16; q += (size_t)gridDim.x * kBlock)
            d4[q] = s4[q];
    } else {
        for (size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x; q < row_bytes;
             q += (size_t)gridDim.x * kBlock)
            d[q] = s[q];
    }
}

// Fast path for the usual kernels (odd ksize 3..15, every Q8.8 tap <= 255).  Same arithmetic as
// gauss_blur_kernel, reorganised for the integer dot-product units:
//   1. the (TY+2r) x (TXP+2r) patch is loaded once (12-byte unaligned groups), de-interleaved
//      into one byte plane per channel in LDS, BORDER_REFLECT_101 applied while loading;
//   2. horizontal pass: a thread makes 4 pixels of two consecutive rows with v_dot4_u32_u8 on
//      byte-aligned windows (v_alignbyte) and stores them as (row 2p | row 2p+1 << 16) pairs;
//   3. vertical pass: v_dot2_u32_u16 over those row pairs with (even, odd) tap pairs, the
//      accumulator seeded with 2^15 so that the result is simply byte 2;
//   4. planes are re-interleaved and stored 12 bytes per thread.
struct __attribute__((packed)) Bytes12 {
    unsigned a, b, c;
};
struct __attribute__((packed)) Bytes4 {
    unsigned a;
};
typedef unsigned short lf_us2 __attribute__((ext_vector_type(2)));
constexpr int kFTX = 64, kFTY = 32;

template <int CH, int KS>
__global__ __launch_bounds__(kBlock) void gauss_blur_fast_kernel(const uint8_t* __restrict__ in,
                                                                 uint8_t* __restrict__ out, int h,
                                                                 int w, BlurTaps taps, int n_images) {
    const uint16_t* kq = taps.k;
    constexpr int R = KS / 2, PR = kFTY + 2 * R;
    constexpr int NG = (kFTX + 2 * R + 3) / 4;  // 4-pixel groups per patch row
    constexpr int PITCH = 4 * (NG + 1);         // + one dword the shifted windows may touch
    constexpr int NK = (KS + 3) / 4, NP = (KS + 1) / 2;
    constexpr int PLANE = PR * PITCH;
    constexpr int MIDP = (PR / 2) * kFTX;  // dwords per channel
    static_assert(PR % 2 == 0 && CH * kFTY * kFTX <= CH * PLANE, "tile geometry");
    __shared__ __attribute__((aligned(16))) unsigned char patch[CH * PLANE];
    __shared__ __attribute__((aligned(16))) unsigned midT[CH * MIDP];
    unsigned char* stage = patch;  // [CH][kFTY][kFTX], reuses the patch after the horizontal pass

    const lf::TileId tile = lf::xcd_tile((w + kFTX - 1) / kFTX, (h + kFTY - 1) / kFTY, n_images);
    if (!tile.ok) return;
    const unsigned n = (unsigned)tile.n;
    const int x0 = tile.tx * kFTX, y0 = tile.ty * kFTY;
    const uint8_t* src = in + (size_t)n * h * w * CH;
    uint8_t* dst = out + (size_t)n * h * w * CH;

    // taps packed for the dot products (uniform -> scalar registers)
    unsigned k4[NK], ke[NP], ko[NP];
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        unsigned v = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (4 * j + t < KS) v |= (unsigned)kq[4 * j + t] << (8 * t);
        k4[j] = v;
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        ke[j] = (unsigned)kq[2 * j] | (2 * j + 1 < KS ? (unsigned)kq[2 * j + 1] << 16 : 0u);
        ko[j] = (j > 0 ? (unsigned)kq[2 * j - 1] : 0u) | ((unsigned)kq[2 * j] << 16);
    }

    // ---- 1. patch -> byte planes
    for (int i = threadIdx.x; i < PR * NG; i += kBlock) {
        const int py = i / NG, g = i - py * NG;
        const int sy = reflect101(y0 + py - R, h);
        const int gx0 = x0 - R + 4 * g;
        unsigned pl[CH];
        if (gx0 >= 0 && gx0 + 3 < w) {
            const uint8_t* s = src + ((size_t)sy * w + gx0) * CH;
            if (CH == 3) {
                const Bytes12 v = *reinterpret_cast<const Bytes12*>(s);
                pl[0] = byte_of(v.a, 0) | byte_of(v.a, 3) << 8 | byte_of(v.b, 2) << 16 | byte_of(v.c, 1) << 24;
                pl[CH > 1 ? 1 : 0] = byte_of(v.a, 1) | byte_of(v.b, 0) << 8 | byte_of(v.b, 3) << 16 | byte_of(v.c, 2) << 24;
                pl[CH > 2 ? 2 : 0] = byte_of(v.a, 2) | byte_of(v.b, 1) << 8 | byte_of(v.c, 0) << 16 | byte_of(v.c, 3) << 24;
            } else {
                pl[0] = reinterpret_cast<const Bytes4*>(s)->a;
            }
        } else {
#pragma unroll
            for (int c = 0; c < CH; ++c) pl[c] = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int sx = reflect101(gx0 + t, w);
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    pl[c] |= (unsigned)src[((size_t)sy * w + sx) * CH + c] << (8 * t);
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c)
            *reinterpret_cast<unsigned*>(patch + c * PLANE + py * PITCH + 4 * g) = pl[c];
    }
    __syncthreads();

    // ---- 2. horizontal pass -> row-pair packed Q8.8
    for (int i = threadIdx.x; i < CH * (PR / 2) * (kFTX / 4); i += kBlock) {
        const int c = i / ((PR / 2) * (kFTX / 4)), rem = i - c * ((PR / 2) * (kFTX / 4));
        const int rp = rem / (kFTX / 4), g = rem - rp * (kFTX / 4);
        unsigned res[2][4];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const unsigned* row =
                reinterpret_cast<const unsigned*>(patch + c * PLANE + (2 * rp + rr) * PITCH + 4 * g);
            unsigned d[NK + 1];
#pragma unroll
            for (int j = 0; j <= NK; ++j) d[j] = row[j];
#pragma unroll
            for (int sft = 0; sft < 4; ++sft) {
                unsigned acc = 0;
#pragma unroll
                for (int j = 0; j < NK; ++j) {
                    const unsigned a = sft == 0 ? d[j] : __builtin_amdgcn_alignbyte(d[j + 1], d[j], sft);
                    acc = __builtin_amdgcn_udot4(a, k4[j], acc, false);
                }
                res[rr][sft] = acc;  // <= 255 * 256: fits 16 bits
            }
        }
        uint4 o;
        o.x = res[0][0] | res[1][0] << 16;
        o.y = res[0][1] | res[1][1] << 16;
        o.z = res[0][2] | res[1][2] << 16;
        o.w = res[0][3] | res[1][3] << 16;
        *reinterpret_cast<uint4*>(midT + c * MIDP + rp * kFTX + 4 * g) = o;
    }
    __syncthreads();

    // ---- 3. vertical pass -> byte planes of the output tile (two row pairs per item)
    constexpr int RB = 2;
    for (int i = threadIdx.x; i < CH * (kFTX / 4) * (kFTY / 2 / RB); i += kBlock) {
        const int c = i / ((kFTX / 4) * (kFTY / 2 / RB)), rem = i - c * ((kFTX / 4) * (kFTY / 2 / RB));
        const int yb = rem / (kFTX / 4), g = rem - yb * (kFTX / 4);
        uint4 m[RB + NP - 1];
#pragma unroll
        for (int j = 0; j < RB + NP - 1; ++j)
            m[j] = *reinterpret_cast<const uint4*>(midT + c * MIDP + (yb * RB + j) * kFTX + 4 * g);
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            unsigned ev[4] = {1u << 15, 1u << 15, 1u << 15, 1u << 15};
            unsigned od[4] = {1u << 15, 1u << 15, 1u << 15, 1u << 15};
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const unsigned mv[4] = {m[b + j].x, m[b + j].y, m[b + j].z, m[b + j].w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    ev[t] = __builtin_amdgcn_udot2(__builtin_bit_cast(lf_us2, mv[t]),
                                                   __builtin_bit_cast(lf_us2, ke[j]), ev[t], false);
                    od[t] = __builtin_amdgcn_udot2(__builtin_bit_cast(lf_us2, mv[t]),
                                                   __builtin_bit_cast(lf_us2, ko[j]), od[t], false);
                }
            }
            const unsigned oe = byte_of(ev[0], 2) | byte_of(ev[1], 2) << 8 | byte_of(ev[2], 2) << 16 |
                                byte_of(ev[3], 2) << 24;
            const unsigned oo = byte_of(od[0], 2) | byte_of(od[1], 2) << 8 | byte_of(od[2], 2) << 16 |
                                byte_of(od[3], 2) << 24;
            const int ty = 2 * (yb * RB + b);
            *reinterpret_cast<unsigned*>(stage + (c * kFTY + ty) * kFTX + 4 * g) = oe;
            *reinterpret_cast<unsigned*>(stage + (c * kFTY + ty + 1) * kFTX + 4 * g) = oo;
        }
    }
    __syncthreads();

    // ---- 4. interleave and store
    for (int i = threadIdx.x; i < kFTY * (kFTX / 4); i += kBlock) {
        const int ty = i / (kFTX / 4), g = i - ty * (kFTX / 4);
        const int gy = y0 + ty, gx = x0 + 4 * g;
        if (gy >= h || gx >= w) continue;
        unsigned pl[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c)
            pl[c] = *reinterpret_cast<const unsigned*>(stage + (c * kFTY + ty) * kFTX + 4 * g);
        uint8_t* d = dst + ((size_t)gy * w + gx) * CH;
        if (gx + 3 < w) {
            if (CH == 3) {
                const unsigned r4 = pl[0], g4 = pl[CH > 1 ? 1 : 0], b4 = pl[CH > 2 ? 2 : 0];
                Bytes12 o;
                o.a = byte_of(r4, 0) | byte_of(g4, 0) << 8 | byte_of(b4, 0) << 16 | byte_of(r4, 1) << 24;
                o.b = byte_of(g4, 1) | byte_of(b4, 1) << 8 | byte_of(r4, 2) << 16 | byte_of(g4, 2) << 24;
                o.c = byte_of(b4, 2) | byte_of(r4, 3) << 8 | byte_of(g4, 3) << 16 | byte_of(b4, 3) << 24;
                *reinterpret_cast<Bytes12*>(d) = o;
            } else {
                reinterpret_cast<Bytes4*>(d)->a = pl[0];
            }
        } else {
            for (int t = 0; gx + t < w; ++t)
#pragma unroll
                for (int c = 0; c < CH; ++c) d[t * CH + c] = (uint8_t)byte_of(pl[c], t);
        }
    }
}

template <int CH>
bool launch_blur_fast(const uint8_t* in, uint8_t* out, int n, int h, int w, const BlurTaps& kq,
                      int ksize, hipStream_t s) {
    const unsigned grid = lf::xcd_grid((size_t)((w + kFTX - 1) / kFTX) * ((h + kFTY - 1) / kFTY) * n);
    switch (ksize) {
        case 3: gauss_blur_fast_kernel<CH, 3><<<grid, kBlock, 0, s>>>(in, out, h, w, kq, n); return true;
        case 5: gauss_blur_fast_kernel<CH, 5><<<grid, kBlock, 0, s>>>(in, out, h, w, kq, n); return true;
        case 7: gauss_blur_fast_kernel<CH, 7><<<grid, kBlock, 0, s>>>(in, out, h, w, kq, n); return true;
        case 9: gauss_blur_fast_kernel<CH, 9><<<grid, kBlock, 0, s>>>(in, out, h, w, kq, n); return true;
        case 11: gauss_blur_fast_kernel<CH, 11><<<grid, kBlock, 0, s>>>(in, out, h, w, kq, n); return true;
        case 13: gauss_blur_fast_kernel<CH, 13><<<grid, kBlock, 0, s>>>(in, out, h, w, kq, n); return true;
        case 15: gauss_blur_fast_kernel<CH, 15><<<grid, kBlock, 0, s>>>(in, out, h, w, kq, n); return true;
        default: return false;
    }
}

}  // namespace

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

int lf_pack_hwc_u8_to_nchw_f32(const uint8_t* in, float* out, int n, int h, int w,
                               const float* mean3, const float* denom3, lf_stream_t stream) {
    LF_REQUIRE(in && out, "lf_pack: null buffer");
    LF_REQUIRE(n > 0 && h > 0 && w > 0, "lf_pack: bad dims n=%d h=%d w=%d", n, h, w);
    LF_REQUIRE((mean3 == nullptr) == (denom3 == nullptr), "lf_pack: mean/denom must both be set");
    const size_t hw = (size_t)h * w;
    float m[3] = {0, 0, 0}, d[3] = {1, 1, 1};
    const bool norm = mean3 != nullptr;
    if (norm) {
        // mean/denom are HOST pointers (3 floats each): they are model constants.
        for (int c = 0; c < 3; ++c) {
            m[c] = mean3[c];
            d[c] = denom3[c];
        }
    }
    hipStream_t s = lf::as_stream(stream);
    if (hw % 4 == 0) {
        dim3 grid(lf::stream_grid(hw / 4, kBlock, 1024), n);
        const bool nt = lf::streaming((size_t)n * hw * 15);
        auto kern = norm ? (nt ? pack_kernel<true, true> : pack_kernel<true, false>)
                         : (nt ? pack_kernel<false, true> : pack_kernel<false, false>);
        kern<<<grid, kBlock, 0, s>>>(in, out, (unsigned)(hw / 4), m[0], m[1], m[2], d[0], d[1], d[2]);
    } else {
        dim3 grid(lf::stream_grid(hw, kBlock, 64), n);
        if (norm)
            pack_scalar_kernel<true><<<grid, kBlock, 0, s>>>(in, out, hw, m[0], m[1], m[2], d[0],
                                                             d[1], d[2]);
        else
            pack_scalar_kernel<false><<<grid, kBlock, 0, s>>>(in, out, hw, m[0], m[1], m[2], d[0],
                                                              d[1], d[2]);
    }
    return lf::check_launch("lf_pack");
}

int lf_hist_u8(const uint8_t* in, int32_t* hist, int n, int h, int w, lf_stream_t stream) {
    LF_REQUIRE(in && hist, "lf_hist: null buffer");
    LF_REQUIRE(n > 0 && h > 0 && w > 0, "lf_hist: bad dims n=%d h=%d w=%d", n, h, w);
    hipStream_t s = lf::as_stream(stream);
    const size_t nbytes = (size_t)h * w * 3;
    if (n >= 512) {
        // enough images to fill the chip with one workgroup per image: the slotted table, no memset, no
        // global atomics
        const unsigned grid = (unsigned)(n < 256 * 12 ? n : 256 * 12);
        if (lf::streaming((size_t)n * nbytes))
            hist_slot_kernel<true><<<grid, kBlock, 0, s>>>(in, hist, nbytes, n);
        else
            hist_slot_kernel<false><<<grid, kBlock, 0, s>>>(in, hist, nbytes, n);
        return lf::check_launch("lf_hist");
    }
    if (hipMemsetAsync(hist, 0, (size_t)n * 768 * sizeof(int32_t), s) != hipSuccess) {
        lf::set_error("lf_hist: memset failed");
        return LF_ERR_LAUNCH;
    }
    // enough workgroups per image that small batches still fill the chip
    unsigned splits = n >= 2048 ? 1 : (unsigned)((2048 + n - 1) / n);
    const unsigned maxs = lf::stream_grid(nbytes / 16 + 1, kBlock, 64);
    if (splits > maxs) splits = maxs;
    dim3 grid(splits, n);
    if (lf::streaming((size_t)n * nbytes))
        hist_kernel<true><<<grid, kBlock, 0, s>>>(in, hist, nbytes);
    else
        hist_kernel<false><<<grid, kBlock, 0, s>>>(in, hist, nbytes);
    return lf::check_launch("lf_hist");
}

int lf_autocontrast_lut(const int32_t* hist, const double* cutoff, uint8_t* lut, int n,
                        lf_stream_t stream) {
    LF_REQUIRE(hist && cutoff && lut, "lf_autocontrast_lut: null buffer");
    LF_REQUIRE(n > 0, "lf_autocontrast_lut: bad n=%d", n);
    const int total = n * 3;
    autocontrast_lut_kernel<<<(total + 63) / 64, 64, 0, lf::as_stream(stream)>>>(hist, cutoff, lut,
                                                                               total);
    return lf::check_launch("lf_autocontrast_lut");
}

int lf_lut_apply_u8(const uint8_t* in, const uint8_t* lut, uint8_t* out, int n, int h, int w,
                    lf_stream_t stream) {
    LF_REQUIRE(in && lut && out, "lf_lut_apply: null buffer");
    LF_REQUIRE(n > 0 && h > 0 && w > 0, "lf_lut_apply: bad dims n=%d h=%d w=%d", n, h, w);
    const size_t nbytes = (size_t)h * w * 3;
    hipStream_t s = lf::as_stream(stream);
    unsigned splits = n >= 2048 ? 1 : (unsigned)((2048 + n - 1) / n);
    const bool aligned = nbytes % 16 == 0 && ((reinterpret_cast<size_t>(in) |
                                               reinterpret_cast<size_t>(out)) & 15) == 0;
    if (aligned) {
        const unsigned maxs = lf::stream_grid(nbytes / 16, kBlock, 64);
        const unsigned few = (maxs + 3) / 4;  // about four 16-byte chunks per thread
        if (splits < few) splits = few;
        if (splits > maxs) splits = maxs;
        if (lf::streaming((size_t)n * nbytes * 2))
            lut_apply_kernel<true><<<dim3(splits, n), kBlock, 0, s>>>(in, lut, out, nbytes);
        else
            lut_apply_kernel<false><<<dim3(splits, n), kBlock, 0, s>>>(in, lut, out, nbytes);
    } else {
        const unsigned maxs = lf::stream_grid(nbytes, kBlock, 64);
        if (splits > maxs) splits = maxs;
        lut_apply_scalar_kernel<<<dim3(splits, n), kBlock, 0, s>>>(in, lut, out, nbytes);
    }
    return lf::check_launch("lf_lut_apply");
}

int lf_gather_rows_u8(const uint8_t* src, const int32_t* index, uint8_t* dst, int n_out,
                      size_t row_bytes, lf_stream_t stream) {
    LF_REQUIRE(src && index && dst, "lf_gather_rows: null buffer");
    LF_REQUIRE(n_out > 0 && n_out <= 65535 && row_bytes > 0, "lf_gather_rows: bad n_out=%d row_bytes=%zu",
               n_out, row_bytes);
    const int vec16 = row_bytes % 16 == 0 &&
                      ((reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(dst)) & 15) == 0;
    dim3 grid(lf::stream_grid(vec16 ? row_bytes / 16 : row_bytes, kBlock, 16), n_out);
    gather_rows_kernel<<<grid, kBlock, 0, lf::as_stream(stream)>>>(src, index, dst, row_bytes, vec16);
    return lf::check_launch("lf_gather_rows");
}

int lf_flip_u8(const uint8_t* in, uint8_t* out, const int32_t* mode, int n, int h, int w,
               lf_stream_t stream) {
    LF_REQUIRE(in && out && mode, "lf_flip: null buffer");
    LF_REQUIRE(n > 0 && h > 0 && w > 0, "lf_flip: bad dims n=%d h=%d w=%d", n, h, w);
    LF_REQUIRE(in != out, "lf_flip: in-place flip is not supported");
    hipStream_t s = lf::as_stream(stream);
    const bool aligned = w % 4 == 0 && ((reinterpret_cast<size_t>(in) |
                                         reinterpret_cast<size_t>(out)) & 3) == 0;
    if (aligned) {
        dim3 grid(lf::stream_grid((size_t)h * (w / 4), kBlock, 64), n);
        // plain accesses, destination order: nontemporal or source-ordered variants measured
        // within noise of this one (5.6-5.8 TB/s) or slower
        flip_kernel<<<grid, kBlock, 0, s>>>(in, out, mode, h, w / 4);
    } else {
        dim3 grid(lf::stream_grid((size_t)h * w, kBlock, 64), n);
        flip_scalar_kernel<<<grid, kBlock, 0, s>>>(in, out, mode, h, w);
    }
    return lf::check_launch("lf_flip");
}

int lf_noise_wrap_add_u8(const uint8_t* in, const double* noise, uint8_t* out, size_t nbytes,
                         lf_stream_t stream) {
    LF_REQUIRE(in && noise && out, "lf_noise_wrap_add: null buffer");
    LF_REQUIRE(nbytes > 0, "lf_noise_wrap_add: empty input");
    noise_add_kernel<<<lf::stream_grid(nbytes, kBlock), kBlock, 0, lf::as_stream(stream)>>>(
        in, noise, out, nbytes);
    return lf::check_launch("lf_noise_wrap_add");
}

int lf_add_wrap_u8(const uint8_t* in, const uint8_t* add, uint8_t* out, size_t nbytes, lf_stream_t stream) {
    LF_REQUIRE(in && add && out, "lf_add_wrap_u8: null buffer");
    LF_REQUIRE(nbytes > 0 && nbytes % 4 == 0, "lf_add_wrap_u8: size must be a positive multiple of 4 bytes");
    LF_REQUIRE(((reinterpret_cast<size_t>(in) | reinterpret_cast<size_t>(add) | reinterpret_cast<size_t>(out)) & 3) == 0,
               "lf_add_wrap_u8: buffers must be 4-byte aligned");
    add_wrap_u8_kernel<<<lf::stream_grid(nbytes / 4, kBlock, lf::kFullGrid), kBlock, 0, lf::as_stream(stream)>>>(
        reinterpret_cast<const uint32_t*>(in), reinterpret_cast<const uint32_t*>(add),
        reinterpret_cast<uint32_t*>(out), nbytes / 4);
    return lf::check_launch("lf_add_wrap_u8");
}

int lf_noise_philox_add_u8(const uint8_t* in, uint8_t* out, size_t nbytes, uint64_t seed,
                           float sigma, lf_stream_t stream) {
    LF_REQUIRE(in && out, "lf_noise_philox_add: null buffer");
    LF_REQUIRE(nbytes > 0, "lf_noise_philox_add: empty input");
    LF_REQUIRE(((reinterpret_cast<size_t>(in) | reinterpret_cast<size_t>(out)) & 3) == 0,
               "lf_noise_philox_add: buffers must be 4-byte aligned");
    const size_t nwords = (nbytes + 3) / 4;
    const unsigned grid = lf::stream_grid(nwords, kBlock, lf::kFullGrid);
    if (lf::streaming(nbytes * 2))
        noise_philox_kernel<true><<<grid, kBlock, 0, lf::as_stream(stream)>>>(in, out, nwords, nbytes,
                                                                             seed, sigma);
    else
        noise_philox_kernel<false><<<grid, kBlock, 0, lf::as_stream(stream)>>>(in, out, nwords,
                                                                              nbytes, seed, sigma);
    return lf::check_launch("lf_noise_philox_add");
}

int lf_noise_hist_u8(const uint8_t* in, const uint8_t* add, uint8_t* out, int32_t* hist, int n, int h, int w,
                     uint64_t seed, float sigma, lf_stream_t stream) {
    LF_REQUIRE(in && out && hist, "lf_noise_hist_u8: null buffer");
    LF_REQUIRE(n > 0 && h > 0 && w > 0, "lf_noise_hist_u8: bad dims n=%d h=%d w=%d", n, h, w);
    const size_t nbytes = (size_t)h * w * 3;
    LF_REQUIRE(nbytes % 16 == 0, "lf_noise_hist_u8: an image must be a multiple of 16 bytes (got %zu)", nbytes);
    LF_REQUIRE(((reinterpret_cast<size_t>(in) | reinterpret_cast<size_t>(add) | reinterpret_cast<size_t>(out)) & 15) == 0,
               "lf_noise_hist_u8: buffers must be 16-byte aligned");
    const unsigned grid = (unsigned)std::min<size_t>((size_t)n, (size_t)256 * 8);
    if (add != nullptr)
        noise_hist_kernel<true><<<grid, kBlock, 0, lf::as_stream(stream)>>>(in, add, out, hist, nbytes, n, seed, sigma);
    else
        noise_hist_kernel<false><<<grid, kBlock, 0, lf::as_stream(stream)>>>(in, add, out, hist, nbytes, n, seed, sigma);
    return lf::check_launch("lf_noise_hist_u8");
}

int lf_mask_composite_u8(const uint8_t* img, const uint8_t* mask, uint8_t* out, int n, int h,
                         int w, int color, lf_stream_t stream) {
    LF_REQUIRE(img && mask && out, "lf_mask_composite: null buffer");
    LF_REQUIRE(n > 0 && h > 0 && w > 0, "lf_mask_composite: bad dims n=%d h=%d w=%d", n, h, w);
    LF_REQUIRE(color == 0 || color == 255, "lf_mask_composite: color must be 0 or 255");
    const size_t npx = (size_t)n * h * w;
    hipStream_t s = lf::as_stream(stream);
    const bool aligned = ((reinterpret_cast<size_t>(img) | reinterpret_cast<size_t>(mask) |
                           reinterpret_cast<size_t>(out)) & 3) == 0;
    const size_t npx4 = aligned ? npx / 4 : 0;
    if (npx4) {
        const unsigned grid = lf::stream_grid(npx4, kBlock, lf::kFullGrid);
        if (lf::streaming(npx * 7))
            composite_kernel<true><<<grid, kBlock, 0, s>>>(img, mask, out, npx4, (unsigned)color);
        else
            composite_kernel<false><<<grid, kBlock, 0, s>>>(img, mask, out, npx4, (unsigned)color);
    }
    if (npx4 * 4 < npx)
        composite_scalar_kernel<<<lf::stream_grid(npx - npx4 * 4, kBlock), kBlock, 0, s>>>(
            img, mask, out, npx4 * 4, npx, (unsigned)color);
    return lf::check_launch("lf_mask_composite");
}

int lf_rgb2hsv_u8(const uint8_t* rgb, uint8_t* hsv, size_t npixels, lf_stream_t stream) {
    LF_REQUIRE(rgb && hsv, "lf_rgb2hsv: null buffer");
    LF_REQUIRE(npixels > 0, "lf_rgb2hsv: empty input");
    LF_REQUIRE(((reinterpret_cast<size_t>(rgb) | reinterpret_cast<size_t>(hsv)) & 3) == 0,
               "lf_rgb2hsv: buffers must be 4-byte aligned");
    // four 4-pixel groups per thread: the per-workgroup divide tables stay a small share
    const unsigned grid = lf::stream_grid(npixels / 16 + 1, kBlock, lf::kFullGrid);
    if (lf::streaming(npixels * 6))
        rgb2hsv_kernel<true><<<grid, kBlock, 0, lf::as_stream(stream)>>>(rgb, hsv, npixels);
    else
        rgb2hsv_kernel<false><<<grid, kBlock, 0, lf::as_stream(stream)>>>(rgb, hsv, npixels);
    return lf::check_launch("lf_rgb2hsv");
}

int lf_rgb2gray_u8(const uint8_t* rgb, uint8_t* gray, size_t npixels, lf_stream_t stream) {
    LF_REQUIRE(rgb && gray, "lf_rgb2gray: null buffer");
    LF_REQUIRE(npixels > 0, "lf_rgb2gray: empty input");
    LF_REQUIRE(((reinterpret_cast<size_t>(rgb) | reinterpret_cast<size_t>(gray)) & 3) == 0,
               "lf_rgb2gray: buffers must be 4-byte aligned");
    const unsigned grid = lf::stream_grid(npixels / 4 + 1, kBlock, lf::kFullGrid);
    if (lf::streaming(npixels * 4))
        rgb2gray_kernel<true><<<grid, kBlock, 0, lf::as_stream(stream)>>>(rgb, gray, npixels);
    else
        rgb2gray_kernel<false><<<grid, kBlock, 0, lf::as_stream(stream)>>>(rgb, gray, npixels);
    return lf::check_launch("lf_rgb2gray");
}

int lf_hsv_region_stats(const uint8_t* rgb, int32_t* counts, int32_t* hsv_hist, int n, int h,
                        int w, lf_stream_t stream) {
    LF_REQUIRE(rgb && counts && hsv_hist, "lf_hsv_region_stats: null buffer");
    LF_REQUIRE(n > 0 && h > 0 && w > 0, "lf_hsv_region_stats: bad dims n=%d h=%d w=%d", n, h, w);
    hipStream_t s = lf::as_stream(stream);
    if (hipMemsetAsync(counts, 0, (size_t)n * LF_HSV_NCOUNTS * sizeof(int32_t), s) != hipSuccess ||
        hipMemsetAsync(hsv_hist, 0, (size_t)n * 768 * sizeof(int32_t), s) != hipSuccess) {
        lf::set_error("lf_hsv_region_stats: memset failed");
        return LF_ERR_LAUNCH;
    }
    const size_t npx = (size_t)h * w;
    unsigned splits = n >= 2048 ? 1 : (unsigned)((2048 + n - 1) / n);
    const unsigned maxs = lf::stream_grid(npx, kBlock, 64);
    if (splits > maxs) splits = maxs;
    hsv_stats_kernel<<<dim3(splits, n), kBlock, 0, s>>>(rgb, counts, hsv_hist, npx);
    return lf::check_launch("lf_hsv_region_stats");
}

int lf_gauss_blur_u8(const uint8_t* in, uint8_t* out, int n, int h, int w, int channels,
                     const uint16_t* kq, int ksize, lf_stream_t stream) {
    LF_REQUIRE(in && out && kq, "lf_gauss_blur: null buffer");
    LF_REQUIRE(n > 0 && h > 0 && w > 0, "lf_gauss_blur: bad dims n=%d h=%d w=%d", n, h, w);
    LF_REQUIRE(channels == 1 || channels == 3, "lf_gauss_blur: channels must be 1 or 3");
    LF_REQUIRE(ksize >= 1 && ksize <= 2 * kMaxR + 1 && (ksize & 1), "lf_gauss_blur: bad ksize %d",
               ksize);
    LF_REQUIRE(in != out, "lf_gauss_blur: in-place blur is not supported");
    LF_REQUIRE(n <= 65535, "lf_gauss_blur: batch too large for grid.z");
    BlurTaps taps{};
    bool fit_u8 = ksize >= 3;
    for (int i = 0; i < ksize; ++i) {  // kq is a HOST array (it is a handful of model constants)
        taps.k[i] = kq[i];
        fit_u8 = fit_u8 && kq[i] <= 255;
    }
    hipStream_t s = lf::as_stream(stream);
    if (lf::blur_mfma_launch(in, out, n, h, w, channels, taps, ksize, s)) return lf::check_launch("lf_gauss_blur");
    if (fit_u8 && (channels == 3 ? launch_blur_fast<3>(in, out, n, h, w, taps, ksize, s)
                                 : launch_blur_fast<1>(in, out, n, h, w, taps, ksize, s)))
        return lf::check_launch("lf_gauss_blur");
    const int pw = kBT + (ksize / 2) * 2;
    const size_t lds = ((size_t)(pw * pw * channels + 15) & ~(size_t)15) +
                       (size_t)pw * kBT * channels * sizeof(uint16_t);
    dim3 grid((w + kBT - 1) / kBT, (h + kBT - 1) / kBT, n);
    if (channels == 3)
        gauss_blur_kernel<3><<<grid, kBlock, lds, s>>>(in, out, h, w, taps, ksize);
    else
        gauss_blur_kernel<1><<<grid, kBlock, lds, s>>>(in, out, h, w, taps, ksize);
    return lf::check_launch("lf_gauss_blur");
}

}  // extern "C"

extern "C" int lf_copy_rows(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width, size_t rows,
                            int to_host, lf_stream_t stream) {
    LF_REQUIRE(dst && src, "lf_copy_rows: null buffer");
    LF_REQUIRE(width > 0 && rows > 0 && width <= dst_pitch && width <= src_pitch, "lf_copy_rows: bad extent");
    const hipError_t e = hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width, rows,
                                          to_host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice, lf::as_stream(stream));
    if (e != hipSuccess) {
        lf::set_error("lf_copy_rows: %s", hipGetErrorString(e));
        return LF_ERR_LAUNCH;
    }
    return LF_OK;
}
