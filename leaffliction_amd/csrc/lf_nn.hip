// libleafhip — leaf_cnn non-conv kernels (fp32, NCHW): input stage (pack + in-model
// augmentation + Normalization), BatchNorm (train statistics / backward), Squeeze-Excite,
// residual tail (add + ReLU + SpatialDropout + MaxPool) forward/backward, GAP + Dense +
// softmax cross-entropy head, AdamW with per-tensor clipnorm + EMA.
//
// All of these are HBM-bound streaming / reduction kernels: float4 accesses, one
// (n, c) plane per workgroup row so per-channel / per-sample parameters are uniform, and
// every cross-workgroup reduction goes through partial sums that are combined in a fixed
// order (deterministic; no float atomics).  Reference semantics: srcs/model/cnn.py:9-104,
// srcs/train/utils.py:17-57 (Keras 3 layer / optimizer definitions).
#include "lf_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kBnSplit = 64;  // partial sums per channel in the BN reductions

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// block-wide sum of up to 3 values; result valid in thread 0
template <int K>
__device__ __forceinline__ void block_sum(float (&v)[K], float* red /* [K][4] */) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        v[k] = wave_sum(v[k]);
        if (lane == 0) red[k * 4 + wid] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = red[k * 4] + red[k * 4 + 1] + red[k * 4 + 2] + red[k * 4 + 3];
    }
}

// ---------------------------------------------------------------------------
// input stage: u8 HWC -> [flip, rotate(bilinear, reflect), contrast] -> normalise -> f32 NCHW
// ---------------------------------------------------------------------------
// aug[n] = {flip, cos, sin, contrast}.  Sampling follows keras RandomFlip("horizontal") ->
// RandomRotation (affine about the image centre, bilinear, fill_mode="reflect") ->
// RandomContrast ((x - mean_hw) * f + mean_hw, clipped to [0, 255]) on the [0,1] image.
__device__ __forceinline__ int reflect_idx(int i, int n) {
    // fill_mode="reflect": (d c b a | a b c d | d c b a)
    const int period = 2 * n;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}

__device__ __forceinline__ void sample_aug(const uint8_t* __restrict__ src, int h, int w, int oy,
                                           int ox, float flip, float cs, float sn, float* rgb) {
    // output -> input coordinates (keras RandomRotation's projective matrix)
    const float wm = (float)(w - 1), hm = (float)(h - 1);
    const float xoff = (wm - (cs * wm - sn * hm)) * 0.5f;
    const float yoff = (hm - (sn * wm + cs * hm)) * 0.5f;
    const float fx = cs * ox - sn * oy + xoff;
    const float fy = sn * ox + cs * oy + yoff;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float ax = fx - x0f, ay = fy - y0f;
    int xs[2] = {reflect_idx((int)x0f, w), reflect_idx((int)x0f + 1, w)};
    const int ys[2] = {reflect_idx((int)y0f, h), reflect_idx((int)y0f + 1, h)};
    if (flip != 0.f) {  // the flip precedes the rotation: sample the mirrored source
        xs[0] = w - 1 - xs[0];
        xs[1] = w - 1 - xs[1];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v00 = src[((size_t)ys[0] * w + xs[0]) * 3 + c];
        const float v01 = src[((size_t)ys[0] * w + xs[1]) * 3 + c];
        const float v10 = src[((size_t)ys[1] * w + xs[0]) * 3 + c];
        const float v11 = src[((size_t)ys[1] * w + xs[1]) * 3 + c];
        const float top = v00 + (v01 - v00) * ax, bot = v10 + (v11 - v10) * ax;
        rgb[c] = (top + (bot - top) * ay) * (1.0f / 255.0f);
    }
}

// per-image channel sums of the flipped+rotated image (RandomContrast's mean), in kMeanSplit
// interleaved slices per image: grid = (kMeanSplit, n); input_aug_kernel adds the slices in order
constexpr int kMeanSplit = 8;
__global__ __launch_bounds__(kBlock) void aug_means_kernel(const uint8_t* __restrict__ in,
                                                           const float* __restrict__ aug,
                                                           float* __restrict__ means, int h, int w) {
    __shared__ float red[12];
    const int n = blockIdx.y;
    const uint8_t* src = in + (size_t)n * h * w * 3;
    const float flip = aug[4 * n], cs = aug[4 * n + 1], sn = aug[4 * n + 2];
    float acc[3] = {0.f, 0.f, 0.f};
    for (int p = blockIdx.x * kBlock + threadIdx.x; p < h * w; p += kMeanSplit * kBlock) {
        float rgb[3];
        sample_aug(src, h, w, p / w, p % w, flip, cs, sn, rgb);
        acc[0] += rgb[0];
        acc[1] += rgb[1];
        acc[2] += rgb[2];
    }
    block_sum<3>(acc, red);
    if (threadIdx.x == 0) {
        float* dst = means + ((size_t)n * kMeanSplit + blockIdx.x) * 3;
        dst[0] = acc[0];
        dst[1] = acc[1];
        dst[2] = acc[2];
    }
}

__global__ __launch_bounds__(kBlock) void input_aug_kernel(const uint8_t* __restrict__ in,
                                                           float* __restrict__ out,
                                                           const float* __restrict__ aug,
                                                           const float* __restrict__ means, int h,
                                                           int w, float m0, float m1, float m2,
                                                           float d0, float d1, float d2) {
    const int n = blockIdx.y;
    const size_t hw = (size_t)h * w;
    const uint8_t* src = in + (size_t)n * hw * 3;
    float* dst = out + (size_t)n * hw * 3;
    const float flip = aug[4 * n], cs = aug[4 * n + 1], sn = aug[4 * n + 2], ct = aug[4 * n + 3];
    float mu[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < kMeanSplit; ++k) {
        const float* part = means + ((size_t)n * kMeanSplit + k) * 3;
        mu[0] += part[0];
        mu[1] += part[1];
        mu[2] += part[2];
    }
    {
        const float inv = 1.0f / (float)hw;
        mu[0] *= inv;
        mu[1] *= inv;
        mu[2] *= inv;
    }
    const float nm[3] = {m0, m1, m2}, nd[3] = {d0, d1, d2};
    for (int p = blockIdx.x * kBlock + threadIdx.x; p < (int)hw; p += gridDim.x * kBlock) {
        float rgb[3];
        sample_aug(src, h, w, p / w, p % w, flip, cs, sn, rgb);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = (rgb[c] - mu[c]) * ct + mu[c];
            v = fminf(fmaxf(v, 0.f), 255.f);
            dst[(size_t)c * hw + p] = (v - nm[c]) / nd[c];
        }
    }
}

// ---------------------------------------------------------------------------
// plane-wise elementwise: out = act(x * scale[c] + shift[c])
// ---------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(kBlock) void scale_shift_act_kernel(const float* __restrict__ x,
                                                                 float* __restrict__ out, int c,
                                                                 int hwv,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ shift,
                                                                 int relu) {
    const int plane = blockIdx.x;
    const float sc = scale[plane % c], sh = shift[plane % c];
    const size_t base = (size_t)plane * hwv * VEC;
    for (int i = blockIdx.y * kBlock + threadIdx.x; i < hwv; i += gridDim.y * kBlock) {
        if (VEC == 4) {
            float4 v = reinterpret_cast<const float4*>(x + base)[i];
            v.x = fmaf(v.x, sc, sh); v.y = fmaf(v.y, sc, sh); v.z = fmaf(v.z, sc, sh); v.w = fmaf(v.w, sc, sh);
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            reinterpret_cast<float4*>(out + base)[i] = v;
        } else {
            float v = fmaf(x[base + i], sc, sh);
            out[base + i] = relu ? fmaxf(v, 0.f) : v;
        }
    }
}

// ---------------------------------------------------------------------------
// BatchNorm statistics (training): per-channel sum / sum of squares over N*H*W
// ---------------------------------------------------------------------------
// grid = (kBnSplit, C): workgroup (s, c) reduces the planes n = s, s+kBnSplit, ... of channel c
// around a per-channel pivot (the first element) to keep sum-of-squares well conditioned.
__global__ __launch_bounds__(kBlock) void bn_stats_kernel(const float* __restrict__ y, int n, int c,
                                                          int hw, float* __restrict__ part) {
    __shared__ float red[8];
    const int ch = blockIdx.y, s = blockIdx.x;
    const float pivot = y[(size_t)ch * hw];
    float acc[2] = {0.f, 0.f};
    for (int img = s; img < n; img += kBnSplit) {
        const float* p = y + ((size_t)img * c + ch) * hw;
        if ((hw & 3) == 0) {
            const float4* p4 = reinterpret_cast<const float4*>(p);
            for (int i = threadIdx.x; i < hw / 4; i += kBlock) {
                const float4 v = p4[i];
                const float a = v.x - pivot, b = v.y - pivot, cc = v.z - pivot, d = v.w - pivot;
                acc[0] += (a + b) + (cc + d);
                acc[1] += (a * a + b * b) + (cc * cc + d * d);
            }
        } else {
            for (int i = threadIdx.x; i < hw; i += kBlock) {
                const float a = p[i] - pivot;
                acc[0] += a;
                acc[1] += a * a;
            }
        }
    }
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) {
        part[((size_t)ch * kBnSplit + s) * 2] = acc[0];
        part[((size_t)ch * kBnSplit + s) * 2 + 1] = acc[1];
    }
}

// One thread per channel: combine partials in double, produce mean / invstd / scale / shift and
// update the moving statistics (keras BatchNormalization: biased variance, momentum 0.99).
// grid = (kBnSplit, C): workgroup (s, c) adds slice s of channel c's per-tile partial sums
// (written by the convolution epilogue, lf_conv2d_stats_f32) in a fixed order.
__global__ __launch_bounds__(kBlock) void bn_tile_reduce_kernel(const float* __restrict__ tile_part,
                                                                long long tiles,
                                                                float* __restrict__ part) {
    __shared__ float red[8];
    const int ch = blockIdx.y, s = blockIdx.x;
    const long long per = (tiles + kBnSplit - 1) / kBnSplit;
    const long long t0 = per * s, t1 = t0 + per < tiles ? t0 + per : tiles;
    const float2* src = reinterpret_cast<const float2*>(tile_part) + (size_t)ch * (size_t)tiles;
    float acc[2] = {0.f, 0.f};
    for (long long t = t0 + threadIdx.x; t < t1; t += kBlock) {
        const float2 v = src[t];
        acc[0] += v.x;
        acc[1] += v.y;
    }
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) {
        part[((size_t)ch * kBnSplit + s) * 2] = acc[0];
        part[((size_t)ch * kBnSplit + s) * 2 + 1] = acc[1];
    }
}

// `pivot_src[ch * pivot_stride]` is the value the partial sums were taken about (null = 0).
__global__ void bn_finalize_kernel(const float* __restrict__ pivot_src, int pivot_stride,
                                   const float* __restrict__ part, int c, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ mmean,
                                   float* __restrict__ mvar, float momentum, float eps,
                                   float* __restrict__ mean_o, float* __restrict__ invstd_o,
                                   float* __restrict__ scale_o, float* __restrict__ shift_o) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    double s = 0.0, q = 0.0;
    for (int k = 0; k < kBnSplit; ++k) {
        s += part[((size_t)ch * kBnSplit + k) * 2];
        q += part[((size_t)ch * kBnSplit + k) * 2 + 1];
    }
    const double pivot = pivot_src != nullptr ? (double)pivot_src[(size_t)ch * pivot_stride] : 0.0;
    const double dm = s / count;
    double var = q / count - dm * dm;
    if (var < 0.0) var = 0.0;
    const float mean = (float)(pivot + dm);
    const float fvar = (float)var;
    const float invstd = 1.0f / sqrtf(fvar + eps);
    mean_o[ch] = mean;
    invstd_o[ch] = invstd;
    const float sc = gamma[ch] * invstd;
    scale_o[ch] = sc;
    shift_o[ch] = beta[ch] - mean * sc;
    mmean[ch] = mmean[ch] * momentum + mean * (1.0f - momentum);
    mvar[ch] = mvar[ch] * momentum + fvar * (1.0f - momentum);
}

__global__ void bn_infer_kernel(int c, const float* __restrict__ gamma,
                                const float* __restrict__ beta, const float* __restrict__ mmean,
                                const float* __restrict__ mvar, float eps,
                                float* __restrict__ scale_o, float* __restrict__ shift_o) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    const float sc = gamma[ch] / sqrtf(mvar[ch] + eps);
    scale_o[ch] = sc;
    shift_o[ch] = beta[ch] - mmean[ch] * sc;
}

// ---------------------------------------------------------------------------
// BatchNorm backward.  dz = (g * alpha[n,c] + beta[n,c]) masked by (mask > 0);
// xhat = (y - mean) * invstd.  Reduce: sum dz, sum dz*xhat per channel; apply:
// dy = gamma*invstd * (dz - sum_dz/M - xhat * sum_dzx/M).
// ---------------------------------------------------------------------------
struct BnBwdArgs {
    const float* g;      // upstream gradient [N][C][HW]
    const float* alpha;  // [N][C] or null (1)
    const float* addnc;  // [N][C] or null (0)
    const float* y;      // pre-BN activations
    const float* mean;
    const float* invstd;
    const float* scale;  // gamma*invstd, beta - mean*scale: the forward's BN apply
    const float* shift;
    int relu;            // 1: the BN was followed by ReLU -> gradient passes where y*scale+shift > 0
    int n, c, hw;
};

// dz for one element: the ReLU mask is recomputed from y with the forward's own fmaf, so the
// activation tensor is never materialised
__device__ __forceinline__ float bn_dz1(float g, float y, float al, float ad, float sc, float sh,
                                        int relu) {
    float dz = fmaf(g, al, ad);
    if (relu && !(fmaf(y, sc, sh) > 0.f)) dz = 0.f;
    return dz;
}

__global__ __launch_bounds__(kBlock) void bn_bwd_reduce_kernel(BnBwdArgs a, float* __restrict__ part) {
    __shared__ float red[8];
    const int ch = blockIdx.y, s = blockIdx.x;
    const float mean = a.mean[ch], invstd = a.invstd[ch], sc = a.scale[ch], sh = a.shift[ch];
    float acc[2] = {0.f, 0.f};
    for (int img = s; img < a.n; img += kBnSplit) {
        const size_t base = ((size_t)img * a.c + ch) * a.hw;
        const float al = a.alpha ? a.alpha[img * a.c + ch] : 1.f;
        const float ad = a.addnc ? a.addnc[img * a.c + ch] : 0.f;
        if ((a.hw & 3) == 0) {
            const float4* g4 = reinterpret_cast<const float4*>(a.g + base);
            const float4* y4 = reinterpret_cast<const float4*>(a.y + base);
            for (int i = threadIdx.x; i < a.hw / 4; i += kBlock) {
                const float4 g = g4[i], y = y4[i];
                const float d0 = bn_dz1(g.x, y.x, al, ad, sc, sh, a.relu);
                const float d1 = bn_dz1(g.y, y.y, al, ad, sc, sh, a.relu);
                const float d2 = bn_dz1(g.z, y.z, al, ad, sc, sh, a.relu);
                const float d3 = bn_dz1(g.w, y.w, al, ad, sc, sh, a.relu);
                acc[0] += (d0 + d1) + (d2 + d3);
                acc[1] += (d0 * ((y.x - mean) * invstd) + d1 * ((y.y - mean) * invstd)) +
                          (d2 * ((y.z - mean) * invstd) + d3 * ((y.w - mean) * invstd));
            }
        } else {
            for (int i = threadIdx.x; i < a.hw; i += kBlock) {
                const float y = a.y[base + i];
                const float dz = bn_dz1(a.g[base + i], y, al, ad, sc, sh, a.relu);
                acc[0] += dz;
                acc[1] += dz * ((y - mean) * invstd);
            }
        }
    }
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) {
        part[((size_t)ch * kBnSplit + s) * 2] = acc[0];
        part[((size_t)ch * kBnSplit + s) * 2 + 1] = acc[1];
    }
}

__global__ void bn_bwd_finalize_kernel(const float* __restrict__ part, int c,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    double s = 0.0, q = 0.0;
    for (int k = 0; k < kBnSplit; ++k) {
        s += part[((size_t)ch * kBnSplit + k) * 2];
        q += part[((size_t)ch * kBnSplit + k) * 2 + 1];
    }
    dbeta[ch] = (float)s;
    dgamma[ch] = (float)q;
}

// dbeta = sum d, dgamma = invstd * (sum d*y - mean * sum d) from the kBnSplit partial pairs
// {sum d, sum d*y} (convolution-epilogue tile sums, lf_conv2d_bnbwd_f32)
__global__ void bn_bwd_tiles_finalize_kernel(const float* __restrict__ part, int c,
                                             const float* __restrict__ mean,
                                             const float* __restrict__ invstd,
                                             float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    double s = 0.0, q = 0.0;
    for (int k = 0; k < kBnSplit; ++k) {
        s += part[((size_t)ch * kBnSplit + k) * 2];
        q += part[((size_t)ch * kBnSplit + k) * 2 + 1];
    }
    dbeta[ch] = (float)s;
    dgamma[ch] = (float)((q - (double)mean[ch] * s) * (double)invstd[ch]);
}

// dy = coef2*dz + coef3*y + coef4 with dz masked by y*coef0+coef1 > 0: the apply step as five
// per-channel coefficients, for consumers that form dy while they read g and y
__global__ void bn_bwd_coef_kernel(const float* __restrict__ mean, const float* __restrict__ invstd,
                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                   const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                   const float* __restrict__ dbeta, float inv_count, int c,
                                   float* __restrict__ coef) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    const float k = gamma[ch] * invstd[ch];
    const float mdz = dbeta[ch] * inv_count, mdzx = dgamma[ch] * inv_count;
    coef[ch] = scale[ch];
    coef[c + ch] = shift[ch];
    coef[2 * c + ch] = k;
    coef[3 * c + ch] = -k * invstd[ch] * mdzx;
    coef[4 * c + ch] = k * (mean[ch] * invstd[ch] * mdzx - mdz);
}

__global__ __launch_bounds__(kBlock) void bn_bwd_apply_kernel(BnBwdArgs a,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ dgamma,
                                                              const float* __restrict__ dbeta,
                                                              float inv_count,
                                                              float* __restrict__ dy) {
    const int plane = blockIdx.x, ch = plane % a.c;
    const float mean = a.mean[ch], invstd = a.invstd[ch], sc = a.scale[ch], sh = a.shift[ch];
    const float k = gamma[ch] * invstd;
    const float mdz = dbeta[ch] * inv_count, mdzx = dgamma[ch] * inv_count;
    const float al = a.alpha ? a.alpha[plane] : 1.f;
    const float ad = a.addnc ? a.addnc[plane] : 0.f;
    const size_t base = (size_t)plane * a.hw;
    if ((a.hw & 3) == 0) {
        const float4* g4 = reinterpret_cast<const float4*>(a.g + base);
        const float4* y4 = reinterpret_cast<const float4*>(a.y + base);
        float4* o4 = reinterpret_cast<float4*>(dy + base);
        for (int i = blockIdx.y * kBlock + threadIdx.x; i < a.hw / 4; i += gridDim.y * kBlock) {
            const float4 g = g4[i], y = y4[i];
            float4 o;
            o.x = k * (bn_dz1(g.x, y.x, al, ad, sc, sh, a.relu) - mdz - ((y.x - mean) * invstd) * mdzx);
            o.y = k * (bn_dz1(g.y, y.y, al, ad, sc, sh, a.relu) - mdz - ((y.y - mean) * invstd) * mdzx);
            o.z = k * (bn_dz1(g.z, y.z, al, ad, sc, sh, a.relu) - mdz - ((y.z - mean) * invstd) * mdzx);
            o.w = k * (bn_dz1(g.w, y.w, al, ad, sc, sh, a.relu) - mdz - ((y.w - mean) * invstd) * mdzx);
            o4[i] = o;
        }
    } else {
        for (int i = blockIdx.y * kBlock + threadIdx.x; i < a.hw; i += gridDim.y * kBlock) {
            const float y = a.y[base + i];
            const float dz = bn_dz1(a.g[base + i], y, al, ad, sc, sh, a.relu);
            dy[base + i] = k * (dz - mdz - ((y - mean) * invstd) * mdzx);
        }
    }
}

// ---------------------------------------------------------------------------
// global average pool per plane, and its broadcast backward
// ---------------------------------------------------------------------------
// out[plane] = mean_hw act(x*scale[c]+shift[c]) (scale null = plain mean); with mask_sums also
// {count of x*scale+shift > 0, sum of x over those} per plane (BatchNorm backward needs them)
__global__ __launch_bounds__(kBlock) void gap_kernel(const float* __restrict__ x,
                                                     float* __restrict__ out, int hw, int c,
                                                     const float* __restrict__ scale,
                                                     const float* __restrict__ shift, int relu,
                                                     float* __restrict__ mask_sums) {
    __shared__ float red[12];
    const size_t base = (size_t)blockIdx.x * hw;
    const bool pro = scale != nullptr;
    const float sc = pro ? scale[blockIdx.x % c] : 1.f, sh = pro ? shift[blockIdx.x % c] : 0.f;
    float acc[3] = {0.f, 0.f, 0.f};
    auto one = [&](float xv) {
        float v = xv;
        if (pro) {
            v = fmaf(xv, sc, sh);
            if (v > 0.f) {
                acc[1] += 1.f;
                acc[2] += xv;
            }
            if (relu) v = fmaxf(v, 0.f);
        }
        return v;
    };
    if ((hw & 3) == 0) {
        const float4* x4 = reinterpret_cast<const float4*>(x + base);
        for (int i = threadIdx.x; i < hw / 4; i += kBlock) {
            const float4 v = x4[i];
            const float a = one(v.x), b = one(v.y), cc = one(v.z), d = one(v.w);
            acc[0] += (a + b) + (cc + d);
        }
    } else {
        for (int i = threadIdx.x; i < hw; i += kBlock) acc[0] += one(x[base + i]);
    }
    block_sum<3>(acc, red);
    if (threadIdx.x == 0) {
        out[blockIdx.x] = acc[0] / (float)hw;
        if (mask_sums != nullptr) {
            mask_sums[2 * (size_t)blockIdx.x] = acc[1];
            mask_sums[2 * (size_t)blockIdx.x + 1] = acc[2];
        }
    }
}

__global__ __launch_bounds__(kBlock) void bcast_planes_kernel(const float* __restrict__ v,
                                                              float* __restrict__ out, int hw,
                                                              float scale) {
    const float val = v[blockIdx.x] * scale;
    const size_t base = (size_t)blockIdx.x * hw;
    for (int i = blockIdx.y * kBlock + threadIdx.x; i < hw; i += gridDim.y * kBlock) out[base + i] = val;
}

// ---------------------------------------------------------------------------
// Squeeze-Excite FC pair (1x1 convs with bias on [N,C,1,1]): one workgroup per sample
// ---------------------------------------------------------------------------
// w1 [C][Cr], b1 [Cr], w2 [Cr][C], b2 [C]  (keras kernel [1,1,in,out] flattened)
__global__ __launch_bounds__(kBlock) void se_fwd_kernel(const float* __restrict__ m,
                                                        const float* __restrict__ w1,
                                                        const float* __restrict__ b1,
                                                        const float* __restrict__ w2,
                                                        const float* __restrict__ b2,
                                                        float* __restrict__ z1,
                                                        float* __restrict__ s, int c, int cr) {
    extern __shared__ float sm[];  // [c] m, [cr] z
    float* lm = sm;
    float* lz = sm + c;
    const int n = blockIdx.x;
    for (int i = threadIdx.x; i < c; i += kBlock) lm[i] = m[(size_t)n * c + i];
    __syncthreads();
    for (int j = threadIdx.x; j < cr; j += kBlock) {
        float acc = b1[j];
        for (int i = 0; i < c; ++i) acc = fmaf(lm[i], w1[(size_t)i * cr + j], acc);
        acc = fmaxf(acc, 0.f);
        lz[j] = acc;
        z1[(size_t)n * cr + j] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < c; i += kBlock) {
        float acc = b2[i];
        for (int j = 0; j < cr; ++j) acc = fmaf(lz[j], w2[(size_t)j * c + i], acc);
        s[(size_t)n * c + i] = 1.0f / (1.0f + __expf(-acc));
    }
}

// per sample: dpre2 = ds*s*(1-s); dpre1 = (dpre2 . w2^T) * (z1>0); dm = dpre1 . w1^T
__global__ __launch_bounds__(kBlock) void se_bwd_sample_kernel(
    const float* __restrict__ ds, const float* __restrict__ s, const float* __restrict__ z1,
    const float* __restrict__ w1, const float* __restrict__ w2, float* __restrict__ dpre2,
    float* __restrict__ dpre1, float* __restrict__ dm, int c, int cr, float dm_scale) {
    extern __shared__ float sm[];  // [c] dpre2, [cr] dpre1
    float* l2 = sm;
    float* l1 = sm + c;
    const int n = blockIdx.x;
    for (int i = threadIdx.x; i < c; i += kBlock) {
        const float sv = s[(size_t)n * c + i];
        const float v = ds[(size_t)n * c + i] * sv * (1.f - sv);
        l2[i] = v;
        dpre2[(size_t)n * c + i] = v;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < cr; j += kBlock) {
        float acc = 0.f;
        for (int i = 0; i < c; ++i) acc = fmaf(l2[i], w2[(size_t)j * c + i], acc);
        if (!(z1[(size_t)n * cr + j] > 0.f)) acc = 0.f;
        l1[j] = acc;
        dpre1[(size_t)n * cr + j] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < c; i += kBlock) {
        float acc = 0.f;
        for (int j = 0; j < cr; ++j) acc = fmaf(l1[j], w1[(size_t)i * cr + j], acc);
        dm[(size_t)n * c + i] = acc * dm_scale;
    }
}

// out[i][j] = sum_n a[n][i] * b[n][j]  (i < ra, j < rb); i == ra row holds sum_n b[n][j] (bias)
// One workgroup = 64 outputs x 4 batch slices (one wave each); slices combined in fixed order.
__global__ __launch_bounds__(kBlock) void outer_sum_kernel(const float* __restrict__ a,
                                                           const float* __restrict__ b,
                                                           float* __restrict__ out,
                                                           float* __restrict__ bias_out, int n,
                                                           int ra, int rb) {
    static_assert(kBlock == 256, "outer_sum_kernel: 4 waves");
    __shared__ float part[4][64];
    const int total = (ra + 1) * rb;
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + lane;
    const int per = (n + 3) / 4;
    const int k0 = slice * per, k1 = min(n, k0 + per);
    float acc = 0.f;
    if (t < total) {
        const int i = t / rb, j = t - i * rb;
        if (i < ra) {
#pragma unroll 8
            for (int k = k0; k < k1; ++k) acc = fmaf(a[(size_t)k * ra + i], b[(size_t)k * rb + j], acc);
        } else {
#pragma unroll 8
            for (int k = k0; k < k1; ++k) acc += b[(size_t)k * rb + j];
        }
    }
    part[slice][lane] = acc;
    __syncthreads();
    if (slice == 0 && t < total) {
        const float r = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
        const int i = t / rb, j = t - i * rb;
        if (i < ra)
            out[(size_t)i * rb + j] = r;
        else
            bias_out[j] = r;
    }
}

// ---------------------------------------------------------------------------
// residual tail: r = relu(sc' + a*s[n,c]); p = drop[n,c] * maxpool2x2(r)
// ---------------------------------------------------------------------------
// a = relu(y*a_scale[c]+a_shift[c]) when a_scale is given (BN2+ReLU fused), else a = y;
// sc' = act(sc*sc_scale[c] + sc_shift[c]) (projection BN, or the stem's BN+ReLU) or sc.
struct TailArgs {
    const float* y;
    const float* a_scale;
    const float* a_shift;
    const float* s;
    const float* sc;
    const float* sc_scale;
    const float* sc_shift;
    const float* drop;
    int sc_relu;
    int c, h, w;
};

__device__ __forceinline__ float tail_r(const TailArgs& t, float yv, float scv, float as, float ab,
                                        float sv, float ks, float kb) {
    float a = yv;
    if (t.a_scale) a = fmaxf(fmaf(yv, as, ab), 0.f);
    float sh = scv;
    if (t.sc_scale) {
        sh = fmaf(scv, ks, kb);
        if (t.sc_relu) sh = fmaxf(sh, 0.f);
    }
    return fmaxf(sh + a * sv, 0.f);
}

// window code: bits 0-1 = position of the first maximum in scan order (0,0),(0,1),(1,0),(1,1);
// bit 2 = that maximum is > 0 (the gradient of the block's final ReLU)
__device__ __forceinline__ unsigned tail_code(float r00, float r01, float r10, float r11, float& best) {
    best = r00;
    unsigned bi = 0;
    if (r01 > best) { best = r01; bi = 1; }
    if (r10 > best) { best = r10; bi = 2; }
    if (r11 > best) { best = r11; bi = 3; }
    return bi | (best > 0.f ? 4u : 0u);
}

// The residual r is not stored: backward only needs where each pooled value came from.
__global__ __launch_bounds__(kBlock) void tail_fwd_kernel(TailArgs t, uint8_t* __restrict__ route,
                                                          float* __restrict__ p) {
    const int plane = blockIdx.x, ch = plane % t.c;
    const float sv = t.s ? t.s[plane] : 1.f;
    const float as = t.a_scale ? t.a_scale[ch] : 1.f, ab = t.a_scale ? t.a_shift[ch] : 0.f;
    const float ks = t.sc_scale ? t.sc_scale[ch] : 1.f, kb = t.sc_scale ? t.sc_shift[ch] : 0.f;
    const float dv = t.drop ? t.drop[plane] : 1.f;
    const int h = t.h, w = t.w, ph = h / 2, pw = w / 2;
    const size_t base = (size_t)plane * h * w, pbase = (size_t)plane * ph * pw;
    if ((w & 3) == 0) {
        // one thread = 2 pooled outputs: a 2x4 window pair, float4 rows
        const int pw2 = pw / 2;
        for (int q = blockIdx.y * kBlock + threadIdx.x; q < ph * pw2; q += gridDim.y * kBlock) {
            const int py = q / pw2, px2 = q - py * pw2;
            float4 rv[2];
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const size_t o = base + (size_t)(2 * py + dy) * w + 4 * px2;
                const float4 yv = *reinterpret_cast<const float4*>(t.y + o);
                const float4 sv4 = *reinterpret_cast<const float4*>(t.sc + o);
                rv[dy].x = tail_r(t, yv.x, sv4.x, as, ab, sv, ks, kb);
                rv[dy].y = tail_r(t, yv.y, sv4.y, as, ab, sv, ks, kb);
                rv[dy].z = tail_r(t, yv.z, sv4.z, as, ab, sv, ks, kb);
                rv[dy].w = tail_r(t, yv.w, sv4.w, as, ab, sv, ks, kb);
            }
            float m0, m1;
            const unsigned c0 = tail_code(rv[0].x, rv[0].y, rv[1].x, rv[1].y, m0);
            const unsigned c1 = tail_code(rv[0].z, rv[0].w, rv[1].z, rv[1].w, m1);
            const size_t po = pbase + (size_t)py * pw + 2 * px2;  // even: pw is even here
            *reinterpret_cast<uint16_t*>(route + po) = (uint16_t)(c0 | (c1 << 8));
            *reinterpret_cast<float2*>(p + po) = make_float2(m0 * dv, m1 * dv);
        }
    } else {
        for (int q = blockIdx.y * kBlock + threadIdx.x; q < ph * pw; q += gridDim.y * kBlock) {
            const int py = q / pw, px = q - py * pw;
            const size_t o0 = base + (size_t)(2 * py) * w + 2 * px, o1 = o0 + w;
            const float r00 = tail_r(t, t.y[o0], t.sc[o0], as, ab, sv, ks, kb);
            const float r01 = tail_r(t, t.y[o0 + 1], t.sc[o0 + 1], as, ab, sv, ks, kb);
            const float r10 = tail_r(t, t.y[o1], t.sc[o1], as, ab, sv, ks, kb);
            const float r11 = tail_r(t, t.y[o1 + 1], t.sc[o1 + 1], as, ab, sv, ks, kb);
            float mx;
            route[pbase + q] = (uint8_t)tail_code(r00, r01, r10, r11, mx);
            p[pbase + q] = mx * dv;
        }
    }
}

// dr = dp*drop routed to the recorded position of each 2x2 window when its maximum was > 0.
// Per plane: ds = sum dr*a with a = relu(y*a_scale+a_shift) (or y), and (a_scale given)
// plane_sums = {sum dr*[a>0], sum dr*[a>0]*y}: BatchNorm-2's backward sums without a second pass.
__global__ __launch_bounds__(kBlock) void tail_bwd_kernel(const float* __restrict__ dp,
                                                          const uint8_t* __restrict__ route,
                                                          const float* __restrict__ y,
                                                          const float* __restrict__ a_scale,
                                                          const float* __restrict__ a_shift,
                                                          const float* __restrict__ drop,
                                                          float* __restrict__ dr,
                                                          float* __restrict__ ds,
                                                          float* __restrict__ plane_sums,
                                                          const float* __restrict__ sc_y,
                                                          float* __restrict__ sc_sums, int c,
                                                          int h, int w) {
    __shared__ float red[20];
    const int plane = blockIdx.x, ch = plane % c;
    const float dv = drop ? drop[plane] : 1.f;
    const float as = a_scale ? a_scale[ch] : 1.f, ab = a_scale ? a_shift[ch] : 0.f;
    const int ph = h / 2, pw = w / 2;
    const size_t base = (size_t)plane * h * w, pbase = (size_t)plane * ph * pw;
    const bool sums = ds != nullptr || plane_sums != nullptr;
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    auto tally = [&](float gg, size_t pos) {
        if (sc_y != nullptr && gg != 0.f) {  // the shortcut branch's (projection) BN: no mask
            acc[3] += gg;
            acc[4] += gg * sc_y[pos];
        }
        if (sums && gg != 0.f) {
            const float yv = y[pos];
            float av = yv;
            if (a_scale) {
                av = fmaf(yv, as, ab);
                if (av > 0.f) {
                    acc[1] += gg;
                    acc[2] += gg * yv;
                } else {
                    av = 0.f;
                }
            }
            acc[0] += gg * av;
        }
    };
    if ((w & 3) == 0) {
        const int pw2 = pw / 2;
        for (int q = threadIdx.x; q < ph * pw2; q += kBlock) {
            const int py = q / pw2, px2 = q - py * pw2;
            const size_t po = pbase + (size_t)py * pw + 2 * px2;
            const unsigned codes = *reinterpret_cast<const uint16_t*>(route + po);
            const float2 g2 = *reinterpret_cast<const float2*>(dp + po);
            const unsigned c0 = codes & 0xff, c1 = codes >> 8;
            const float g0 = (c0 & 4u) ? g2.x * dv : 0.f, g1 = (c1 & 4u) ? g2.y * dv : 0.f;
            const unsigned b0 = c0 & 3u, b1 = c1 & 3u;
            const size_t o0 = base + (size_t)(2 * py) * w + 4 * px2, o1 = o0 + w;
            *reinterpret_cast<float4*>(dr + o0) = make_float4(b0 == 0 ? g0 : 0.f, b0 == 1 ? g0 : 0.f,
                                                              b1 == 0 ? g1 : 0.f, b1 == 1 ? g1 : 0.f);
            *reinterpret_cast<float4*>(dr + o1) = make_float4(b0 == 2 ? g0 : 0.f, b0 == 3 ? g0 : 0.f,
                                                              b1 == 2 ? g1 : 0.f, b1 == 3 ? g1 : 0.f);
            tally(g0, (b0 < 2 ? o0 : o1) + (b0 & 1u));
            tally(g1, (b1 < 2 ? o0 : o1) + 2 + (b1 & 1u));
        }
    } else {
        for (int t = threadIdx.x; t < ph * pw; t += kBlock) {
            const int py = t / pw, px = t - py * pw;
            const size_t o0 = base + (size_t)(2 * py) * w + 2 * px, o1 = o0 + w;
            const unsigned code = route[pbase + t];
            const float gg = (code & 4u) ? dp[pbase + t] * dv : 0.f;
            const unsigned bi = code & 3u;
            dr[o0] = bi == 0 ? gg : 0.f;
            dr[o0 + 1] = bi == 1 ? gg : 0.f;
            dr[o1] = bi == 2 ? gg : 0.f;
            dr[o1 + 1] = bi == 3 ? gg : 0.f;
            tally(gg, (bi < 2 ? o0 : o1) + (bi & 1u));
        }
    }
    if ((h & 1) || (w & 1)) {
        for (int t = threadIdx.x; t < h * w; t += kBlock) {
            const int yy = t / w, x = t - yy * w;
            if (yy >= 2 * ph || x >= 2 * pw) dr[base + t] = 0.f;
        }
    }
    if (sums || sc_y != nullptr) {
        block_sum<5>(acc, red);
        if (threadIdx.x == 0) {
            if (ds != nullptr) ds[plane] = acc[0];
            if (plane_sums != nullptr) {
                plane_sums[2 * (size_t)plane] = acc[1];
                plane_sums[2 * (size_t)plane + 1] = acc[2];
            }
            if (sc_sums != nullptr) {
                sc_sums[2 * (size_t)plane] = acc[3];
                sc_sums[2 * (size_t)plane + 1] = acc[4];
            }
        }
    }
}

// BatchNorm backward sums from per-plane sums (no pass over the activation): with
// dz = (g*alpha + add) * [y*scale+shift > 0],
//   sum dz       = sum_n alpha*G0 + add*M0          G = tail_bwd plane sums {sum g*mask, sum g*mask*y}
//   sum dz*xhat  = invstd * sum_n alpha*(G1 - mean*G0) + add*(M1 - mean*M0)   M = gap mask sums
// one workgroup per channel, fixed-order double accumulation.
__global__ __launch_bounds__(kBlock) void bn_bwd_planes_kernel(const float* __restrict__ plane_g,
                                                               const float* __restrict__ plane_m,
                                                               const float* __restrict__ alpha,
                                                               const float* __restrict__ addnc,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, int n,
                                                               int c, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta) {
    __shared__ double red[2][kBlock / 64];
    const int ch = blockIdx.x;
    const double mu = mean[ch];
    double s0 = 0.0, s1 = 0.0;
    for (int img = threadIdx.x; img < n; img += kBlock) {
        const size_t pl = (size_t)img * c + ch;
        const double al = alpha ? alpha[pl] : 1.0, ad = addnc ? addnc[pl] : 0.0;
        const double g0 = plane_g[2 * pl], g1 = plane_g[2 * pl + 1];
        const double m0 = plane_m ? plane_m[2 * pl] : 0.0, m1 = plane_m ? plane_m[2 * pl + 1] : 0.0;
        s0 += al * g0 + ad * m0;
        s1 += al * (g1 - mu * g0) + ad * (m1 - mu * m0);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_down(s0, off, 64);
        s1 += __shfl_down(s1, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s0;
        red[1][threadIdx.x >> 6] = s1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < kBlock / 64; ++k) {
            a += red[0][k];
            b += red[1][k];
        }
        dbeta[ch] = (float)a;
        dgamma[ch] = (float)(b * (double)invstd[ch]);
    }
}

// ---------------------------------------------------------------------------
// head: logits = (g*drop) W + b; softmax; label-smoothed cross-entropy (keras semantics)
// ---------------------------------------------------------------------------
// one workgroup (64 threads) per sample; `feat` is the (dropped-out) GAP feature vector
__global__ void head_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ w,
                                const float* __restrict__ b, const float* __restrict__ ytrue,
                                float* __restrict__ probs, float* __restrict__ loss, int f, int c) {
    extern __shared__ float sm[];  // [c] logits
    const int n = blockIdx.x;
    for (int j = threadIdx.x; j < c; j += blockDim.x) {
        float acc = b[j];
        for (int i = 0; i < f; ++i) acc = fmaf(feat[(size_t)n * f + i], w[(size_t)i * c + j], acc);
        sm[j] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float mx = sm[0];
        for (int j = 1; j < c; ++j) mx = fmaxf(mx, sm[j]);
        float den = 0.f;
        for (int j = 0; j < c; ++j) den += expf(sm[j] - mx);
        float l = 0.f;
        for (int j = 0; j < c; ++j) {
            const float pr = expf(sm[j] - mx) / den;
            probs[(size_t)n * c + j] = pr;
            if (ytrue != nullptr) {
                // keras categorical_crossentropy: clip to [1e-7, 1-1e-7] then -sum y log p
                const float pc = fminf(fmaxf(pr, 1e-7f), 1.f - 1e-7f);
                l -= ytrue[(size_t)n * c + j] * logf(pc);
            }
        }
        if (loss != nullptr) loss[n] = l;
    }
}

// dlogits = (probs - ytrue) * inv_n; dfeat = dlogits W^T; (dW, db) by outer_sum_kernel
__global__ void head_bwd_kernel(const float* __restrict__ probs, const float* __restrict__ ytrue,
                                const float* __restrict__ w, float* __restrict__ dlogits,
                                float* __restrict__ dfeat, int f, int c, float inv_n) {
    extern __shared__ float sm[];  // [c]
    const int n = blockIdx.x;
    for (int j = threadIdx.x; j < c; j += blockDim.x) {
        const float d = (probs[(size_t)n * c + j] - ytrue[(size_t)n * c + j]) * inv_n;
        sm[j] = d;
        dlogits[(size_t)n * c + j] = d;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < f; i += blockDim.x) {
        float acc = 0.f;
        for (int j = 0; j < c; ++j) acc = fmaf(sm[j], w[(size_t)i * c + j], acc);
        dfeat[(size_t)n * f + i] = acc;
    }
}

__global__ __launch_bounds__(kBlock) void mul_kernel(const float* __restrict__ a,
                                                     const float* __restrict__ b,
                                                     float* __restrict__ out, size_t count) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count;
         i += (size_t)gridDim.x * kBlock)
        out[i] = a[i] * b[i];
}

// ---------------------------------------------------------------------------
// AdamW (keras 3): per-tensor clipnorm, L2-regulariser gradient, decoupled weight decay,
// bias-corrected Adam, EMA of the updated weights.  Flat buffers + a segment table.
// ---------------------------------------------------------------------------
// norms: one workgroup per tensor: sum over (g + 2*l2*w)^2
// squared gradient norm of tensor t in kNormSplit contiguous slices: grid = (kNormSplit, ntensors)
constexpr int kNormSplit = 32;
__global__ __launch_bounds__(kBlock) void adam_norm_kernel(const float* __restrict__ p,
                                                           const float* __restrict__ g,
                                                           const long long* __restrict__ offs,
                                                           const float* __restrict__ l2,
                                                           double* __restrict__ partial) {
    __shared__ double red[kBlock / 64];
    const int t = blockIdx.y;
    const long long b = offs[t], e = offs[t + 1];
    const long long per = (e - b + kNormSplit - 1) / kNormSplit;
    const long long s0 = b + per * blockIdx.x, s1 = s0 + per < e ? s0 + per : e;
    const float l2c = 2.f * l2[t];
    double acc = 0.0;
#pragma unroll 4
    for (long long i = s0 + threadIdx.x; i < s1; i += kBlock) {
        const float gv = fmaf(l2c, p[i], g[i]);
        acc += (double)gv * gv;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int k = 0; k < kBlock / 64; ++k) s += red[k];
        partial[(size_t)t * kNormSplit + blockIdx.x] = s;
    }
}

struct AdamArgs {
    float* p;
    const float* g;
    float* m;
    float* v;
    float* ema;  // may be null
    const long long* offs;
    const float* l2;
    const double* norm_partial;
    float* norms;
    float lr, beta1, beta2, eps, wd, clipnorm, alpha, ema_decay;
    int ema_copy;
};

__global__ __launch_bounds__(kBlock) void adam_step_kernel(AdamArgs a) {
    const int t = blockIdx.y;
    const long long b = a.offs[t], e = a.offs[t + 1];
    const float l2c = 2.f * a.l2[t];
    // keras clip_by_norm: g * clip / max(norm, clip)
    double sq = 0.0;
    for (int k = 0; k < kNormSplit; ++k) sq += a.norm_partial[(size_t)t * kNormSplit + k];
    const float norm = (float)sqrt(sq);
    if (blockIdx.x == 0 && threadIdx.x == 0) a.norms[t] = norm;
    const float cf = a.clipnorm > 0.f ? a.clipnorm / fmaxf(norm, a.clipnorm) : 1.f;
    for (long long i = b + (long long)blockIdx.x * kBlock + threadIdx.x; i < e;
         i += (long long)gridDim.x * kBlock) {
        float w = a.p[i];
        const float gv = fmaf(l2c, w, a.g[i]) * cf;
        w -= w * a.wd * a.lr;  // decoupled decay first (keras _apply_weight_decay)
        float m = a.m[i], v = a.v[i];
        m += (gv - m) * (1.f - a.beta1);
        v += (gv * gv - v) * (1.f - a.beta2);
        w -= m * a.alpha / (sqrtf(v) + a.eps);
        a.p[i] = w;
        a.m[i] = m;
        a.v[i] = v;
        if (a.ema != nullptr) a.ema[i] = a.ema_copy ? w : a.ema_decay * a.ema[i] + (1.f - a.ema_decay) * w;
    }
}

__global__ __launch_bounds__(kBlock) void ema_kernel(float* __restrict__ ema,
                                                     const float* __restrict__ w, size_t count,
                                                     float decay, int copy) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count;
         i += (size_t)gridDim.x * kBlock)
        ema[i] = copy ? w[i] : decay * ema[i] + (1.f - decay) * w[i];
}

inline unsigned plane_grid(int hw_items) { return lf::stream_grid((size_t)hw_items, kBlock, 64); }

}  // namespace

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

int lf_input_stage_f32(const uint8_t* in, float* out, int n, int h, int w, const float* aug4,
                       const float* mean3, const float* denom3, float* means_ws,
                       lf_stream_t stream) {
    LF_REQUIRE(in && out && aug4 && means_ws, "lf_input_stage: null buffer");
    LF_REQUIRE(n > 0 && h > 1 && w > 1, "lf_input_stage: bad dims n=%d h=%d w=%d", n, h, w);
    LF_REQUIRE((mean3 == nullptr) == (denom3 == nullptr), "lf_input_stage: mean/denom must both be set");
    LF_REQUIRE(n <= 65535, "lf_input_stage: batch too large");
    float m[3] = {0, 0, 0}, d[3] = {1, 1, 1};
    if (mean3)
        for (int c = 0; c < 3; ++c) {
            m[c] = mean3[c];
            d[c] = denom3[c];
        }
    hipStream_t s = lf::as_stream(stream);
    aug_means_kernel<<<dim3(kMeanSplit, n), kBlock, 0, s>>>(in, aug4, means_ws, h, w);
    input_aug_kernel<<<dim3(plane_grid(h * w), n), kBlock, 0, s>>>(in, out, aug4, means_ws, h, w,
                                                                   m[0], m[1], m[2], d[0], d[1], d[2]);
    return lf::check_launch("lf_input_stage");
}

int lf_scale_shift_act_f32(const float* x, float* out, int n, int c, int hw, const float* scale,
                           const float* shift, int relu, lf_stream_t stream) {
    LF_REQUIRE(x && out && scale && shift, "lf_scale_shift_act: null buffer");
    LF_REQUIRE(n > 0 && c > 0 && hw > 0 && (long long)n * c < (1LL << 31),
               "lf_scale_shift_act: bad dims n=%d c=%d hw=%d", n, c, hw);
    hipStream_t s = lf::as_stream(stream);
    const int planes = n * c;
    if ((hw & 3) == 0)
        scale_shift_act_kernel<4><<<dim3(planes, plane_grid(hw / 4)), kBlock, 0, s>>>(x, out, c, hw / 4,
                                                                                 scale, shift, relu);
    else
        scale_shift_act_kernel<1><<<dim3(planes, plane_grid(hw)), kBlock, 0, s>>>(x, out, c, hw, scale,
                                                                             shift, relu);
    return lf::check_launch("lf_scale_shift_act");
}

size_t lf_bn_workspace(int c) { return c > 0 ? (size_t)c * kBnSplit * 2 * sizeof(float) : 0; }

int lf_bn_train_stats_f32(const float* y, int n, int c, int hw, const float* gamma,
                          const float* beta, float* moving_mean, float* moving_var, float momentum,
                          float eps, float* mean, float* invstd, float* scale, float* shift,
                          void* workspace, size_t ws_bytes, lf_stream_t stream) {
    LF_REQUIRE(y && gamma && beta && moving_mean && moving_var && mean && invstd && scale && shift &&
                   workspace,
               "lf_bn_train_stats: null buffer");
    LF_REQUIRE(n > 0 && c > 0 && hw > 0 && c <= 65535, "lf_bn_train_stats: bad dims n=%d c=%d hw=%d", n, c, hw);
    if (ws_bytes < lf_bn_workspace(c)) {
        lf::set_error("lf_bn_train_stats: workspace %zu < %zu", ws_bytes, lf_bn_workspace(c));
        return LF_ERR_WORKSPACE;
    }
    hipStream_t s = lf::as_stream(stream);
    float* part = static_cast<float*>(workspace);
    bn_stats_kernel<<<dim3(kBnSplit, c), kBlock, 0, s>>>(y, n, c, hw, part);
    bn_finalize_kernel<<<(c + 63) / 64, 64, 0, s>>>(y, hw, part, c, (double)n * hw, gamma, beta,
                                                   moving_mean, moving_var, momentum, eps, mean,
                                                   invstd, scale, shift);
    return lf::check_launch("lf_bn_train_stats");
}

int lf_bn_train_stats_tiles_f32(const float* tile_part, long long tiles, int n, int c, int hw,
                                const float* gamma, const float* beta, float* moving_mean,
                                float* moving_var, float momentum, float eps, float* mean,
                                float* invstd, float* scale, float* shift, void* workspace,
                                size_t ws_bytes, lf_stream_t stream) {
    LF_REQUIRE(tile_part && gamma && beta && moving_mean && moving_var && mean && invstd && scale &&
                   shift && workspace,
               "lf_bn_train_stats_tiles: null buffer");
    LF_REQUIRE(n > 0 && c > 0 && hw > 0 && c <= 65535 && tiles > 0,
               "lf_bn_train_stats_tiles: bad dims n=%d c=%d hw=%d tiles=%lld", n, c, hw, tiles);
    if (ws_bytes < lf_bn_workspace(c)) {
        lf::set_error("lf_bn_train_stats_tiles: workspace %zu < %zu", ws_bytes, lf_bn_workspace(c));
        return LF_ERR_WORKSPACE;
    }
    hipStream_t s = lf::as_stream(stream);
    float* part = static_cast<float*>(workspace);
    bn_tile_reduce_kernel<<<dim3(kBnSplit, c), kBlock, 0, s>>>(tile_part, tiles, part);
    // the tile sums were taken about moving_mean as it still is here (updated by this kernel)
    bn_finalize_kernel<<<(c + 63) / 64, 64, 0, s>>>(moving_mean, 1, part, c, (double)n * hw, gamma,
                                                   beta, moving_mean, moving_var, momentum, eps,
                                                   mean, invstd, scale, shift);
    return lf::check_launch("lf_bn_train_stats_tiles");
}

int lf_bn_infer_scale_shift_f32(int c, const float* gamma, const float* beta,
                                const float* moving_mean, const float* moving_var, float eps,
                                float* scale, float* shift, lf_stream_t stream) {
    LF_REQUIRE(gamma && beta && moving_mean && moving_var && scale && shift, "lf_bn_infer: null buffer");
    LF_REQUIRE(c > 0, "lf_bn_infer: bad c=%d", c);
    bn_infer_kernel<<<(c + 63) / 64, 64, 0, lf::as_stream(stream)>>>(c, gamma, beta, moving_mean,
                                                                   moving_var, eps, scale, shift);
    return lf::check_launch("lf_bn_infer");
}

int lf_bn_bwd_f32(const float* g, const float* alpha_nc, const float* add_nc, const float* y,
                  const float* mean, const float* invstd, const float* scale, const float* shift,
                  int relu, const float* gamma, float* dy, float* dgamma, float* dbeta,
                  const float* plane_g, const float* plane_m, int have_sums, int n, int c, int hw,
                  void* workspace, size_t ws_bytes, lf_stream_t stream) {
    LF_REQUIRE(g && y && mean && invstd && scale && shift && gamma && dy && dgamma && dbeta && workspace,
               "lf_bn_bwd: null buffer");
    LF_REQUIRE(!have_sums || plane_g == nullptr, "lf_bn_bwd: have_sums excludes plane sums");
    LF_REQUIRE(plane_g != nullptr || plane_m == nullptr, "lf_bn_bwd: plane_m needs plane_g");
    LF_REQUIRE(plane_g == nullptr || relu != 0 || (alpha_nc == nullptr && add_nc == nullptr),
               "lf_bn_bwd: plane sums without a mask take no alpha / add");
    LF_REQUIRE(plane_g == nullptr || add_nc == nullptr || plane_m != nullptr,
               "lf_bn_bwd: add_nc with plane sums needs plane_m");
    LF_REQUIRE(n > 0 && c > 0 && hw > 0 && c <= 65535 && (long long)n * c < (1LL << 31),
               "lf_bn_bwd: bad dims n=%d c=%d hw=%d", n, c, hw);
    if (ws_bytes < lf_bn_workspace(c)) {
        lf::set_error("lf_bn_bwd: workspace %zu < %zu", ws_bytes, lf_bn_workspace(c));
        return LF_ERR_WORKSPACE;
    }
    BnBwdArgs a{g, alpha_nc, add_nc, y, mean, invstd, scale, shift, relu, n, c, hw};
    hipStream_t s = lf::as_stream(stream);
    float* part = static_cast<float*>(workspace);
    if (have_sums) {
        // dgamma / dbeta already hold the two channel sums (lf_bn_bwd_sums*_f32)
    } else if (plane_g != nullptr) {
        bn_bwd_planes_kernel<<<c, kBlock, 0, s>>>(plane_g, plane_m, alpha_nc, add_nc, mean, invstd, n, c,
                                                  dgamma, dbeta);
    } else {
        bn_bwd_reduce_kernel<<<dim3(kBnSplit, c), kBlock, 0, s>>>(a, part);
        bn_bwd_finalize_kernel<<<(c + 63) / 64, 64, 0, s>>>(part, c, dgamma, dbeta);
    }
    bn_bwd_apply_kernel<<<dim3(n * c, plane_grid((hw & 3) ? hw : hw / 4)), kBlock, 0, s>>>(
        a, gamma, dgamma, dbeta, 1.0f / ((float)n * (float)hw), dy);
    return lf::check_launch("lf_bn_bwd");
}

int lf_bn_bwd_sums_f32(const float* g, const float* alpha_nc, const float* add_nc, const float* y,
                       const float* mean, const float* invstd, const float* scale,
                       const float* shift, int relu, const float* gamma, float* dgamma, float* dbeta,
                       float* coef, const float* plane_g, const float* plane_m, int n, int c, int hw,
                       void* workspace, size_t ws_bytes, lf_stream_t stream) {
    LF_REQUIRE(mean && invstd && scale && shift && gamma && dgamma && dbeta && coef && workspace,
               "lf_bn_bwd_sums: null buffer");
    LF_REQUIRE(plane_g != nullptr || (g != nullptr && y != nullptr), "lf_bn_bwd_sums: g / y missing");
    LF_REQUIRE(n > 0 && c > 0 && hw > 0 && c <= 65535 && (long long)n * c < (1LL << 31),
               "lf_bn_bwd_sums: bad dims n=%d c=%d hw=%d", n, c, hw);
    LF_REQUIRE(plane_g != nullptr || plane_m == nullptr, "lf_bn_bwd_sums: plane_m needs plane_g");
    LF_REQUIRE(plane_g == nullptr || relu != 0 || (alpha_nc == nullptr && add_nc == nullptr),
               "lf_bn_bwd_sums: plane sums without a mask take no alpha / add");
    LF_REQUIRE(plane_g == nullptr || add_nc == nullptr || plane_m != nullptr,
               "lf_bn_bwd_sums: add_nc with plane sums needs plane_m");
    if (ws_bytes < lf_bn_workspace(c)) {
        lf::set_error("lf_bn_bwd_sums: workspace %zu < %zu", ws_bytes, lf_bn_workspace(c));
        return LF_ERR_WORKSPACE;
    }
    hipStream_t s = lf::as_stream(stream);
    if (plane_g != nullptr) {
        bn_bwd_planes_kernel<<<c, kBlock, 0, s>>>(plane_g, plane_m, alpha_nc, add_nc, mean, invstd, n, c,
                                                  dgamma, dbeta);
    } else {
        BnBwdArgs a{g, alpha_nc, add_nc, y, mean, invstd, scale, shift, relu, n, c, hw};
        float* part = static_cast<float*>(workspace);
        bn_bwd_reduce_kernel<<<dim3(kBnSplit, c), kBlock, 0, s>>>(a, part);
        bn_bwd_finalize_kernel<<<(c + 63) / 64, 64, 0, s>>>(part, c, dgamma, dbeta);
    }
    bn_bwd_coef_kernel<<<(c + 63) / 64, 64, 0, s>>>(mean, invstd, scale, shift, gamma, dgamma, dbeta,
                                                   1.0f / ((float)n * (float)hw), c, coef);
    return lf::check_launch("lf_bn_bwd_sums");
}

int lf_bn_bwd_sums_tiles_f32(const float* tile_part, long long tiles, const float* mean,
                             const float* invstd, const float* scale, const float* shift,
                             const float* gamma, float* dgamma, float* dbeta, float* coef, int n,
                             int c, int hw, void* workspace, size_t ws_bytes, lf_stream_t stream) {
    LF_REQUIRE(tile_part && mean && invstd && scale && shift && gamma && dgamma && dbeta && coef &&
                   workspace,
               "lf_bn_bwd_sums_tiles: null buffer");
    LF_REQUIRE(n > 0 && c > 0 && hw > 0 && c <= 65535 && tiles > 0,
               "lf_bn_bwd_sums_tiles: bad dims n=%d c=%d hw=%d tiles=%lld", n, c, hw, tiles);
    if (ws_bytes < lf_bn_workspace(c)) {
        lf::set_error("lf_bn_bwd_sums_tiles: workspace %zu < %zu", ws_bytes, lf_bn_workspace(c));
        return LF_ERR_WORKSPACE;
    }
    hipStream_t s = lf::as_stream(stream);
    float* part = static_cast<float*>(workspace);
    bn_tile_reduce_kernel<<<dim3(kBnSplit, c), kBlock, 0, s>>>(tile_part, tiles, part);
    bn_bwd_tiles_finalize_kernel<<<(c + 63) / 64, 64, 0, s>>>(part, c, mean, invstd, dgamma, dbeta);
    bn_bwd_coef_kernel<<<(c + 63) / 64, 64, 0, s>>>(mean, invstd, scale, shift, gamma, dgamma, dbeta,
                                                   1.0f / ((float)n * (float)hw), c, coef);
    return lf::check_launch("lf_bn_bwd_sums_tiles");
}

int lf_gap_f32(const float* x, float* out, int planes, int hw, int c, const float* scale,
               const float* shift, int relu, float* mask_sums, lf_stream_t stream) {
    LF_REQUIRE(x && out, "lf_gap: null buffer");
    LF_REQUIRE(mask_sums == nullptr || scale != nullptr, "lf_gap: mask_sums needs scale/shift");
    LF_REQUIRE(planes > 0 && hw > 0 && c > 0, "lf_gap: bad dims planes=%d hw=%d c=%d", planes, hw, c);
    LF_REQUIRE((scale == nullptr) == (shift == nullptr), "lf_gap: scale/shift must both be set");
    gap_kernel<<<planes, kBlock, 0, lf::as_stream(stream)>>>(x, out, hw, c, scale, shift, relu,
                                                             mask_sums);
    return lf::check_launch("lf_gap");
}

int lf_bcast_planes_f32(const float* v, float* out, int planes, int hw, float scale,
                        lf_stream_t stream) {
    LF_REQUIRE(v && out, "lf_bcast_planes: null buffer");
    LF_REQUIRE(planes > 0 && hw > 0, "lf_bcast_planes: bad dims planes=%d hw=%d", planes, hw);
    bcast_planes_kernel<<<dim3(planes, plane_grid(hw)), kBlock, 0, lf::as_stream(stream)>>>(v, out, hw,
                                                                                           scale);
    return lf::check_launch("lf_bcast_planes");
}

int lf_se_fwd_f32(const float* m, const float* w1, const float* b1, const float* w2,
                  const float* b2, float* z1, float* s, int n, int c, int cr, lf_stream_t stream) {
    LF_REQUIRE(m && w1 && b1 && w2 && b2 && z1 && s, "lf_se_fwd: null buffer");
    LF_REQUIRE(n > 0 && c > 0 && cr > 0 && c + cr <= 8192, "lf_se_fwd: bad dims n=%d c=%d cr=%d", n, c, cr);
    se_fwd_kernel<<<n, kBlock, (size_t)(c + cr) * sizeof(float), lf::as_stream(stream)>>>(
        m, w1, b1, w2, b2, z1, s, c, cr);
    return lf::check_launch("lf_se_fwd");
}

size_t lf_se_bwd_workspace(int n, int c, int cr) {
    return (n > 0 && c > 0 && cr > 0) ? (size_t)n * (c + cr) * sizeof(float) : 0;
}

int lf_se_bwd_f32(const float* ds, const float* m, const float* z1, const float* s,
                  const float* w1, const float* w2, float* dm, float* dw1, float* db1, float* dw2,
                  float* db2, int n, int c, int cr, float dm_scale, void* workspace,
                  size_t ws_bytes, lf_stream_t stream) {
    LF_REQUIRE(ds && m && z1 && s && w1 && w2 && dm && dw1 && db1 && dw2 && db2 && workspace,
               "lf_se_bwd: null buffer");
    LF_REQUIRE(n > 0 && c > 0 && cr > 0 && c + cr <= 8192, "lf_se_bwd: bad dims n=%d c=%d cr=%d", n, c, cr);
    if (ws_bytes < lf_se_bwd_workspace(n, c, cr)) {
        lf::set_error("lf_se_bwd: workspace too small");
        return LF_ERR_WORKSPACE;
    }
    hipStream_t st = lf::as_stream(stream);
    float* dpre2 = static_cast<float*>(workspace);
    float* dpre1 = dpre2 + (size_t)n * c;
    se_bwd_sample_kernel<<<n, kBlock, (size_t)(c + cr) * sizeof(float), st>>>(ds, s, z1, w1, w2, dpre2,
                                                                              dpre1, dm, c, cr, dm_scale);
    // dw1[c][cr] = sum_n m[n][c] dpre1[n][cr]; dw2[cr][c] = sum_n z1[n][cr] dpre2[n][c]
    outer_sum_kernel<<<(unsigned)(((size_t)(c + 1) * cr + 63) / 64), kBlock, 0, st>>>(m, dpre1, dw1, db1,
                                                                                    n, c, cr);
    outer_sum_kernel<<<(unsigned)(((size_t)(cr + 1) * c + 63) / 64), kBlock, 0, st>>>(z1, dpre2, dw2,
                                                                                    db2, n, cr, c);
    return lf::check_launch("lf_se_bwd");
}

int lf_block_tail_fwd_f32(const float* y, const float* a_scale, const float* a_shift,
                          const float* s, const float* sc, const float* sc_scale,
                          const float* sc_shift, int sc_relu, const float* drop, uint8_t* route,
                          float* p, int n, int c, int h, int w, lf_stream_t stream) {
    LF_REQUIRE(y && sc && route && p, "lf_block_tail_fwd: null buffer");
    LF_REQUIRE(n > 0 && c > 0 && h > 1 && w > 1 && (long long)n * c < (1LL << 31),
               "lf_block_tail_fwd: bad dims n=%d c=%d h=%d w=%d", n, c, h, w);
    LF_REQUIRE((w & 1) == 0, "lf_block_tail_fwd: width must be even");
    LF_REQUIRE((sc_scale == nullptr) == (sc_shift == nullptr), "lf_block_tail_fwd: sc_scale/sc_shift");
    LF_REQUIRE((a_scale == nullptr) == (a_shift == nullptr), "lf_block_tail_fwd: a_scale/a_shift");
    TailArgs t{y, a_scale, a_shift, s, sc, sc_scale, sc_shift, drop, sc_relu, c, h, w};
    const int items = (w & 3) == 0 ? (h / 2) * (w / 4) : (h / 2) * (w / 2);
    tail_fwd_kernel<<<dim3(n * c, plane_grid(items)), kBlock, 0, lf::as_stream(stream)>>>(t, route, p);
    return lf::check_launch("lf_block_tail_fwd");
}

int lf_block_tail_bwd_f32(const float* dp, const uint8_t* route, const float* y,
                          const float* a_scale, const float* a_shift, const float* drop, float* dr,
                          float* ds, float* plane_sums, const float* sc_y, float* sc_sums, int n,
                          int c, int h, int w, lf_stream_t stream) {
    LF_REQUIRE((sc_y == nullptr) == (sc_sums == nullptr), "lf_block_tail_bwd: sc_y and sc_sums go together");
    LF_REQUIRE(dp && route && dr, "lf_block_tail_bwd: null buffer");
    LF_REQUIRE(plane_sums == nullptr || (y != nullptr && a_scale != nullptr),
               "lf_block_tail_bwd: plane_sums needs y and a_scale/a_shift");
    LF_REQUIRE(n > 0 && c > 0 && h > 1 && w > 1, "lf_block_tail_bwd: bad dims n=%d c=%d h=%d w=%d", n, c, h, w);
    LF_REQUIRE((w & 1) == 0, "lf_block_tail_bwd: width must be even (float2 rows)");
    LF_REQUIRE((y == nullptr) == (ds == nullptr && plane_sums == nullptr),
               "lf_block_tail_bwd: y goes with ds / plane_sums");
    LF_REQUIRE((a_scale == nullptr) == (a_shift == nullptr), "lf_block_tail_bwd: a_scale/a_shift");
    tail_bwd_kernel<<<n * c, kBlock, 0, lf::as_stream(stream)>>>(dp, route, y, a_scale, a_shift, drop, dr,
                                                                ds, plane_sums, sc_y, sc_sums, c, h, w);
    return lf::check_launch("lf_block_tail_bwd");
}

int lf_head_fwd_f32(const float* feat, const float* w, const float* b, const float* ytrue,
                    float* probs, float* loss, int n, int f, int c, lf_stream_t stream) {
    LF_REQUIRE(feat && w && b && probs, "lf_head_fwd: null buffer");
    LF_REQUIRE(n > 0 && f > 0 && c > 0 && c <= 4096, "lf_head_fwd: bad dims n=%d f=%d c=%d", n, f, c);
    LF_REQUIRE((ytrue == nullptr) == (loss == nullptr), "lf_head_fwd: ytrue and loss go together");
    head_fwd_kernel<<<n, 64, (size_t)c * sizeof(float), lf::as_stream(stream)>>>(feat, w, b, ytrue, probs,
                                                                               loss, f, c);
    return lf::check_launch("lf_head_fwd");
}

int lf_head_bwd_f32(const float* feat, const float* w, const float* probs, const float* ytrue,
                    float* dlogits, float* dfeat, float* dw, float* db, int n, int f, int c,
                    float inv_n, lf_stream_t stream) {
    LF_REQUIRE(feat && w && probs && ytrue && dlogits && dfeat && dw && db, "lf_head_bwd: null buffer");
    LF_REQUIRE(n > 0 && f > 0 && c > 0 && c <= 4096, "lf_head_bwd: bad dims n=%d f=%d c=%d", n, f, c);
    hipStream_t st = lf::as_stream(stream);
    head_bwd_kernel<<<n, 64, (size_t)c * sizeof(float), st>>>(probs, ytrue, w, dlogits, dfeat, f, c, inv_n);
    // dW[f][c] = sum_n feat[n][f] dlogits[n][c]; db[c] = sum_n dlogits[n][c]
    outer_sum_kernel<<<(unsigned)(((size_t)(f + 1) * c + 63) / 64), kBlock, 0, st>>>(feat, dlogits, dw, db,
                                                                                   n, f, c);
    return lf::check_launch("lf_head_bwd");
}

int lf_mul_f32(const float* a, const float* b, float* out, size_t count, lf_stream_t stream) {
    LF_REQUIRE(a && b && out, "lf_mul: null buffer");
    LF_REQUIRE(count > 0, "lf_mul: empty");
    mul_kernel<<<lf::stream_grid(count, kBlock), kBlock, 0, lf::as_stream(stream)>>>(a, b, out, count);
    return lf::check_launch("lf_mul");
}

size_t lf_adamw_workspace(int ntensors) {
    return ntensors > 0 ? (size_t)ntensors * kNormSplit * sizeof(double) : 0;
}

int lf_adamw_step_f32(float* param, const float* grad, float* m, float* v, float* ema,
                      const long long* offsets, const float* l2, int ntensors, long long max_count,
                      float lr, float beta1, float beta2, float eps, float weight_decay,
                      float clipnorm, long long step, float ema_decay, int ema_copy,
                      float* norms_out, void* workspace, size_t ws_bytes, lf_stream_t stream) {
    LF_REQUIRE(param && grad && m && v && offsets && l2 && norms_out && workspace,
               "lf_adamw_step: null buffer");
    if (ws_bytes < lf_adamw_workspace(ntensors)) {
        lf::set_error("lf_adamw_step: workspace %zu < %zu", ws_bytes, lf_adamw_workspace(ntensors));
        return LF_ERR_WORKSPACE;
    }
    LF_REQUIRE(ntensors > 0 && ntensors <= 65535 && max_count > 0 && step >= 1,
               "lf_adamw_step: bad ntensors=%d max_count=%lld step=%lld", ntensors, max_count, step);
    hipStream_t st = lf::as_stream(stream);
    double* partial = static_cast<double*>(workspace);
    adam_norm_kernel<<<dim3(kNormSplit, ntensors), kBlock, 0, st>>>(param, grad, offsets, l2, partial);
    AdamArgs a;
    a.p = param; a.g = grad; a.m = m; a.v = v; a.ema = ema; a.offs = offsets; a.l2 = l2;
    a.norm_partial = partial;
    a.norms = norms_out;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
    a.clipnorm = clipnorm;
    const double b1p = pow((double)beta1, (double)step), b2p = pow((double)beta2, (double)step);
    a.alpha = (float)((double)lr * sqrt(1.0 - b2p) / (1.0 - b1p));
    a.ema_decay = ema_decay;
    a.ema_copy = ema_copy;
    const unsigned gx = lf::stream_grid((size_t)max_count, kBlock, 64);
    adam_step_kernel<<<dim3(gx, ntensors), kBlock, 0, st>>>(a);
    return lf::check_launch("lf_adamw_step");
}

int lf_ema_update_f32(float* ema, const float* w, size_t count, float decay, int copy,
                      lf_stream_t stream) {
    LF_REQUIRE(ema && w, "lf_ema_update: null buffer");
    LF_REQUIRE(count > 0, "lf_ema_update: empty");
    ema_kernel<<<lf::stream_grid(count, kBlock), kBlock, 0, lf::as_stream(stream)>>>(ema, w, count, decay,
                                                                                 copy);
    return lf::check_launch("lf_ema_update");
}

}  // extern "C"
