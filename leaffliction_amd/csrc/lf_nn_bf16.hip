// libleafhip — the plane kernels of the TRAINING step on bf16 activation / gradient storage
// (the mixed-precision step, srcs/cli/train.py:179-190; BASELINE configs[3]).  Same arithmetic
// as gap_kernel / tail_fwd_kernel / tail_bwd_kernel / bcast_planes_kernel of lf_nn.hip — fp32
// after widening the operands, BatchNorm / SE / dropout factors in fp32 — what differs is the
// storage type: every tensor these kernels read or write is bf16 NCHW, values are rounded
// (nearest even) exactly where they are stored, and every sum a later kernel relies on is taken
// over the ROUNDED values, so that forward statistics and backward sums describe the tensors as
// they sit in HBM.  All are HBM-bound streaming passes: 8-byte accesses (four bf16), one (n, c)
// plane per workgroup row, fixed-order reductions.
#include "lf_common.h"

namespace {

constexpr int kBlock = 256;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float up(unsigned bits16) { return __uint_as_float(bits16 << 16); }
__device__ __forceinline__ unsigned down(float v) { return (unsigned)__builtin_bit_cast(uint16_t, (__bf16)v); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

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

// out[plane] = mean_hw act(x*scale[c]+shift[c]); mask_sums[plane] = {count of x*scale+shift > 0,
// sum of x over those} (the SE squeeze of relu(BN2(y2)), and what BN2's backward needs from it).
// V = bf16 values per load: 8 (16-byte lanes: what the vector memory pipe needs to reach its rate,
// scripts/microbench/seg_bw.hip) when hw % 8 == 0, else 4.
template <int V>
__global__ __launch_bounds__(kBlock) void gap_stats_bf16_kernel(const uint16_t* __restrict__ x,
                                                                float* __restrict__ out, int hw, int c,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift, int relu,
                                                                float* __restrict__ mask_sums) {
    typedef unsigned uvec __attribute__((ext_vector_type(V / 2)));
    __shared__ float red[12];
    const size_t base = (size_t)blockIdx.x * hw;
    const bool pro = scale != nullptr;
    const float sc = pro ? scale[blockIdx.x % c] : 1.f, sh = pro ? shift[blockIdx.x % c] : 0.f;
    float acc[3] = {0.f, 0.f, 0.f};
    auto one = [&](unsigned bits) {
        const float xv = up(bits);
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
    const uvec* xv = reinterpret_cast<const uvec*>(x + base);
    for (int i = threadIdx.x; i < hw / V; i += kBlock) {
        const uvec v = xv[i];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < V / 2; ++e) s += one(v[e] & 0xffffu) + one(v[e] >> 16);
        acc[0] += s;
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

struct TailTrainArgs {
    const uint16_t* y;       // second conv of the block (raw, before its BatchNorm)
    const float* a_scale;    // its BatchNorm (batch statistics)
    const float* a_shift;
    const float* s;          // SE gate [n][c] or null
    const uint16_t* sc;      // shortcut tensor
    const float* sc_scale;   // its scale / shift (+ReLU), or null when it is final already
    const float* sc_shift;
    const float* drop;       // SpatialDropout2D keep-scale [n][c] or null
    int sc_relu;
    int c, h, w;
};

__device__ __forceinline__ float tail_r(const TailTrainArgs& t, unsigned yb, unsigned sb, float as, float ab,
                                        float sv, float ks, float kb) {
    float a = up(yb);
    if (t.a_scale) a = fmaxf(fmaf(a, as, ab), 0.f);
    float sh = up(sb);
    if (t.sc_scale) {
        sh = fmaf(sh, ks, kb);
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

// p = bf16(drop * maxpool2x2(relu(shortcut' + relu(BN(y)) * gate))), route byte per pooled value.
// One thread = two rows x V input pixels (V = 8: 16-byte loads, w % 8 == 0; else 4) = V/2 pooled values.
template <int V>
__global__ __launch_bounds__(kBlock) void tail_fwd_train_bf16_kernel(TailTrainArgs t, uint8_t* __restrict__ route,
                                                                     uint16_t* __restrict__ p) {
    typedef unsigned uvec __attribute__((ext_vector_type(V / 2)));
    typedef unsigned ovec __attribute__((ext_vector_type(V / 4)));
    const int plane = blockIdx.x, ch = plane % t.c;
    const float sv = t.s ? t.s[plane] : 1.f;
    const float as = t.a_scale ? t.a_scale[ch] : 1.f, ab = t.a_scale ? t.a_shift[ch] : 0.f;
    const float ks = t.sc_scale ? t.sc_scale[ch] : 1.f, kb = t.sc_scale ? t.sc_shift[ch] : 0.f;
    const float dv = t.drop ? t.drop[plane] : 1.f;
    const int h = t.h, w = t.w, ph = h / 2, pw = w / 2, pwv = w / V;
    const size_t base = (size_t)plane * h * w, pbase = (size_t)plane * ph * pw;
    for (int q = blockIdx.y * kBlock + threadIdx.x; q < ph * pwv; q += gridDim.y * kBlock) {
        const int py = q / pwv, pxv = q - py * pwv;
        float r[2][V];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const size_t o = base + (size_t)(2 * py + dy) * w + V * pxv;
            const uvec yv = *reinterpret_cast<const uvec*>(t.y + o);
            const uvec s4 = *reinterpret_cast<const uvec*>(t.sc + o);
#pragma unroll
            for (int e = 0; e < V / 2; ++e) {
                r[dy][2 * e] = tail_r(t, yv[e] & 0xffffu, s4[e] & 0xffffu, as, ab, sv, ks, kb);
                r[dy][2 * e + 1] = tail_r(t, yv[e] >> 16, s4[e] >> 16, as, ab, sv, ks, kb);
            }
        }
        const size_t po = pbase + (size_t)py * pw + (V / 2) * pxv;
        ovec packed;
        unsigned codes = 0;
#pragma unroll
        for (int k = 0; k < V / 2; k += 2) {
            float m0, m1;
            const unsigned c0 = tail_code(r[0][2 * k], r[0][2 * k + 1], r[1][2 * k], r[1][2 * k + 1], m0);
            const unsigned c1 = tail_code(r[0][2 * k + 2], r[0][2 * k + 3], r[1][2 * k + 2], r[1][2 * k + 3], m1);
            codes |= (c0 | (c1 << 8)) << (8 * k);
            packed[k / 2] = down(m0 * dv) | down(m1 * dv) << 16;
        }
        if (V == 8) {
            *reinterpret_cast<unsigned*>(route + po) = codes;
        } else {
            *reinterpret_cast<uint16_t*>(route + po) = (uint16_t)codes;
        }
        *reinterpret_cast<ovec*>(p + po) = packed;
    }
}

// dr = bf16(dp*drop) routed to the recorded position of each 2x2 window when its maximum was > 0.
// Per plane: ds = sum dr*a with a = relu(y*a_scale+a_shift), plane_sums = {sum dr*[a>0],
// sum dr*[a>0]*y} (BatchNorm-2's backward sums), sc_sums = {sum dr, sum dr*sc_y} (the projection
// shortcut's BatchNorm).  The sums are over the rounded dr.  One thread = V/2 pooled values ->
// two rows x V gradient pixels (V = 8: 16-byte stores, w % 8 == 0; else 4).
template <int V>
__global__ __launch_bounds__(kBlock) void tail_bwd_bf16_kernel(const uint16_t* __restrict__ dp,
                                                               const uint8_t* __restrict__ route,
                                                               const uint16_t* __restrict__ y,
                                                               const float* __restrict__ a_scale,
                                                               const float* __restrict__ a_shift,
                                                               const float* __restrict__ drop,
                                                               uint16_t* __restrict__ dr, float* __restrict__ ds,
                                                               float* __restrict__ plane_sums,
                                                               const uint16_t* __restrict__ sc_y,
                                                               float* __restrict__ sc_sums, int c, int h, int w) {
    typedef unsigned uvec __attribute__((ext_vector_type(V / 2)));
    typedef unsigned ivec __attribute__((ext_vector_type(V / 4)));
    __shared__ float red[20];
    const int plane = blockIdx.x, ch = plane % c;
    const float dv = drop ? drop[plane] : 1.f;
    const float as = a_scale ? a_scale[ch] : 1.f, ab = a_scale ? a_shift[ch] : 0.f;
    const int ph = h / 2, pw = w / 2, pwv = w / V;
    const size_t base = (size_t)plane * h * w, pbase = (size_t)plane * ph * pw;
    const bool sums = ds != nullptr || plane_sums != nullptr;
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    auto tally = [&](float gg, size_t pos) {
        if (sc_y != nullptr && gg != 0.f) {  // the shortcut branch's (projection) BN: no mask
            acc[3] += gg;
            acc[4] += gg * up(sc_y[pos]);
        }
        if (sums && gg != 0.f) {
            const float yv = up(y[pos]);
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
    for (int q = threadIdx.x; q < ph * pwv; q += kBlock) {
        const int py = q / pwv, pxv = q - py * pwv;
        const size_t po = pbase + (size_t)py * pw + (V / 2) * pxv;
        unsigned codes;
        if (V == 8) codes = *reinterpret_cast<const unsigned*>(route + po);
        else codes = *reinterpret_cast<const uint16_t*>(route + po);
        const ivec g2 = *reinterpret_cast<const ivec*>(dp + po);
        const size_t o0 = base + (size_t)(2 * py) * w + V * pxv, o1 = o0 + w;
        uvec top, bot;
#pragma unroll
        for (int k = 0; k < V / 2; ++k) {   // pooled value k -> input columns 2k, 2k+1
            const unsigned cd = (codes >> (8 * k)) & 0xffu;
            const unsigned gw = g2[k / 2];
            const unsigned gb = (cd & 4u) ? down(up((k & 1) ? gw >> 16 : gw & 0xffffu) * dv) : 0u;
            const unsigned b = cd & 3u;
            top[k] = (b == 0 ? gb : 0u) | (b == 1 ? gb << 16 : 0u);
            bot[k] = (b == 2 ? gb : 0u) | (b == 3 ? gb << 16 : 0u);
            tally(up(gb), (b < 2 ? o0 : o1) + 2 * k + (b & 1u));
        }
        *reinterpret_cast<uvec*>(dr + o0) = top;
        *reinterpret_cast<uvec*>(dr + o1) = bot;
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

// out[plane][:] = bf16(v[plane] * scale)  (the head's gradient spread over the last pooled map)
__global__ __launch_bounds__(kBlock) void bcast_planes_bf16_kernel(const float* __restrict__ v,
                                                                   uint16_t* __restrict__ out, int hw4,
                                                                   float scale) {
    const unsigned b = down(v[blockIdx.x] * scale);
    u32x2 o;
    o.x = o.y = b | b << 16;
    u32x2* dst = reinterpret_cast<u32x2*>(out + (size_t)blockIdx.x * hw4 * 4);
    for (int i = blockIdx.y * kBlock + threadIdx.x; i < hw4; i += gridDim.y * kBlock) dst[i] = o;
}

__global__ __launch_bounds__(kBlock) void cast_f32_bf16_kernel(const float* __restrict__ in,
                                                               uint16_t* __restrict__ out, size_t count) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (size_t)gridDim.x * kBlock)
        out[i] = (uint16_t)down(in[i]);
}

__global__ __launch_bounds__(kBlock) void cast_bf16_f32_kernel(const uint16_t* __restrict__ in,
                                                               float* __restrict__ out, size_t count) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (size_t)gridDim.x * kBlock)
        out[i] = up(in[i]);
}

inline unsigned plane_grid(int items) { return lf::stream_grid((size_t)items, kBlock, 64); }
inline bool al8(const void* p) { return (reinterpret_cast<size_t>(p) & 7) == 0; }

}  // namespace

extern "C" {

int lf_gap_stats_bf16(const uint16_t* x, float* out, float* mask_sums, int n, int c, int hw, const float* scale,
                      const float* shift, int relu, lf_stream_t stream) {
    LF_REQUIRE(x && out, "lf_gap_stats_bf16: null buffer");
    LF_REQUIRE(n > 0 && c > 0 && hw > 0 && hw % 4 == 0, "lf_gap_stats_bf16: bad dims n=%d c=%d hw=%d (hw %% 4 == 0)",
               n, c, hw);
    LF_REQUIRE((scale == nullptr) == (shift == nullptr), "lf_gap_stats_bf16: scale/shift must both be set");
    LF_REQUIRE(mask_sums == nullptr || scale != nullptr, "lf_gap_stats_bf16: mask_sums needs scale/shift");
    LF_REQUIRE(al8(x), "lf_gap_stats_bf16: x must be 8-byte aligned");
    if (hw % 8 == 0 && (reinterpret_cast<size_t>(x) & 15) == 0)
        gap_stats_bf16_kernel<8><<<n * c, kBlock, 0, lf::as_stream(stream)>>>(x, out, hw, c, scale, shift, relu, mask_sums);
    else
        gap_stats_bf16_kernel<4><<<n * c, kBlock, 0, lf::as_stream(stream)>>>(x, out, hw, c, scale, shift, relu, mask_sums);
    return lf::check_launch("lf_gap_stats_bf16");
}

int lf_block_tail_fwd_train_bf16(const uint16_t* y, const float* a_scale, const float* a_shift, const float* s,
                                 const uint16_t* sc, const float* sc_scale, const float* sc_shift, int sc_relu,
                                 const float* drop, uint8_t* route, uint16_t* pooled, int n, int c, int h, int w,
                                 lf_stream_t stream) {
    LF_REQUIRE(y && sc && route && pooled, "lf_block_tail_fwd_train_bf16: null buffer");
    LF_REQUIRE(n > 0 && c > 0 && h > 1 && w > 3 && h % 2 == 0 && w % 4 == 0 && (long long)n * c < (1LL << 31),
               "lf_block_tail_fwd_train_bf16: bad dims n=%d c=%d h=%d w=%d (h even, w %% 4 == 0)", n, c, h, w);
    LF_REQUIRE((sc_scale == nullptr) == (sc_shift == nullptr), "lf_block_tail_fwd_train_bf16: sc_scale/sc_shift");
    LF_REQUIRE((a_scale == nullptr) == (a_shift == nullptr), "lf_block_tail_fwd_train_bf16: a_scale/a_shift");
    LF_REQUIRE(al8(y) && al8(sc) && (reinterpret_cast<size_t>(pooled) & 3) == 0 &&
                   (reinterpret_cast<size_t>(route) & 1) == 0,
               "lf_block_tail_fwd_train_bf16: misaligned buffer");
    TailTrainArgs t{y, a_scale, a_shift, s, sc, sc_scale, sc_shift, drop, sc_relu, c, h, w};
    const bool wide = w % 8 == 0 && ((reinterpret_cast<size_t>(y) | reinterpret_cast<size_t>(sc)) & 15) == 0 &&
                      (reinterpret_cast<size_t>(pooled) & 7) == 0 && (reinterpret_cast<size_t>(route) & 3) == 0;
    if (wide)
        tail_fwd_train_bf16_kernel<8><<<dim3(n * c, plane_grid((h / 2) * (w / 8))), kBlock, 0, lf::as_stream(stream)>>>(
            t, route, pooled);
    else
        tail_fwd_train_bf16_kernel<4><<<dim3(n * c, plane_grid((h / 2) * (w / 4))), kBlock, 0, lf::as_stream(stream)>>>(
            t, route, pooled);
    return lf::check_launch("lf_block_tail_fwd_train_bf16");
}

int lf_block_tail_bwd_bf16(const uint16_t* dp, const uint8_t* route, const uint16_t* y, const float* a_scale,
                           const float* a_shift, const float* drop, uint16_t* dr, float* ds, float* plane_sums,
                           const uint16_t* sc_y, float* sc_sums, int n, int c, int h, int w, lf_stream_t stream) {
    LF_REQUIRE((sc_y == nullptr) == (sc_sums == nullptr), "lf_block_tail_bwd_bf16: sc_y and sc_sums go together");
    LF_REQUIRE(dp && route && dr, "lf_block_tail_bwd_bf16: null buffer");
    LF_REQUIRE(plane_sums == nullptr || (y != nullptr && a_scale != nullptr),
               "lf_block_tail_bwd_bf16: plane_sums needs y and a_scale/a_shift");
    LF_REQUIRE(n > 0 && c > 0 && h > 1 && w > 3 && h % 2 == 0 && w % 4 == 0,
               "lf_block_tail_bwd_bf16: bad dims n=%d c=%d h=%d w=%d (h even, w %% 4 == 0)", n, c, h, w);
    LF_REQUIRE((y == nullptr) == (ds == nullptr && plane_sums == nullptr),
               "lf_block_tail_bwd_bf16: y goes with ds / plane_sums");
    LF_REQUIRE((a_scale == nullptr) == (a_shift == nullptr), "lf_block_tail_bwd_bf16: a_scale/a_shift");
    LF_REQUIRE(al8(dr) && (reinterpret_cast<size_t>(dp) & 3) == 0, "lf_block_tail_bwd_bf16: misaligned buffer");
    const bool wide = w % 8 == 0 && (reinterpret_cast<size_t>(dr) & 15) == 0 && (reinterpret_cast<size_t>(dp) & 7) == 0 &&
                      (reinterpret_cast<size_t>(route) & 3) == 0;
    if (wide)
        tail_bwd_bf16_kernel<8><<<n * c, kBlock, 0, lf::as_stream(stream)>>>(dp, route, y, a_scale, a_shift, drop, dr,
                                                                            ds, plane_sums, sc_y, sc_sums, c, h, w);
    else
        tail_bwd_bf16_kernel<4><<<n * c, kBlock, 0, lf::as_stream(stream)>>>(dp, route, y, a_scale, a_shift, drop, dr,
                                                                            ds, plane_sums, sc_y, sc_sums, c, h, w);
    return lf::check_launch("lf_block_tail_bwd_bf16");
}

int lf_bcast_planes_bf16(const float* v, uint16_t* out, int planes, int hw, float scale, lf_stream_t stream) {
    LF_REQUIRE(v && out, "lf_bcast_planes_bf16: null buffer");
    LF_REQUIRE(planes > 0 && hw > 0 && hw % 4 == 0, "lf_bcast_planes_bf16: bad dims planes=%d hw=%d (hw %% 4 == 0)",
               planes, hw);
    LF_REQUIRE(al8(out), "lf_bcast_planes_bf16: out must be 8-byte aligned");
    bcast_planes_bf16_kernel<<<dim3(planes, plane_grid(hw / 4)), kBlock, 0, lf::as_stream(stream)>>>(v, out, hw / 4,
                                                                                                scale);
    return lf::check_launch("lf_bcast_planes_bf16");
}

int lf_cast_f32_bf16(const float* in, uint16_t* out, size_t count, lf_stream_t stream) {
    LF_REQUIRE(in && out && count > 0, "lf_cast_f32_bf16: null / empty buffer");
    cast_f32_bf16_kernel<<<lf::stream_grid(count, kBlock), kBlock, 0, lf::as_stream(stream)>>>(in, out, count);
    return lf::check_launch("lf_cast_f32_bf16");
}

int lf_cast_bf16_f32(const uint16_t* in, float* out, size_t count, lf_stream_t stream) {
    LF_REQUIRE(in && out && count > 0, "lf_cast_bf16_f32: null / empty buffer");
    cast_bf16_f32_kernel<<<lf::stream_grid(count, kBlock), kBlock, 0, lf::as_stream(stream)>>>(in, out, count);
    return lf::check_launch("lf_cast_bf16_f32");
}

}  // extern "C"
