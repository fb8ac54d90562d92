// libleafhip — conv2d forward / dgrad / wgrad as implicit GEMM on the fp32 MFMA
// (v_mfma_f32_32x32x2_f32: exact f32, bitwise an fmaf chain; 64 FLOP/clk/SIMD).
//
// Layouts: activations NCHW f32; conv weights "IKO" = [Cin][kh*kw][Cout] (the GEMM's
// K-major order, Cout contiguous).  Stride 1, "same" zero padding, no bias
// (keras Conv2D(filters, 3|1, padding="same", use_bias=False), srcs/model/cnn.py:27-29,44).
//
// Forward / dgrad kernel (conv_mfma_kernel):  D[co][pixel] = sum_k W[co][k] X[k][pixel],
//   k = (ci, tap).  A operand = weights (lane -> co), B operand = pixels (lane -> pixel),
//   so one accumulator register holds 32 consecutive pixels of one output channel and the
//   NCHW store is 128-byte contiguous per half-wave.
//   A workgroup (4 waves) owns a TWxTH pixel tile (flat-indexed 32-pixel blocks, so tile
//   shapes like 28x8 work) x CT output channels; the input patch (+halo) and the weight
//   slice for KC input channels are staged in LDS per K-chunk; several workgroups are
//   resident per CU (2-4 waves per SIMD) so one workgroup's staging overlaps another's MFMAs.
//   An optional prologue applies y = relu(x*scale[c]+shift[c]) to the input while staging
//   (BatchNorm+ReLU of the producer fused into the consumer; zero padding stays zero).
//
// wgrad kernel (wgrad_mfma_kernel): dW[ci][tap][co] = sum_pixels X[ci][p+tap] dY[co][p],
//   K = pixels, split across workgroups (and across the waves of a workgroup); every wave
//   writes a partial slab to a workspace and a reduce kernel sums the slabs in a fixed order
//   (deterministic, no float atomics).
#include "lf_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int kKC = 8;  // input channels per K-chunk

struct ConvArgs {
    const float* x;
    const float* w;      // [Cin][TAPS][Cout]
    float* y;
    const float* in_scale;  // optional prologue (nullptr = none)
    const float* in_shift;
    int n, cin, cout, h, wd;
    int tiles_x, tiles_y;
    int in_relu;
    int accumulate;  // y += conv(...) instead of y = conv(...)
};

// ---------------------------------------------------------------------------
// forward / dgrad
// ---------------------------------------------------------------------------
// TAPS: 9 (3x3) or 1 (1x1).  Tile TW x TH pixels = NPB blocks of 32 (flat index).
// Waves: WCO x WPX = 4; each wave computes MB cout-blocks x NB pixel-blocks.
// min waves/SIMD asked of the register allocator: accumulators (AGPRs) + VGPRs share one
// 512-entry file per SIMD lane, so <=64 accumulators leave room for 3 waves, more for 2.
template <int TAPS, int TW, int TH, int WCO, int MB, int WPX, int NB>
__global__ __launch_bounds__(kThreads, (MB * NB * 16 <= 32 ? 4 : (MB * NB * 16 <= 64 ? 3 : 2))) void conv_mfma_kernel(ConvArgs p) {
    constexpr int NPB = TW * TH / 32;
    static_assert(TW * TH % 32 == 0, "tile must be whole 32-pixel blocks");
    static_assert(WCO * WPX == 4 && WPX * NB == NPB, "wave decomposition");
    constexpr int CT = 32 * WCO * MB;
    constexpr int HALO = TAPS == 9 ? 1 : 0;
    constexpr int PW = TW + 2 * HALO, PH = TH + 2 * HALO, PP = PW * PH;
    constexpr int PATCH = kKC * PP;
    constexpr int WSZ = kKC * TAPS * CT;

    __shared__ float lds[PATCH + WSZ];
    float* lp = lds;
    float* lw = lds + PATCH;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wave_co = wid % WCO, wave_px = wid / WCO;
    const int tile = blockIdx.x;
    const int tx0 = (tile % p.tiles_x) * TW, ty0 = (tile / p.tiles_x) * TH;
    const int co0 = blockIdx.y * CT;
    const int n = blockIdx.z;
    const size_t hw = (size_t)p.h * p.wd;
    const float* xin = p.x + (size_t)n * p.cin * hw;

    // per-lane LDS read bases
    const int khalf = lane >> 5, j = lane & 31;
    int bbase[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int f = (wave_px * NB + nb) * 32 + j;
        bbase[nb] = khalf * PP + (f / TW) * PW + (f % TW);
    }
    const int abase = khalf * TAPS * CT + wave_co * MB * 32 + j;

    f32x16 acc[MB][NB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][nb][r] = 0.f;

    const bool pro = p.in_scale != nullptr;
    const unsigned uhw = (unsigned)hw;

    // Staging map: a thread owns one patch position (two when the patch has more than 256)
    // and walks the chunk's channels, so the position -> (row, col, bounds, global offset)
    // arithmetic is done once per kernel and every load is base + c*H*W.
    constexpr int SLOTS = PP <= 64 ? 64 : (PP <= 128 ? 128 : 256);
    constexpr int G = kThreads / SLOTS;
    constexpr int ROUNDS = (PP + SLOTS - 1) / SLOTS;
    const int slot = tid % SLOTS, grp = tid / SLOTS;
    unsigned goff[ROUNDS];
    bool inb[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int pos = slot + r * SLOTS;
        const int py = pos / PW, px = pos - py * PW;
        const int gy = ty0 + py - HALO, gx = tx0 + px - HALO;
        inb[r] = pos < PP && gy >= 0 && gy < p.h && gx >= 0 && gx < p.wd;
        goff[r] = inb[r] ? (unsigned)gy * (unsigned)p.wd + (unsigned)gx : 0u;
    }
    constexpr int WROWS = kKC * TAPS, RPP = kThreads / CT;
    const int wcol = tid % CT, wrow0 = tid / CT;
    const bool wcol_ok = co0 + wcol < p.cout;

    auto stage_chunk = [&](int c0) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int pos = slot + r * SLOTS;
            if (pos < PP) {
#pragma unroll 4
                for (int kc = grp; kc < kKC; kc += G) {
                    const int c = c0 + kc;
                    float v = 0.f;
                    if (inb[r] && c < p.cin) {
                        v = xin[(unsigned)c * uhw + goff[r]];
                        if (pro) {
                            v = fmaf(v, p.in_scale[c], p.in_shift[c]);
                            if (p.in_relu) v = fmaxf(v, 0.f);
                        }
                    }
                    lp[kc * PP + pos] = v;
                }
            }
        }
        const int wvalid = (p.cin - c0) * TAPS;  // rows of this chunk that exist
#pragma unroll 4
        for (int row = wrow0; row < WROWS; row += RPP) {
            float v = 0.f;
            if (wcol_ok && row < wvalid)
                v = p.w[((unsigned)c0 * TAPS + row) * (unsigned)p.cout + (unsigned)(co0 + wcol)];
            lw[row * CT + wcol] = v;
        }
    };

    const int nchunks = (p.cin + kKC - 1) / kKC;
    for (int ch = 0; ch < nchunks; ++ch) {
        __syncthreads();  // previous chunk's LDS reads are done
        stage_chunk(ch * kKC);
        __syncthreads();
#pragma unroll
        for (int cp = 0; cp < kKC / 2; ++cp) {
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int dy = TAPS == 9 ? tap / 3 : 0, dx = TAPS == 9 ? tap % 3 : 0;
                float a[MB], b[NB];
#pragma unroll
                for (int m = 0; m < MB; ++m) a[m] = lw[abase + (2 * cp * TAPS + tap) * CT + m * 32];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) b[nb] = lp[bbase[nb] + 2 * cp * PP + dy * PW + dx];
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[m][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[nb], acc[m][nb], 0,
                                                                          0, 0);
            }
        }
    }

    // epilogue: D[row = co][col = pixel]; row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* yout = p.y + (size_t)n * p.cout * hw;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int f = (wave_px * NB + nb) * 32 + j;
        const int oy = ty0 + f / TW, ox = tx0 + f % TW;
        if (oy >= p.h || ox >= p.wd) continue;
        const size_t pix = (size_t)oy * p.wd + ox;
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const int cb = co0 + (wave_co * MB + m) * 32 + 4 * khalf;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cb + (r & 3) + 8 * (r >> 2);
                if (co < p.cout) {
                    float* dst = yout + (size_t)co * hw + pix;
                    *dst = p.accumulate ? *dst + acc[m][nb][r] : acc[m][nb][r];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// wgrad
// ---------------------------------------------------------------------------
// D[ci][co] (per tap) += sum over pixel pairs.  A operand = X (lane -> ci, k = pixel),
// B operand = dY (lane -> co).  A workgroup owns a (32*WCI ci) x (32*WCO co) weight block and
// a range of (image, tile) work items; its 4 waves are WCI x WCO x KSPL with the KSPL waves
// splitting the rows of each tile.
struct WgradArgs {
    const float* x;   // [N][Cin][H][W]
    const float* dy;  // [N][Cout][H][W]
    float* part;      // [splits][Cin][TAPS][Cout] partial slabs
    const float* in_scale;
    const float* in_shift;
    int n, cin, cout, h, wd;
    int tiles_x, tiles_y, items, items_per_split;
    int in_relu;
};

template <int TAPS, int TW, int TH, int WCI, int WCO, int KSPL>
__global__ __launch_bounds__(kThreads, (TAPS == 9 ? 2 : 4)) void wgrad_mfma_kernel(WgradArgs p) {
    static_assert(WCI * WCO * KSPL == 4, "wave decomposition");
    static_assert(TH % KSPL == 0 && TW % 2 == 0, "rows split across waves, pixel pairs");
    constexpr int HALO = TAPS == 9 ? 1 : 0;
    constexpr int PW = TW + 2 * HALO, PH = TH + 2 * HALO;
    constexpr int PP = (PW * PH) | 1;  // odd plane pitch: 32 lanes on 32 channels hit 32 banks
    constexpr int DP = (TW * TH) | 1;
    constexpr int CI_T = 32 * WCI, CO_T = 32 * WCO;
    constexpr int XSZ = CI_T * PP, DSZ = CO_T * DP;
    __shared__ float lds[XSZ + DSZ];
    float* lx = lds;
    float* ld = lds + XSZ;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int w_ci = wid % WCI, w_co = (wid / WCI) % WCO, w_k = wid / (WCI * WCO);
    const int ci0 = blockIdx.y * CI_T, co0 = blockIdx.z * CO_T;
    const int khalf = lane >> 5, j = lane & 31;
    const size_t hw = (size_t)p.h * p.wd;

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int abase = (w_ci * 32 + j) * PP + khalf;  // + row*PW + x + tap offset
    const int bbase = (w_co * 32 + j) * DP + khalf;
    const bool pro = p.in_scale != nullptr;

    const unsigned uhw = (unsigned)hw;
    // thread -> one tile position, walking channels (see conv_mfma_kernel's staging map)
    constexpr int XPOS = PW * PH, DPOS = TW * TH;
    constexpr int XSLOTS = XPOS <= 64 ? 64 : (XPOS <= 128 ? 128 : 256);
    constexpr int DSLOTS = DPOS <= 64 ? 64 : (DPOS <= 128 ? 128 : 256);
    constexpr int XG = kThreads / XSLOTS, DG = kThreads / DSLOTS;
    static_assert(XPOS <= 256 && DPOS <= 256, "wgrad tiles are at most 256 positions");
    const int xslot = tid % XSLOTS, xgrp = tid / XSLOTS;
    const int dslot = tid % DSLOTS, dgrp = tid / DSLOTS;
    const int xpy = xslot / PW, xpx = xslot - xpy * PW;
    const int dpy = dslot / TW, dpx = dslot - dpy * TW;

    auto stage_item = [&](int item) {
        const int n = item / (p.tiles_x * p.tiles_y), t = item - n * (p.tiles_x * p.tiles_y);
        const int tx0 = (t % p.tiles_x) * TW, ty0 = (t / p.tiles_x) * TH;
        const float* xin = p.x + (size_t)n * p.cin * hw;
        const float* din = p.dy + (size_t)n * p.cout * hw;
        if (xslot < XPOS) {
            const int gy = ty0 + xpy - HALO, gx = tx0 + xpx - HALO;
            const bool ok = gy >= 0 && gy < p.h && gx >= 0 && gx < p.wd;
            const unsigned off = ok ? (unsigned)gy * (unsigned)p.wd + (unsigned)gx : 0u;
#pragma unroll 4
            for (int c = xgrp; c < CI_T; c += XG) {
                const int gc = ci0 + c;
                float v = 0.f;
                if (ok && gc < p.cin) {
                    v = xin[(unsigned)gc * uhw + off];
                    if (pro) {
                        v = fmaf(v, p.in_scale[gc], p.in_shift[gc]);
                        if (p.in_relu) v = fmaxf(v, 0.f);
                    }
                }
                lx[c * PP + xslot] = v;
            }
        }
        if (dslot < DPOS) {
            const int gy = ty0 + dpy, gx = tx0 + dpx;
            const bool ok = gy < p.h && gx < p.wd;
            const unsigned off = ok ? (unsigned)gy * (unsigned)p.wd + (unsigned)gx : 0u;
#pragma unroll 4
            for (int c = dgrp; c < CO_T; c += DG) {
                const int gc = co0 + c;
                float v = 0.f;
                if (ok && gc < p.cout) v = din[(unsigned)gc * uhw + off];
                ld[c * DP + dslot] = v;
            }
        }
    };

    const int first = blockIdx.x * p.items_per_split;
    const int last = min(first + p.items_per_split, p.items);
    for (int item = first; item < last; ++item) {
        __syncthreads();
        stage_item(item);
        __syncthreads();
        constexpr int ROWS = TH / KSPL;
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            const int row = w_k * ROWS + rr;
#pragma unroll 4
            for (int xx = 0; xx < TW; xx += 2) {
                const float b = ld[bbase + row * TW + xx];
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const int dy = TAPS == 9 ? t / 3 : 0, dx = TAPS == 9 ? t % 3 : 0;
                    const float a = lx[abase + (row + dy) * PW + xx + dx];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                }
            }
        }
    }

    // every K-split wave writes its own partial slab (no LDS reduction, no atomics)
    {
        float* out = p.part + ((size_t)blockIdx.x * KSPL + w_k) * p.cin * TAPS * p.cout;
        const int co = co0 + w_co * 32 + j;
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + w_ci * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (ci < p.cin && co < p.cout) out[((size_t)ci * TAPS + t) * p.cout + co] = acc[t][r];
            }
    }
}

// sum the per-split slabs: dw[i] = sum_s part[s][i]  (fixed order -> deterministic)
__global__ __launch_bounds__(kThreads) void slab_reduce_kernel(const float* __restrict__ part,
                                                               float* __restrict__ dw, size_t count,
                                                               int splits, float beta) {
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < count;
         i += (size_t)gridDim.x * kThreads) {
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += part[(size_t)k * count + i];
        dw[i] = beta != 0.f ? fmaf(beta, dw[i], s) : s;
    }
}

// [Cin][T][Cout] -> dgrad weights [Cout][T flipped][Cin]
__global__ __launch_bounds__(kThreads) void weight_dgrad_kernel(const float* __restrict__ w,
                                                                float* __restrict__ wt, int cin,
                                                                int taps, int cout) {
    const int total = cin * taps * cout;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < total; i += gridDim.x * kThreads) {
        const int ci = i % cin, t = (i / cin) % taps, co = i / (cin * taps);
        wt[i] = w[((size_t)ci * taps + (taps - 1 - t)) * cout + co];
    }
}

// ---------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------
struct FwdVariant {
    int tw, th, ct;
};
constexpr FwdVariant kFwdVariants[] = {{32, 8, 32}, {32, 8, 64}, {16, 16, 32},
                                       {16, 16, 64}, {28, 8, 128}, {32, 8, 128}};
constexpr int kNumFwd = sizeof(kFwdVariants) / sizeof(kFwdVariants[0]);

inline long long padded_work(const FwdVariant& v, int h, int w, int cout) {
    const long long tx = (w + v.tw - 1) / v.tw, ty = (h + v.th - 1) / v.th,
                    tc = (cout + v.ct - 1) / v.ct;
    return tx * v.tw * ty * v.th * tc * v.ct;
}

template <int TAPS>
int launch_fwd(int variant, const ConvArgs& a, dim3 grid, hipStream_t s) {
    switch (variant) {
        case 0: conv_mfma_kernel<TAPS, 32, 8, 1, 1, 4, 2><<<grid, kThreads, 0, s>>>(a); break;
        case 1: conv_mfma_kernel<TAPS, 32, 8, 1, 2, 4, 2><<<grid, kThreads, 0, s>>>(a); break;
        case 2: conv_mfma_kernel<TAPS, 16, 16, 1, 1, 4, 2><<<grid, kThreads, 0, s>>>(a); break;
        case 3: conv_mfma_kernel<TAPS, 16, 16, 1, 2, 4, 2><<<grid, kThreads, 0, s>>>(a); break;
        case 4: conv_mfma_kernel<TAPS, 28, 8, 4, 1, 1, 7><<<grid, kThreads, 0, s>>>(a); break;
        case 5: conv_mfma_kernel<TAPS, 32, 8, 2, 2, 2, 4><<<grid, kThreads, 0, s>>>(a); break;
        default: return LF_ERR_INVALID;
    }
    return LF_OK;
}

struct WgVariant {
    int tw, th, ci_t, co_t, kspl;
};
constexpr WgVariant kWgVariants[] = {{32, 4, 32, 32, 4}, {16, 8, 32, 64, 2}, {16, 4, 64, 64, 1},
                                     {28, 2, 64, 64, 1}, {32, 4, 32, 64, 2}};

template <int TAPS>
int launch_wgrad(int variant, const WgradArgs& a, dim3 grid, hipStream_t s) {
    switch (variant) {
        case 0: wgrad_mfma_kernel<TAPS, 32, 4, 1, 1, 4><<<grid, kThreads, 0, s>>>(a); break;
        case 1: wgrad_mfma_kernel<TAPS, 16, 8, 1, 2, 2><<<grid, kThreads, 0, s>>>(a); break;
        case 2: wgrad_mfma_kernel<TAPS, 16, 4, 2, 2, 1><<<grid, kThreads, 0, s>>>(a); break;
        case 3: wgrad_mfma_kernel<TAPS, 28, 2, 2, 2, 1><<<grid, kThreads, 0, s>>>(a); break;
        case 4: wgrad_mfma_kernel<TAPS, 32, 4, 1, 2, 2><<<grid, kThreads, 0, s>>>(a); break;
        default: return LF_ERR_INVALID;
    }
    return LF_OK;
}

struct WgPlan {
    int variant, tiles_x, tiles_y, items, splits, items_per_split, gy, gz, slabs;
};

WgPlan plan_wgrad(int n, int cin, int cout, int h, int w) {
    WgPlan best{};
    long long best_cost = -1;
    for (int v = 0; v < (int)(sizeof(kWgVariants) / sizeof(kWgVariants[0])); ++v) {
        const WgVariant& k = kWgVariants[v];
        const long long tx = (w + k.tw - 1) / k.tw, ty = (h + k.th - 1) / k.th;
        const long long gy = (cin + k.ci_t - 1) / k.ci_t, gz = (cout + k.co_t - 1) / k.co_t;
        const long long cost = tx * k.tw * ty * k.th * gy * k.ci_t * gz * k.co_t;
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best.variant = v;
            best.tiles_x = (int)tx;
            best.tiles_y = (int)ty;
            best.gy = (int)gy;
            best.gz = (int)gz;
        }
    }
    best.items = n * best.tiles_x * best.tiles_y;
    // ~4 workgroups per CU overall; every split gets the same number of items
    int splits = (256 * 4) / (best.gy * best.gz);
    if (splits < 1) splits = 1;
    if (splits > best.items) splits = best.items;
    best.items_per_split = (best.items + splits - 1) / splits;
    best.splits = (best.items + best.items_per_split - 1) / best.items_per_split;
    best.slabs = best.splits * kWgVariants[best.variant].kspl;
    return best;
}

}  // namespace

extern "C" {

int lf_conv2d_variant(int h, int wd, int cout);

int lf_conv2d_f32(const float* x, const float* w, float* y, int n, int cin, int h, int wd, int cout,
                  int ksize, const float* in_scale, const float* in_shift, int in_relu,
                  int accumulate, lf_stream_t stream) {
    LF_REQUIRE(x && w && y, "lf_conv2d: null buffer");
    LF_REQUIRE(n > 0 && cin > 0 && cout > 0 && h > 0 && wd > 0,
               "lf_conv2d: bad dims n=%d cin=%d cout=%d h=%d w=%d", n, cin, cout, h, wd);
    LF_REQUIRE(ksize == 3 || ksize == 1, "lf_conv2d: ksize must be 1 or 3 (got %d)", ksize);
    LF_REQUIRE((in_scale == nullptr) == (in_shift == nullptr),
               "lf_conv2d: in_scale/in_shift must both be set");
    LF_REQUIRE(n <= 65535, "lf_conv2d: batch too large for grid.z");
    const int best = lf_conv2d_variant(h, wd, cout);
    const FwdVariant& v = kFwdVariants[best];
    ConvArgs a;
    a.x = x; a.w = w; a.y = y; a.in_scale = in_scale; a.in_shift = in_shift;
    a.n = n; a.cin = cin; a.cout = cout; a.h = h; a.wd = wd;
    a.tiles_x = (wd + v.tw - 1) / v.tw;
    a.tiles_y = (h + v.th - 1) / v.th;
    a.in_relu = in_relu;
    a.accumulate = accumulate;
    dim3 grid(a.tiles_x * a.tiles_y, (cout + v.ct - 1) / v.ct, n);
    hipStream_t s = lf::as_stream(stream);
    const int rc = ksize == 3 ? launch_fwd<9>(best, a, grid, s) : launch_fwd<1>(best, a, grid, s);
    if (rc != LF_OK) return rc;
    return lf::check_launch("lf_conv2d");
}

int lf_conv2d_variant(int h, int wd, int cout) {
    int best = 0;
    long long bw = -1;
    for (int v = 0; v < kNumFwd; ++v) {
        const long long c = padded_work(kFwdVariants[v], h, wd, cout);
        if (bw < 0 || c < bw) {
            bw = c;
            best = v;
        }
    }
    return best;
}

int lf_conv2d_wgrad_variant(int n, int cin, int h, int wd, int cout) {
    if (n <= 0 || cin <= 0 || cout <= 0 || h <= 0 || wd <= 0) return -1;
    return plan_wgrad(n, cin, cout, h, wd).variant;
}

int lf_conv2d_dgrad_weights_f32(const float* w, float* wt, int cin, int ksize, int cout,
                                lf_stream_t stream) {
    LF_REQUIRE(w && wt, "lf_conv2d_dgrad_weights: null buffer");
    LF_REQUIRE(cin > 0 && cout > 0 && (ksize == 1 || ksize == 3), "lf_conv2d_dgrad_weights: bad dims");
    const int total = cin * ksize * ksize * cout;
    weight_dgrad_kernel<<<lf::stream_grid(total, kThreads), kThreads, 0, lf::as_stream(stream)>>>(
        w, wt, cin, ksize * ksize, cout);
    return lf::check_launch("lf_conv2d_dgrad_weights");
}

size_t lf_conv2d_wgrad_workspace(int n, int cin, int h, int wd, int cout, int ksize) {
    if (n <= 0 || cin <= 0 || cout <= 0 || h <= 0 || wd <= 0) return 0;
    const WgPlan pl = plan_wgrad(n, cin, cout, h, wd);
    return (size_t)pl.slabs * cin * ksize * ksize * cout * sizeof(float);
}

int lf_conv2d_wgrad_f32(const float* x, const float* dy, int n, int cin, int h, int wd, int cout,
                        int ksize, const float* in_scale, const float* in_shift, int in_relu,
                        void* workspace, size_t ws_bytes, lf_stream_t stream) {
    LF_REQUIRE(x && dy && workspace, "lf_conv2d_wgrad: null buffer");
    LF_REQUIRE(n > 0 && cin > 0 && cout > 0 && h > 0 && wd > 0,
               "lf_conv2d_wgrad: bad dims n=%d cin=%d cout=%d h=%d w=%d", n, cin, cout, h, wd);
    LF_REQUIRE(ksize == 3 || ksize == 1, "lf_conv2d_wgrad: ksize must be 1 or 3 (got %d)", ksize);
    LF_REQUIRE((in_scale == nullptr) == (in_shift == nullptr),
               "lf_conv2d_wgrad: in_scale/in_shift must both be set");
    const WgPlan pl = plan_wgrad(n, cin, cout, h, wd);
    const size_t count = (size_t)cin * ksize * ksize * cout;
    if (ws_bytes < (size_t)pl.slabs * count * sizeof(float)) {
        lf::set_error("lf_conv2d_wgrad: workspace %zu < %zu bytes", ws_bytes,
                      (size_t)pl.slabs * count * sizeof(float));
        return LF_ERR_WORKSPACE;
    }
    WgradArgs a;
    a.x = x; a.dy = dy; a.part = static_cast<float*>(workspace);
    a.in_scale = in_scale; a.in_shift = in_shift;
    a.n = n; a.cin = cin; a.cout = cout; a.h = h; a.wd = wd;
    a.tiles_x = pl.tiles_x; a.tiles_y = pl.tiles_y; a.items = pl.items;
    a.items_per_split = pl.items_per_split;
    a.in_relu = in_relu;
    dim3 grid(pl.splits, pl.gy, pl.gz);
    hipStream_t s = lf::as_stream(stream);
    const int rc = ksize == 3 ? launch_wgrad<9>(pl.variant, a, grid, s)
                              : launch_wgrad<1>(pl.variant, a, grid, s);
    if (rc != LF_OK) return rc;
    return lf::check_launch("lf_conv2d_wgrad");
}

int lf_conv2d_wgrad_reduce_f32(const void* workspace, float* dw, int n, int cin, int h, int wd,
                               int cout, int ksize, float beta, lf_stream_t stream) {
    LF_REQUIRE(workspace && dw, "lf_conv2d_wgrad_reduce: null buffer");
    LF_REQUIRE(n > 0 && cin > 0 && cout > 0 && h > 0 && wd > 0 && (ksize == 1 || ksize == 3),
               "lf_conv2d_wgrad_reduce: bad dims");
    const WgPlan pl = plan_wgrad(n, cin, cout, h, wd);
    const size_t count = (size_t)cin * ksize * ksize * cout;
    slab_reduce_kernel<<<lf::stream_grid(count, kThreads), kThreads, 0, lf::as_stream(stream)>>>(
        static_cast<const float*>(workspace), dw, count, pl.slabs, beta);
    return lf::check_launch("lf_conv2d_wgrad_reduce");
}

}  // extern "C"
