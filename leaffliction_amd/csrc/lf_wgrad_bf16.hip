// libleafhip — convolution weight gradient with bf16 operands and fp32 accumulation: the
// third of the three convolutions of a mixed-precision training step (the reference trains
// under Keras' mixed_float16 policy by default, srcs/cli/train.py:179-190; BASELINE
// configs[3] asks for the reduced-precision step in bf16).
//
//   dW[ci][tap][co] = sum over (n, y, x) of A[n][ci][y+dy][x+dx] * dY[n][co][y][x]
//
// as an implicit GEMM on v_mfma_f32_32x32x16_bf16 with M = input channels, N = output
// channels and K = PIXELS.  The operand of a lane is one channel and eight consecutive k, so
// with pixel-contiguous (NCHW) planes a tap shifted by one pixel would be a misaligned 16-byte
// LDS read (replayed at 64 cycles).  Instead the tile is kept in LDS TRANSPOSED — [pixel][32
// channels], one 64-byte row per pixel — and operands are fetched with ds_read_b64_tr_b16
// (four pixels x 16 channels per 16 lanes, delivered channel-per-lane): a tap is then a row
// offset, and every read is aligned.  The transposition costs nothing: staging computes in fp32
// anyway (the producer's BatchNorm+ReLU on A, the BatchNorm backward on dY) and the
// f32 -> bf16 packing simply pairs channels of one pixel instead of pixels of one channel.
// 8-byte channel quads are XOR-swizzled with the pixel index so that the stores of lanes that
// hold neighbouring pixel groups (coalesced global loads) spread over the banks.
//
// The step is HBM-bound, not MFMA-bound (bf16 tensors: 72 FLOP per byte moved at 224x224 /
// 32->32): the kernel reads A, the upstream gradient g and the BatchNorm input y once, writes
// dY once, and keeps two tiles ahead of the MFMAs: LDS is double-buffered (one barrier per tile)
// and the loads of the tile after next are in flight in registers.  A workgroup walks DOWN column
// strips of the images (round 3): the A patch in LDS is a ring of 2 * (TH + 2) pixel rows, a tile
// stages only its TH new rows next to the two it shares with the tile above (before, every tile
// staged TH + 2 rows and A crossed the fabric 1.5 times for the halo rows alone), and the strips
// of one image are walked at the same time by neighbouring workgroups of one XCD, so the 128-byte
// lines and halo columns two strips share are fetched once.  A 3x3 workgroup is 6 waves,
// wave (filter row, sub-block / K share): 48-96 accumulator registers per wave instead of 144-576,
// and at most two waves per SIMD, which is what leaves room for the loads in flight (48 registers
// of raw bf16 per thread with 16-byte accesses).
//
// Partial sums: every workgroup owns a (32*CIB x 32*COB) weight block and a range of (image,
// tile) items and writes one fp32 slab; lf_slab_reduce_f32 adds the slabs in a fixed order
// (deterministic, no float atomics), exactly like the fp32 path.
#include "lf_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int kSumT = 256;

using WgBf16Args = lf::WgradBf16Args;

__device__ __forceinline__ float up(unsigned bits16) { return __uint_as_float(bits16 << 16); }

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v;
    v.x = (__bf16)lo;  // round to nearest even
    v.y = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}

// byte offset of (pixel, channel quad) inside one [pixel][32 channels] image
__device__ __forceinline__ unsigned img_off(unsigned pixel, unsigned quad) {
    return pixel * 64u + ((quad ^ ((pixel >> 2) & 7u)) << 3);
}

template <int TAPS, int TW, int TH, int CIB, int COB, bool STEM, int WPRQ = 0>
struct WgShape {
    static constexpr int HALO = (TAPS == 9 && !STEM) ? 1 : 0;
    static constexpr int PW = TW + 2 * HALO, PH = TH + 2 * HALO;
    static constexpr int NSLOT = 2 * PH;               // ring rows of the A patch: this tile's PH and the next one's
    static constexpr int XPIX = PW * NSLOT, DPIX = TW * TH;
    static constexpr int XBYTES = CIB * XPIX * 64, DBYTES = COB * DPIX * 64;
    static constexpr int RED = TAPS * 16 * 64 * 4;    // the accumulators of one sub-block's waves
    static constexpr int LDS = XBYTES + 2 * DBYTES > RED ? XBYTES + 2 * DBYTES : RED;
    static constexpr int ROWS = TAPS == 9 ? 3 : 1;    // filter rows spread over waves
    // waves per filter row (they split the sub-blocks / K): 3x3 with one or two sub-blocks runs 6
    // waves (<= 2 per SIMD: 256 registers each), 2x2 sub-blocks 12 waves (one sub-block each)
    static constexpr int WPR = WPRQ ? WPRQ : (TAPS == 9 ? (CIB * COB == 4 ? 4 : 2) : 4);
    static constexpr int NT = 64 * WPR * ROWS;
};

// G = pixels per staging unit: 8 (16-byte global accesses: the vector memory pipe moves ~5.3 TB/s
// with 16-byte lanes, ~3 TB/s with 8-byte lanes, ~1.5 TB/s with 4-byte lanes —
// scripts/microbench/seg_bw.hip) when rows are 16-byte aligned (w % 8 == 0), else 4.
template <int TAPS, int TW, int TH, int CIB, int COB, bool STEM, int G, int WPRQ = 0>
__global__ __launch_bounds__((WgShape<TAPS, TW, TH, CIB, COB, STEM, WPRQ>::NT), 1) void wgrad_bf16_kernel(WgBf16Args p) {
    using S = WgShape<TAPS, TW, TH, CIB, COB, STEM, WPRQ>;
    static_assert((G == 4 || G == 8) && TW % G == 0, "staging groups of 4 or 8 pixels");
    typedef unsigned uvec __attribute__((ext_vector_type(G / 2)));   // G bf16
    constexpr int kT = S::NT, ROWS = S::ROWS, TPW = TAPS / ROWS;  // taps per wave
    static_assert(TW % 4 == 0 && (TW * TH) % 16 == 0, "tile: whole 4-pixel groups, whole 16-pixel k-steps");
    static_assert(!STEM || (TAPS == 1 && CIB == 1 && TW * TH == kT), "stem: one thread per pixel");
    constexpr int HALO = S::HALO, PW = S::PW, PH = S::PH;
    constexpr int PGS = TW / G;
    constexpr int NQ = CIB * COB, WPR = S::WPR;
    constexpr int SBW = NQ > WPR ? NQ / WPR : 1;   // sub-blocks per wave
    constexpr int KSPL = WPR > NQ ? WPR / NQ : 1;  // K shares per sub-block
    static_assert(NQ == 1 || NQ == 2 || NQ == 4, "wave decomposition");
    constexpr int NS = TW * TH / 16;  // k-steps per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ float lbn[5 * 32 * COB];
    __shared__ float lsc[2 * 32 * CIB];
    // per-(image, channel) factors of the gradient (the SE gate and its additive term) of the image being staged:
    // fetched from global memory where they are used they were a dependent round trip inside every commit
    __shared__ float lal[2 * 32 * COB];
    int lal_n = -1;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tr = wv % ROWS, wrest = wv / ROWS;  // filter row of this wave; its (sub-block, K share)
    const int q0 = NQ >= WPR ? wrest * SBW : wrest % NQ;  // first sub-block of this wave
    const int ks = NQ >= WPR ? 0 : wrest / NQ;
    const int ci0 = blockIdx.y * (32 * CIB), co0 = blockIdx.z * (32 * COB);
    const size_t hw = (size_t)p.h * p.w;
    const bool bn = p.bn_y != nullptr;
    const bool pro = p.in_scale != nullptr;

    if (bn)
        for (int e = tid; e < 5 * 32 * COB; e += kT) {
            const int kk = e / (32 * COB), c = e - kk * (32 * COB);
            lbn[e] = p.bn_coef[(size_t)kk * p.cout + co0 + c];
        }
    if (pro && !STEM)
        for (int c = tid; c < 32 * CIB; c += kT) {
            const bool in = ci0 + c < p.cin;
            lsc[c] = in ? p.in_scale[ci0 + c] : 1.f;
            lsc[32 * CIB + c] = in ? p.in_shift[ci0 + c] : 0.f;
        }

    f32x16 acc[SBW][TPW];
#pragma unroll
    for (int b = 0; b < SBW; ++b)
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][t][r] = 0.f;

    // ---- staging units: 4 channels x G pixels (one 2G-byte load per channel, 8-byte LDS stores per pixel)
    constexpr int NDU = COB * 8 * TH * PGS, DPT = (NDU + kT - 1) / kT;
    constexpr int NXU = STEM ? 0 : CIB * 8 * PH * PGS, XPT = (NXU + kT - 1) / kT;
    constexpr int NHU = STEM ? 0 : CIB * 8 * PH * 2 * HALO, HPT = (NHU + kT - 1) / kT;
    uvec rg[DPT][4], ry[DPT][4];
    uvec rx[XPT > 0 ? XPT : 1][4];
    // Halo columns, 2-byte loads: kept raw until they are used — combining channel pairs at issue puts the
    // wait for ALL of the tile's loads at issue — except in the 2x2 sub-block variants, which are at their
    // register budget (the two extra registers per unit spill: 64->64 @112 0.70 -> 0.80 ms)
    constexpr bool RAWH = CIB * COB < 4;
    unsigned rh[HPT > 0 ? HPT : 1][RAWH ? 4 : 2];
    float rs[STEM ? 27 : 1];
    unsigned dmask = 0, xmask = 0, hmask = 0;  // bit k: unit k lies inside the image

    // What a workgroup walks: segments of column strips (a strip = TW columns of one image, a segment = seg_tiles
    // consecutive tiles of it from top to bottom); see plan_wgrad_bf16 for how they are dealt out.
    const int SG = p.tiles_x, UI = SG * p.segs;
    const int xk = blockIdx.x & 7, xj = blockIdx.x >> 3, xw = gridDim.x >> 3;
    const int my_total = p.interleave ? (p.n > xk ? (p.n - xk + 7) / 8 : 0) * UI : p.n * UI;
    const int my_first = p.interleave ? xj : (int)blockIdx.x, my_step = p.interleave ? xw : (int)gridDim.x;
    const int my_units = my_total > my_first ? (my_total - my_first + my_step - 1) / my_step : 0;

    // a position in the workgroup's sequence of tiles: unit ui, tile tt of it
    struct Cursor {
        int ui, tt, n, tx0, t_first, t_count;
    };
    auto seek = [&](Cursor& c, int ui) {
        c.ui = ui;
        c.tt = 0;
        if (ui >= my_units) return;
        const int q = my_first + ui * my_step;
        const int im = q / UI, rem = q - im * UI, seg = rem / SG;
        c.n = p.interleave ? im * 8 + xk : im;
        c.tx0 = (rem - seg * SG) * TW;
        c.t_first = seg * p.seg_tiles;
        c.t_count = min(p.seg_tiles, p.tiles_y - c.t_first);
    };
    auto advance = [&](Cursor& c) {
        if (++c.tt >= c.t_count) seek(c, c.ui + 1);
    };
    auto valid = [&](const Cursor& c) { return c.ui < my_units; };

    // A tile's A patch is rows ty0 - HALO .. ty0 + TH + HALO - 1 of the strip.  The first tile of a segment stages
    // all PH of them, every other tile only the TH new ones (patch rows 2 * HALO .. PH - 1): the rest is in the ring.
    auto issue = [&](const Cursor& c) {
        const int n = c.n, tx0 = c.tx0, ty0 = (c.t_first + c.tt) * TH;
        const int pr0 = c.tt == 0 ? 0 : 2 * HALO;   // first patch row staged
        dmask = xmask = hmask = 0;
        const uint16_t* gn = p.g + (size_t)n * p.cout * hw;
        const uint16_t* yn = bn ? p.bn_y + (size_t)n * p.cout * hw : nullptr;
#pragma unroll
        for (int k = 0; k < DPT; ++k) {
            const int u = tid + k * kT;
            const int pg = u % PGS, t1 = u / PGS, quad = t1 % (8 * COB), row = t1 / (8 * COB);
            const int gy = ty0 + row, gx = tx0 + G * pg;
            const bool ok = u < NDU && gy < p.h && gx < p.w;
            dmask |= (ok ? 1u : 0u) << k;
            if (!ok) continue;
            const size_t o = (size_t)(co0 + 4 * quad) * hw + (size_t)gy * p.w + gx;
#pragma unroll
            for (int i = 0; i < 4; ++i) rg[k][i] = *reinterpret_cast<const uvec*>(gn + o + (size_t)i * hw);
            if (bn)
#pragma unroll
                for (int i = 0; i < 4; ++i) ry[k][i] = *reinterpret_cast<const uvec*>(yn + o + (size_t)i * hw);
        }
        if (STEM) {
            // one thread per pixel: its 3x3 neighbourhood in every input channel (fp32 input)
            const float* xn = reinterpret_cast<const float*>(p.x) + (size_t)n * p.cin * hw;
            const int row = tid / TW, col = tid - row * TW;
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int gy = ty0 + row + t / 3 - 1, gx = tx0 + col + t % 3 - 1;
                    const bool ok = c < p.cin && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
                    rs[c * 9 + t] = ok ? xn[(size_t)c * hw + (size_t)gy * p.w + gx] : 0.f;
                }
        } else {
            const uint16_t* xn = p.x + (size_t)n * p.cin * hw;
#pragma unroll
            for (int k = 0; k < XPT; ++k) {
                const int u = tid + k * kT;
                const int pg = u % PGS, t1 = u / PGS, quad = t1 % (8 * CIB), pr = pr0 + t1 / (8 * CIB);
                const int gy = ty0 - HALO + pr, gx = tx0 + G * pg;
                const bool ok = pr < PH && gy >= 0 && gy < p.h && gx < p.w && ci0 + 4 * quad < p.cin;
                xmask |= (ok ? 1u : 0u) << k;
                if (!ok) continue;
                const size_t o = (size_t)(ci0 + 4 * quad) * hw + (size_t)gy * p.w + gx;
#pragma unroll
                for (int i = 0; i < 4; ++i) rx[k][i] = *reinterpret_cast<const uvec*>(xn + o + (size_t)i * hw);
            }
#pragma unroll
            for (int k = 0; k < HPT; ++k) {
                const int u = tid + k * kT;
                const int side = u & 1, t1 = u >> 1, quad = t1 % (8 * CIB), pr = pr0 + t1 / (8 * CIB);
                const int gy = ty0 - HALO + pr, gx = side ? tx0 + TW : tx0 - 1;
                const bool ok = pr < PH && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w && ci0 + 4 * quad < p.cin;
                hmask |= (ok ? 1u : 0u) << k;
                if (!ok) continue;
                const size_t o = (size_t)(ci0 + 4 * quad) * hw + (size_t)gy * p.w + gx;
                if (RAWH) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) rh[k][i] = xn[o + (size_t)i * hw];
                } else {
                    rh[k][0] = (unsigned)xn[o] | (unsigned)xn[o + hw] << 16;
                    rh[k][1] = (unsigned)xn[o + 2 * hw] | (unsigned)xn[o + 3 * hw] << 16;
                }
            }
        }
    };

    // rb = ring slot of the tile's patch row 0; buf = which dY buffer
    auto commit = [&](const Cursor& c, int rb, int buf) {
        unsigned char* lx = lds;
        unsigned char* ld = lds + S::XBYTES + buf * S::DBYTES;
        const int n = c.n, tx0 = c.tx0, ty0 = (c.t_first + c.tt) * TH;
        const int pr0 = c.tt == 0 ? 0 : 2 * HALO;
        if (bn && p.bn_alpha != nullptr && n != lal_n) {   // uniform: a workgroup's items run through the images in order
            __syncthreads();   // nobody still reads the previous image's factors
            for (int c = tid; c < 32 * COB; c += kT) {
                const bool in = co0 + c < p.cout;
                lal[c] = in ? p.bn_alpha[(size_t)n * p.cout + co0 + c] : 1.f;
                lal[32 * COB + c] = in && p.bn_add ? p.bn_add[(size_t)n * p.cout + co0 + c] : 0.f;
            }
            __syncthreads();
            lal_n = n;
        }
        const bool gated = p.bn_alpha != nullptr;
        // ---- dY (optionally the BatchNorm backward of g), stored transposed; dy_out on the side
#pragma unroll
        for (int k = 0; k < DPT; ++k) {
            const int u = tid + k * kT;
            if (u >= NDU) continue;
            const int pg = u % PGS, t1 = u / PGS, quad = t1 % (8 * COB), row = t1 / (8 * COB);
            const bool ok = dmask >> k & 1u;
            // channel by channel (few live registers): dY of the channel's G pixels -> dy_out as it
            // stands; every second channel the pair (c-1, c) goes to LDS, one dword per pixel
            unsigned char* img = ld + (quad >> 3) * (S::DPIX * 64);
            const unsigned pd = (unsigned)(row * TW + G * pg);
            float prev[G];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = 4 * quad + i;  // channel inside the workgroup's block
                float gv[G];
#pragma unroll
                for (int e = 0; e < G; ++e) gv[e] = 0.f;
                if (ok) {
#pragma unroll
                    for (int e = 0; e < G; e += 2) {
                        gv[e] = up(rg[k][i][e / 2] & 0xffffu);
                        gv[e + 1] = up(rg[k][i][e / 2] >> 16);
                    }
                    if (bn) {
                        const float c0 = lbn[c], c1 = lbn[32 * COB + c], c2 = lbn[64 * COB + c],
                                    c3 = lbn[96 * COB + c], c4 = lbn[128 * COB + c];
                        const float al = gated ? lal[c] : 1.f, ad = gated ? lal[32 * COB + c] : 0.f;
#pragma unroll
                        for (int e = 0; e < G; ++e) {
                            const unsigned yw = ry[k][i][e / 2];
                            const float yv = up((e & 1) ? yw >> 16 : yw & 0xffffu);
                            float dz = fmaf(gv[e], al, ad);
                            if (p.bn_relu && !(fmaf(yv, c0, c1) > 0.f)) dz = 0.f;
                            gv[e] = fmaf(c2, dz, fmaf(c3, yv, c4));
                        }
                        if (p.dy_out != nullptr && blockIdx.y == 0) {
                            uvec o;
#pragma unroll
                            for (int e = 0; e < G; e += 2) o[e / 2] = pack2(gv[e], gv[e + 1]);
                            *reinterpret_cast<uvec*>(p.dy_out + ((size_t)n * p.cout + co0 + c) * hw +
                                                     (size_t)(ty0 + row) * p.w + tx0 + G * pg) = o;
                        }
                    }
                }
                if (i & 1) {
#pragma unroll
                    for (int e = 0; e < G; ++e)
                        *reinterpret_cast<unsigned*>(img + img_off(pd + e, quad & 7) + 4 * (i >> 1)) =
                            pack2(prev[e], gv[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < G; ++e) prev[e] = gv[e];
                }
            }
        }
        if (STEM) {
            // im2col row of this thread's pixel: "channel" = ci*9 + tap (27 used, 5 zero)
            const unsigned pd = (unsigned)(rb * PW + tid);   // PW = TW: ring slots rb .. rb + TH - 1
#pragma unroll
            for (int qd = 0; qd < 8; ++qd) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (4 * qd + i < 27) ? rs[(4 * qd + i < 27) ? 4 * qd + i : 0] : 0.f;
                u32x2 o;
                o.x = pack2(v[0], v[1]);
                o.y = pack2(v[2], v[3]);
                *reinterpret_cast<u32x2*>(lx + img_off(pd, qd)) = o;
            }
        } else {
#pragma unroll
            for (int k = 0; k < XPT; ++k) {
                const int u = tid + k * kT;
                const int pg = u % PGS, t1 = u / PGS, quad = t1 % (8 * CIB), pr = pr0 + t1 / (8 * CIB);
                if (pr >= PH) continue;
                const bool ok = xmask >> k & 1u;
                unsigned char* img = lx + (quad >> 3) * (S::XPIX * 64);
                int slot = rb + pr;
                slot = slot >= S::NSLOT ? slot - S::NSLOT : slot;
                const unsigned pi = (unsigned)(slot * PW + HALO + G * pg);
                float prev[G];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v[G];
#pragma unroll
                    for (int e = 0; e < G; ++e) v[e] = 0.f;  // zero padding stays zero
                    if (ok) {
#pragma unroll
                        for (int e = 0; e < G; e += 2) {
                            v[e] = up(rx[k][i][e / 2] & 0xffffu);
                            v[e + 1] = up(rx[k][i][e / 2] >> 16);
                        }
                        if (pro) {
                            const float sc = lsc[4 * quad + i], sh = lsc[32 * CIB + 4 * quad + i];
#pragma unroll
                            for (int e = 0; e < G; ++e) {
                                v[e] = fmaf(v[e], sc, sh);
                                if (p.in_relu) v[e] = fmaxf(v[e], 0.f);
                            }
                        }
                    }
                    if (i & 1) {
#pragma unroll
                        for (int e = 0; e < G; ++e)
                            *reinterpret_cast<unsigned*>(img + img_off(pi + e, quad & 7) + 4 * (i >> 1)) =
                                pack2(prev[e], v[e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < G; ++e) prev[e] = v[e];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < HPT; ++k) {
                const int u = tid + k * kT;
                const int side = u & 1, t1 = u >> 1, quad = t1 % (8 * CIB), pr = pr0 + t1 / (8 * CIB);
                if (pr >= PH) continue;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (hmask >> k & 1u) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        v[i] = RAWH ? up(rh[k][i]) : up(i & 1 ? rh[k][i >> 1] >> 16 : rh[k][i >> 1] & 0xffffu);
                    if (pro)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            v[i] = fmaf(v[i], lsc[4 * quad + i], lsc[32 * CIB + 4 * quad + i]);
                            if (p.in_relu) v[i] = fmaxf(v[i], 0.f);
                        }
                }
                u32x2 o;
                o.x = pack2(v[0], v[1]);
                o.y = pack2(v[2], v[3]);
                unsigned char* img = lx + (quad >> 3) * (S::XPIX * 64);
                int slot = rb + pr;
                slot = slot >= S::NSLOT ? slot - S::NSLOT : slot;
                *reinterpret_cast<u32x2*>(img + img_off((unsigned)(slot * PW + (side ? PW - 1 : 0)), quad & 7)) = o;
            }
        }
    };

    // ---- operand fetch: lane l of a 16-lane group supplies the address of pixel (l>>2)&3 of the
    // group's four, channel quad (l&3) of the group's 16 channels; it receives its own channel
    // (16*((l>>4)&1) + (l&15) = l&31) at the four pixels.  k = 8*(l>>5) + j: the two reads of a
    // fragment are pixel groups 2*(l>>5) and 2*(l>>5)+1 of the k-step's four.
    const unsigned rq = (unsigned)((lane >> 2) & 3), rquad = (unsigned)(4 * ((lane >> 4) & 1) + (lane & 3));
    const int kh = lane >> 5;
    auto frag = [&](const unsigned char* img, unsigned pix0, unsigned pix1) -> bf16x8 {
        const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(img + img_off(pix0 + rq, rquad)));
        const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(img + img_off(pix1 + rq, rquad)));
        s16x8 v;
        v.s0 = a.x; v.s1 = a.y; v.s2 = a.z; v.s3 = a.w;
        v.s4 = b.x; v.s5 = b.y; v.s6 = b.z; v.s7 = b.w;
        return __builtin_bit_cast(bf16x8, v);
    };
    auto compute = [&](int rb, int buf) {
        const unsigned char* xb = lds;
        const unsigned char* db = lds + S::XBYTES + buf * S::DBYTES;
#pragma unroll 1
        for (int s = ks; s < NS; s += KSPL) {
            const int f0 = 16 * s + 8 * kh, f1 = f0 + 4;  // flat tile positions of the lane's two groups
            const int r0 = f0 / TW, c0 = f0 - r0 * TW, r1 = f1 / TW, c1 = f1 - r1 * TW;
            // the wave's sub-blocks q0 .. q0+SBW-1 = (cib, cob) pairs, cib fastest: with SBW = 2 they
            // share cob (one B operand) when CIB = 2
            bf16x8 B[SBW];
#pragma unroll
            for (int b = 0; b < SBW; ++b) {
                const int cob = (q0 + b) / CIB;
                if (b == 0 || CIB == 1) B[b] = frag(db + cob * (S::DPIX * 64), (unsigned)f0, (unsigned)f1);
                else B[b] = B[0];
            }
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int dy = TAPS == 9 ? tr : 0, dx = t;
#pragma unroll
                for (int b = 0; b < SBW; ++b) {
                    const int cib = (q0 + b) % CIB;
                    int s0 = rb + r0 + dy, s1 = rb + r1 + dy;   // ring slots of the two pixel groups' rows
                    s0 = s0 >= S::NSLOT ? s0 - S::NSLOT : s0;
                    s1 = s1 >= S::NSLOT ? s1 - S::NSLOT : s1;
                    const bf16x8 A = frag(xb + cib * (S::XPIX * 64), (unsigned)(s0 * PW + c0 + dx),
                                          (unsigned)(s1 * PW + c1 + dx));
                    acc[b][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B[b], acc[b][t], 0, 0, 0);
                }
            }
        }
    };

    // Two tiles ahead: while tile i's MFMAs run (A from its ring slots, dY from one buffer), tile i+1 (loaded an
    // iteration earlier) is transformed into the ring slots behind them / the other dY buffer and the loads of
    // tile i+2 are issued; one barrier per tile.  A tile that continues its segment moves the ring base by TH rows
    // (its first 2 * HALO patch rows are the previous tile's last), one that starts a segment by PH rows.
    Cursor cc, cn, ci;   // computing, committing next, issuing
    seek(cc, 0);
    int rb = 0;
    if (valid(cc)) {
        issue(cc);
        __syncthreads();  // lbn / lsc are staged
        commit(cc, rb, 0);
        cn = cc;
        advance(cn);
        ci = cn;
        if (valid(cn)) {
            issue(cn);
            advance(ci);
        }
    }
    __syncthreads();
    int cur = 0;
    while (valid(cc)) {
        compute(rb, cur);
        int rb_next = rb;
        if (valid(cn)) {
            rb_next = rb + (cn.tt == 0 ? PH : TH);
            rb_next = rb_next >= S::NSLOT ? rb_next - S::NSLOT : rb_next;
            commit(cn, rb_next, cur ^ 1);
            if (valid(ci)) {
                issue(ci);
                advance(ci);
            }
        }
        __syncthreads();
        cc = cn;
        if (valid(cn)) advance(cn);
        rb = rb_next;
        cur ^= 1;
    }

    // K-split partner waves (only when a sub-block has several: SBW == 1) fold into k = 0 through
    // LDS, one (ci, co) sub-block per round (fixed order)
    float* red = reinterpret_cast<float*>(lds);
#pragma unroll 1
    for (int k = 1; k < KSPL; ++k) {
#pragma unroll 1
        for (int qq = 0; qq < NQ; ++qq) {
            __syncthreads();
            if (ks == k && q0 == qq) {
#pragma unroll
                for (int t = 0; t < TPW; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((tr * TPW + t) * 16 + r) * 64 + lane] = acc[0][t][r];
            }
            __syncthreads();
            if (ks == 0 && q0 == qq) {
#pragma unroll
                for (int t = 0; t < TPW; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[0][t][r] += red[((tr * TPW + t) * 16 + r) * 64 + lane];
            }
        }
    }
    if (ks == 0) {
        // D[row = A channel][col = dY channel]: register r of a lane is row (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
        for (int b = 0; b < SBW; ++b) {
            const int cib = (q0 + b) % CIB, cob = (q0 + b) / CIB;
            const int co = co0 + cob * 32 + (lane & 31);
            if (STEM) {
                float* out = p.part + (size_t)blockIdx.x * p.cin * 9 * p.cout;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kidx = (r & 3) + 8 * (r >> 2) + 4 * kh;  // = ci*9 + tap
                    if (kidx < p.cin * 9 && co < p.cout) out[(size_t)kidx * p.cout + co] = acc[b][0][r];
                }
            } else {
                float* out = p.part + (size_t)blockIdx.x * p.cin * TAPS * p.cout;
#pragma unroll
                for (int t = 0; t < TPW; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ci = ci0 + cib * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                        if (ci < p.cin && co < p.cout)
                            out[((size_t)ci * TAPS + tr * TPW + t) * p.cout + co] = acc[b][t][r];
                    }
            }
        }
    }
}

// dst[g][i] = sum over slabs s in group g of part[s][i]; with one group this is the final
// dw = beta*dw + sum.  Fixed order -> deterministic.
__global__ __launch_bounds__(kSumT) void slab_sum_kernel(const float* __restrict__ part, float* __restrict__ dst,
                                                      size_t count, int nslabs, int per_group, float beta,
                                                      int final_pass) {
    const int g = blockIdx.y;
    const int s0 = g * per_group, s1 = min(s0 + per_group, nslabs);
    for (size_t i = (size_t)blockIdx.x * kSumT + threadIdx.x; i < count; i += (size_t)gridDim.x * kSumT) {
        float s = 0.f;
        for (int k = s0; k < s1; ++k) s += part[(size_t)k * count + i];
        float* o = dst + (size_t)g * count + i;
        *o = (final_pass && beta != 0.f) ? fmaf(beta, *o, s) : s;
    }
}

struct WgBf16Plan {
    int variant;  // index into the launch table
    int tw, th, cib, cob, stem;
    int tiles_x, tiles_y, seg_tiles, segs, splits, interleave, gy, gz;
};

// tile shapes: 32x8 (any width, partial tiles masked), 56x4 (112 and 56 wide), 28x4 (28 wide)
WgBf16Plan plan_wgrad_bf16(int n, int cin, int cout, int h, int w, int ksize) {
    WgBf16Plan pl{};
    pl.stem = (ksize == 3 && cin * 9 <= 32) ? 1 : 0;
    if (pl.stem) {
        pl.tw = 32; pl.th = 8; pl.cib = 1; pl.cob = 1;
    } else {
        if (w % 56 == 0) { pl.tw = 56; pl.th = 4; }
        else if (w % 28 == 0 && w % 32 != 0) { pl.tw = 28; pl.th = 4; }
        else { pl.tw = 32; pl.th = 8; }
        pl.cib = cin > 32 ? 2 : 1;
        pl.cob = cout % 64 == 0 ? 2 : 1;
        if (pl.cib == 2 && pl.cob == 1) pl.cib = 1;   // (2,1) is not instantiated
    }
    pl.tiles_x = (w + pl.tw - 1) / pl.tw;
    pl.tiles_y = (h + pl.th - 1) / pl.th;
    pl.gy = pl.stem ? 1 : (cin + 32 * pl.cib - 1) / (32 * pl.cib);
    pl.gz = cout / (32 * pl.cob);
    // one resident 12-wave workgroup per CU (3x3), a few 4-wave ones (1x1): two rounds' worth of workgroups, each
    // with a slab of partial sums of its own.  They walk column strips; strips are cut into segments (which re-stage
    // the two rows above them) only when there are too few to go round — small batches.
    int want = (256 * 2) / (pl.gy * pl.gz);
    if (want < 1) want = 1;
    const int strips = n * pl.tiles_x;
    int segs = (want + strips - 1) / strips;
    const int max_segs = (pl.tiles_y + 3) / 4;        // at least four tiles to a segment
    if (segs > max_segs) segs = max_segs;
    if (segs < 1) segs = 1;
    pl.seg_tiles = (pl.tiles_y + segs - 1) / segs;
    pl.segs = (pl.tiles_y + pl.seg_tiles - 1) / pl.seg_tiles;
    const int units = strips * pl.segs;
    pl.splits = want < units ? want : units;
    // an image's strips side by side on one XCD (see the kernel) when the grid's x extent divides over the XCDs
    pl.interleave = (pl.splits % 8 == 0 && n >= 8 && pl.gy * pl.gz == 1) ? 1 : 0;
    return pl;
}

constexpr int kSumGroup = 32;

template <int TAPS, int TW, int TH, int CIB, int COB, bool STEM, int G, int WPRQ = 0>
int launch_wg(const WgBf16Args& a, dim3 grid, hipStream_t s) {
    using S = WgShape<TAPS, TW, TH, CIB, COB, STEM, WPRQ>;
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16_kernel<TAPS, TW, TH, CIB, COB, STEM, G, WPRQ>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, S::LDS) != hipSuccess) {
            lf::set_error("lf_conv2d_wgrad_bf16: cannot reserve %d bytes of LDS", S::LDS);
            return LF_ERR_LAUNCH;
        }
        raised = true;
    }
    wgrad_bf16_kernel<TAPS, TW, TH, CIB, COB, STEM, G, WPRQ><<<grid, S::NT, S::LDS, s>>>(a);
    return LF_OK;
}

template <int TAPS>
int dispatch_wg(const WgBf16Plan& pl, const WgBf16Args& a, dim3 grid, hipStream_t s) {
    const bool wide = a.w % 8 == 0;  // rows 16-byte aligned: 8-pixel staging groups
#define LF_WG(TW_, TH_, CIB_, COB_, G_) \
    if (pl.tw == TW_ && pl.th == TH_ && pl.cib == CIB_ && pl.cob == COB_ && wide == (G_ == 8)) \
        return launch_wg<TAPS, TW_, TH_, CIB_, COB_, false, G_>(a, grid, s)
    LF_WG(32, 8, 1, 1, 8);
    LF_WG(32, 8, 1, 2, 8);
    LF_WG(32, 8, 2, 2, 8);
    LF_WG(32, 8, 1, 1, 4);
    LF_WG(32, 8, 1, 2, 4);
    LF_WG(32, 8, 2, 2, 4);
    LF_WG(56, 4, 1, 1, 8);
    LF_WG(56, 4, 1, 2, 8);
    LF_WG(56, 4, 2, 2, 8);
    LF_WG(28, 4, 1, 1, 4);
    LF_WG(28, 4, 1, 2, 4);
    LF_WG(28, 4, 2, 2, 4);
#undef LF_WG
    lf::set_error("lf_conv2d_wgrad_bf16: no kernel for tile %dx%d blocks %dx%d", pl.tw, pl.th, pl.cib, pl.cob);
    return LF_ERR_INVALID;
}

}  // namespace

extern "C" {

size_t lf_conv2d_wgrad_bf16_workspace(int n, int cin, int h, int w, int cout, int ksize) {
    if (n <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || cout % 32 != 0) return 0;
    const WgBf16Plan pl = plan_wgrad_bf16(n, cin, cout, h, w, ksize);
    const size_t count = (size_t)cin * ksize * ksize * cout;
    const size_t groups = (pl.splits + kSumGroup - 1) / kSumGroup;
    return ((size_t)pl.splits + groups) * count * sizeof(float);
}

int lf_conv2d_wgrad_bf16(const void* x, const uint16_t* g, const uint16_t* bn_y, const float* alpha_nc,
                         const float* add_nc, const float* coef, int bn_relu, uint16_t* dy_out, float* dw,
                         int n, int cin, int h, int w, int cout, int ksize, const float* in_scale,
                         const float* in_shift, int in_relu, void* workspace, size_t ws_bytes,
                         lf_stream_t stream) {
    LF_REQUIRE(x && g && dw && workspace, "lf_conv2d_wgrad_bf16: null buffer");
    LF_REQUIRE(n > 0 && cin > 0 && cout > 0 && h > 0 && w > 0, "lf_conv2d_wgrad_bf16: bad dims");
    LF_REQUIRE(ksize == 3 || ksize == 1, "lf_conv2d_wgrad_bf16: ksize must be 1 or 3 (got %d)", ksize);
    LF_REQUIRE(w % 4 == 0, "lf_conv2d_wgrad_bf16: width must be a multiple of 4 (got %d)", w);
    LF_REQUIRE(cout % 32 == 0, "lf_conv2d_wgrad_bf16: cout must be a multiple of 32 (got %d)", cout);
    LF_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "lf_conv2d_wgrad_bf16: scale/shift must both be set");
    LF_REQUIRE(bn_y != nullptr || (alpha_nc == nullptr && add_nc == nullptr && dy_out == nullptr),
               "lf_conv2d_wgrad_bf16: alpha / add / dy_out need bn_y");
    LF_REQUIRE(bn_y == nullptr || coef != nullptr, "lf_conv2d_wgrad_bf16: bn_y needs coef");
    LF_REQUIRE(add_nc == nullptr || alpha_nc != nullptr, "lf_conv2d_wgrad_bf16: add needs alpha");
    const WgBf16Plan pl = plan_wgrad_bf16(n, cin, cout, h, w, ksize);
    LF_REQUIRE(pl.stem || cin % 4 == 0, "lf_conv2d_wgrad_bf16: cin must be a multiple of 4 (got %d)", cin);
    LF_REQUIRE(!pl.stem || (in_scale == nullptr), "lf_conv2d_wgrad_bf16: the small-Cin path takes no prologue");
    LF_REQUIRE(((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(g) | reinterpret_cast<size_t>(bn_y) |
                 reinterpret_cast<size_t>(dy_out)) & 15) == 0,
               "lf_conv2d_wgrad_bf16: tensors must be 16-byte aligned");
    if (ws_bytes < lf_conv2d_wgrad_bf16_workspace(n, cin, h, w, cout, ksize)) {
        lf::set_error("lf_conv2d_wgrad_bf16: workspace %zu < %zu bytes", ws_bytes,
                      lf_conv2d_wgrad_bf16_workspace(n, cin, h, w, cout, ksize));
        return LF_ERR_WORKSPACE;
    }
    WgBf16Args a{};
    a.x = static_cast<const uint16_t*>(x); a.g = g; a.part = static_cast<float*>(workspace);
    a.in_scale = in_scale; a.in_shift = in_shift; a.in_relu = in_relu;
    a.n = n; a.cin = cin; a.cout = cout; a.h = h; a.w = w;
    a.tiles_x = pl.tiles_x; a.tiles_y = pl.tiles_y; a.seg_tiles = pl.seg_tiles; a.segs = pl.segs;
    a.interleave = pl.interleave;
    a.bn_y = bn_y; a.bn_alpha = alpha_nc; a.bn_add = add_nc; a.bn_coef = coef; a.dy_out = dy_out;
    a.bn_relu = bn_relu;
    dim3 grid(pl.splits, pl.gy, pl.gz);
    hipStream_t s = lf::as_stream(stream);
    int rc;
    // one 32-channel block each way (32->32 at 224x224, the largest tensors): measured on one MI355X
    // box — 12 waves with 8-byte staging groups 1.35 ms, 6 waves with 16-byte groups 1.52 ms, a
    // producer/consumer split with two tiles of loads in flight 1.57 ms, 12 waves with 16-byte groups
    // (register spills) 2.52 ms: this shape wants waves more than it wants wide loads
    if (!pl.stem && ksize == 3 && pl.cib == 1 && pl.cob == 1 && pl.tw == 56)
        rc = launch_wg<9, 56, 4, 1, 1, false, 4, 4>(a, grid, s);
    else if (pl.stem)
        rc = a.w % 8 == 0 ? launch_wg<1, 32, 8, 1, 1, true, 8>(a, grid, s)
                          : launch_wg<1, 32, 8, 1, 1, true, 4>(a, grid, s);
    else
        rc = ksize == 3 ? dispatch_wg<9>(pl, a, grid, s) : dispatch_wg<1>(pl, a, grid, s);
    if (rc != LF_OK) return rc;
    // slabs -> dw, fixed order
    const size_t count = (size_t)cin * ksize * ksize * cout;
    float* part = static_cast<float*>(workspace);
    const unsigned gx = lf::stream_grid(count, kSumT, 1024);
    if (pl.splits <= kSumGroup) {
        slab_sum_kernel<<<dim3(gx, 1), kSumT, 0, s>>>(part, dw, count, pl.splits, pl.splits, 0.f, 1);
    } else {
        const int groups = (pl.splits + kSumGroup - 1) / kSumGroup;
        float* stage = part + (size_t)pl.splits * count;
        slab_sum_kernel<<<dim3(gx, groups), kSumT, 0, s>>>(part, stage, count, pl.splits, kSumGroup, 0.f, 0);
        slab_sum_kernel<<<dim3(gx, 1), kSumT, 0, s>>>(stage, dw, count, groups, groups, 0.f, 1);
    }
    return lf::check_launch("lf_conv2d_wgrad_bf16");
}

}  // extern "C"
