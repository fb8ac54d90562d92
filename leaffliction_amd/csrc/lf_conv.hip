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
//   slice for KC input channels are staged in LDS per K-chunk.  Staging uses 16-byte
//   loads (row interiors and weight rows) with every load of a chunk in flight at once, and
//   the NEXT chunk's loads are issued into registers before the current chunk's MFMAs
//   (a bounded register prefetch: <= 13 float4 per thread); 2-4 workgroups are resident per
//   CU so one workgroup's LDS write phase overlaps another's MFMAs.
//   An optional prologue applies y = relu(x*scale[c]+shift[c]) to the input while staging
//   (BatchNorm+ReLU of the producer fused into the consumer; zero padding stays zero).
//
// wgrad kernel (wgrad_mfma_kernel): dW[ci][tap][co] = sum_pixels X[ci][p+tap] dY[co][p],
//   K = pixels, split across workgroups and across the waves of a workgroup (reduced through
//   LDS); every workgroup writes one partial slab and a two-stage reduce sums the slabs in a
//   fixed order (deterministic, no float atomics).  wgrad_smallcin_kernel packs (ci, tap)
//   into the MFMA's M dimension for the stem (Cin*9 <= 32): one MFMA per pixel pair.
#include <type_traits>

#include "lf_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
#ifndef LF_KC_SMALL
#define LF_KC_SMALL 8
#endif

__device__ __forceinline__ float pro_apply(float v, float sc, float sh, int relu) {
    v = fmaf(v, sc, sh);
    return relu ? fmaxf(v, 0.f) : v;
}

// Raw buffer loads: SGPR resource (base, byte size) + a 32-bit byte offset per lane; an offset
// at or beyond the size returns 0, which is how zero padding is fetched (no branches, no
// 64-bit address arithmetic per lane).
typedef float lf_f4 __attribute__((ext_vector_type(4)));
constexpr unsigned kBufOob = 0xffffffffu;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const lf_f4 v = __builtin_bit_cast(lf_f4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

struct ConvArgs {
    const float* x;
    const float* w;  // [Cin][TAPS][Cout]
    float* y;
    const float* in_scale;  // optional prologue (nullptr = none)
    const float* in_shift;
    int n, cin, cout, h, wd;
    int tiles_x, tiles_y;
    int in_relu;
    int accumulate;  // y += conv(...) instead of y = conv(...)
    int vec_ok;      // W % 4 == 0, Cout % 4 == 0, 16-byte aligned bases
    // images per workgroup column (1, or 2 for the STK instantiation): with H = 28 and 8-row
    // tiles two images are walked as one 56-row strip, so no tile is half empty; the tile that
    // straddles the seam carries each image's own halo rows
    int stack;
    // optional BatchNorm statistics of the output, gathered in the epilogue: per (channel, tile)
    // sum and sum of squares of (y - pivot[co]) -> stat_part[(co * stat_tiles + tile) * 2 + {0,1}]
    float* stat_part;
    const float* stat_pivot;  // may be null (pivot 0)
    long long stat_tiles;     // n * tiles_x * tiles_y
    // with stat_mask_y (same shape as y) the epilogue gathers BatchNorm-BACKWARD sums instead:
    // d = y_out * [stat_mask_y*mask_scale[co]+mask_shift[co] > 0 or !mask_relu];
    // stat_part gets {sum d, sum d*stat_mask_y}
    const float* stat_mask_y;
    const float* mask_scale;
    const float* mask_shift;
    int mask_relu;
};

// sum over the 32 lanes of each wave half (DPP); the total lands in lanes 16-31 / 48-63
__device__ __forceinline__ float dpp_add(float v, float moved) { return v + moved; }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float half_sum32(float v) {
    v += dpp_move<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v += dpp_move<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v += dpp_move<0x141, 0xf>(v);  // row_half_mirror
    v += dpp_move<0x140, 0xf>(v);  // row_mirror: every lane of a 16-lane row holds the row sum
    v += dpp_move<0x142, 0xa>(v);  // row_bcast15 into rows 1 and 3
    return v;
}

// ---------------------------------------------------------------------------
// forward / dgrad
// ---------------------------------------------------------------------------
// TAPS: 9 (3x3) or 1 (1x1).  Tile TW x TH pixels = NPB blocks of 32 (flat index).
// Waves: WCO x WPX = 4; each wave computes MB cout-blocks x NB pixel-blocks.
// min waves/SIMD asked of the register allocator: accumulators + VGPRs share one 512-entry
// file per SIMD lane; with the prefetch registers <=32 accumulators fit 3 waves, more fit 2.
template <int TAPS, int TW, int TH, int WCO, int MB, int WPX, int NB, int kKC, bool STK = false>
__global__ __launch_bounds__(kThreads, (MB * NB * 16 <= 32 ? 3 : 2))
void conv_mfma_kernel(ConvArgs p) {
    constexpr int NPB = TW * TH / 32;
    static_assert(TW * TH % 32 == 0 && TW % 4 == 0, "tile must be whole 32-pixel blocks, TW % 4 == 0");
    static_assert(WCO * WPX == 4 && WPX * NB == NPB, "wave decomposition");
    constexpr int CT = 32 * WCO * MB;
    constexpr int HALO = TAPS == 9 ? 1 : 0;
    constexpr int PW = TW + 2 * HALO, PH = TH + 2 * HALO + (STK ? 2 * HALO : 0), PP = PW * PH;
    constexpr int PATCH = kKC * PP;
    constexpr int WSZ = kKC * TAPS * CT;
    // vector staging: per patch row TW/4 float4 interior items + 2 halo scalars, kept in two
    // homogeneous item arrays (mixing both kinds in one array makes the compiler wait for
    // every load right where it is issued)
    constexpr int TW4 = TW / 4;
    constexpr int NVI = kKC * PH * TW4, IPT = (NVI + kThreads - 1) / kThreads;
    constexpr int NHI = kKC * PH * 2 * HALO, HPT = (NHI + kThreads - 1) / kThreads;
    constexpr int CT4 = CT / 4, NWI = kKC * TAPS * CT4, WPT = (NWI + kThreads - 1) / kThreads;
    constexpr int kMaxProC = 512;  // prologue scale/shift staged in LDS for up to this many channels

    __shared__ __attribute__((aligned(16))) float lds[PATCH + WSZ];
    __shared__ float lsc[2 * kMaxProC];
    float* lp = lds;
    float* lw = lds + PATCH;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wave_co = wid % WCO, wave_px = wid / WCO;
    // XCD-aware order: workgroups are dealt to the eight XCDs round-robin in dispatch order (x, then
    // y, then z), each XCD with its own L2.  XCD k takes the k-th contiguous share of the
    // (tile, channel group, strip) space, so the tiles that share halo rows meet in one L2.
    const unsigned gxy = gridDim.x * gridDim.y, gtotal = gxy * gridDim.z;
    const unsigned bflat = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned xk = bflat & 7u, xfloor = gtotal >> 3, xrem = gtotal & 7u;
    const unsigned wid_flat = xk * xfloor + (xk < xrem ? xk : xrem) + (bflat >> 3);
    const int bz = (int)(wid_flat / gxy);
    const int by = (int)((wid_flat - (unsigned)bz * gxy) / gridDim.x);
    const int tile = (int)(wid_flat - (unsigned)bz * gxy - (unsigned)by * gridDim.x);
    const int tx0 = (tile % p.tiles_x) * TW, ty0 = (tile / p.tiles_x) * TH;
    const int co0 = by * CT;
    const int n = bz * p.stack;  // first image of this workgroup's strip
    const size_t hw = (size_t)p.h * p.wd;
    const unsigned uhw = (unsigned)hw;
    const float* xin = p.x + (size_t)n * p.cin * hw;
    // strip rows ty0 .. ty0+TH-1: the first `ra` belong to image n + imgA (rows gyA0 ..), the
    // rest (only when the tile straddles a seam, or hangs over the bottom) to the next image
    const int imgA = ty0 / p.h, gyA0 = ty0 - imgA * p.h;
    const int ra = min(TH, p.h - gyA0);
    const bool imgA_ok = imgA < p.stack && n + imgA < p.n;
    const bool imgB_ok = STK && imgA + 1 < p.stack && n + imgA + 1 < p.n;
    // patch row -> (image, row): [0, ra+2H) image A from gyA0-H; then image B from -H
    auto patch_row = [&](int py, int& img, int& gy) {
        if (py < ra + 2 * HALO) {
            img = imgA;
            gy = gyA0 - HALO + py;
            return imgA_ok && gy >= 0 && gy < p.h;
        }
        img = imgA + 1;
        gy = py - (ra + 2 * HALO) - HALO;
        return imgB_ok && ra < TH && gy >= 0 && gy < p.h && gy < TH - ra + HALO;
    };

    // per-lane LDS read bases
    const int khalf = lane >> 5, j = lane & 31;
    int bbase[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int f = (wave_px * NB + nb) * 32 + j;
        const int r = f / TW, prow = (STK && r >= ra) ? r + 2 * HALO : r;
        bbase[nb] = khalf * PP + prow * PW + (f % TW);
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

    // cp loop only partially unrolled: the scheduler otherwise hoists dozens of LDS reads and
    // the accumulators + prefetch registers no longer fit
    auto compute_chunk = [&]() {
#pragma unroll 2
        for (int cp = 0; cp < kKC / 2; ++cp) {
            const float* lwc = lw + abase + 2 * cp * TAPS * CT;
            const float* lpc = lp + 2 * cp * PP;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int dy = TAPS == 9 ? tap / 3 : 0, dx = TAPS == 9 ? tap % 3 : 0;
                float a[MB], b[NB];
#pragma unroll
                for (int m = 0; m < MB; ++m) a[m] = lwc[tap * CT + m * 32];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) b[nb] = lpc[bbase[nb] + dy * PW + dx];
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[m][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[nb], acc[m][nb], 0,
                                                                          0, 0);
            }
        }
    };

    const int nchunks = (p.cin + kKC - 1) / kKC;
    const bool vec = p.vec_ok && (tx0 + TW <= p.wd);  // uniform per workgroup

    if (vec) {
        // ---------------- vector staging with register prefetch ----------------
        float4 pv[IPT];
        float ph[HPT > 0 ? HPT : 1];
        float4 wv[WPT];
        unsigned okmask = 0;  // bit i: interior item i in-image; bit 16+i: halo item i in-image
        // variants with > 64 accumulators have no registers left to hold the weight prefetch:
        // they prefetch the patch only and fetch the (L2-resident) weights in the store phase
        constexpr bool kPrefetchW = MB * NB * 16 <= 64;
        if (pro) {  // BatchNorm scale/shift of the producer: staged once per workgroup
            for (int c = tid; c < p.cin && c < kMaxProC; c += kThreads) {
                lsc[c] = p.in_scale[c];
                lsc[kMaxProC + c] = p.in_shift[c];
            }
        }
        // Per-thread staging invariants (the workgroup has one tile; a chunk only moves the
        // channel base): byte offsets from the chunk's first channel plane / weight row with
        // kBufOob for everything that must read as zero (rows and columns outside the image,
        // slots beyond the item count, cout columns beyond the tensor); LDS index | kc << 16.
        // Channels / weight rows beyond Cin fall outside the chunk's buffer size -> zeros.
        unsigned pg[IPT], pl[IPT];
        unsigned hg[HPT > 0 ? HPT : 1], hl[HPT > 0 ? HPT : 1];
        unsigned wg[WPT];
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const int e = tid + i * kThreads;
            const int kc = e / (PH * TW4), rem = e - kc * (PH * TW4);
            const int py = rem / TW4, slot = rem - py * TW4;
            int img, gy;
            const bool ok = patch_row(py, img, gy) && e < NVI;
            pg[i] = ok ? 4u * ((unsigned)(img * p.cin + kc) * uhw + (unsigned)gy * (unsigned)p.wd + (unsigned)(tx0 + 4 * slot)) : kBufOob;
            pl[i] = (unsigned)(kc * PP + py * PW + HALO + 4 * slot) | ((unsigned)kc << 16);
            okmask |= (ok ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < HPT; ++i) {
            const int e = tid + i * kThreads;
            const int kc = e / (PH * 2), rem = e - kc * (PH * 2);
            const int py = rem >> 1, side = rem & 1;
            const int gx = side ? tx0 + TW : tx0 - 1;
            int img, gy;
            const bool ok = patch_row(py, img, gy) && e < NHI && gx >= 0 && gx < p.wd;
            hg[i] = ok ? 4u * ((unsigned)(img * p.cin + kc) * uhw + (unsigned)gy * (unsigned)p.wd + (unsigned)gx) : kBufOob;
            hl[i] = (unsigned)(kc * PP + py * PW + (side ? PW - 1 : 0)) | ((unsigned)kc << 16);
            okmask |= (ok ? 1u : 0u) << (16 + i);
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int e = tid + i * kThreads;
            const int row = e / CT4, col = (e - row * CT4) * 4;
            wg[i] = (e < NWI && co0 + col < p.cout) ? 4u * ((unsigned)row * (unsigned)p.cout + (unsigned)(co0 + col)) : kBufOob;
        }
        auto load_patch = [&](int c0) {
            // (with a strip of two images the second image's chunk lies Cin planes further on;
            // the host only stacks when Cin is a whole number of chunks)
            const __amdgpu_buffer_rsrc_t rx = buf_rsrc(
                xin + (size_t)c0 * hw,
                4u * (unsigned)((STK ? (p.stack - 1) * p.cin : 0) + min(kKC, p.cin - c0)) * uhw);
#pragma unroll
            for (int i = 0; i < IPT; ++i) pv[i] = buf_load4(rx, pg[i]);
#pragma unroll
            for (int i = 0; i < HPT; ++i) ph[i] = buf_load1(rx, hg[i]);
        };
        auto load_weights = [&](int c0) {
            const __amdgpu_buffer_rsrc_t rw =
                buf_rsrc(p.w + (size_t)c0 * TAPS * p.cout,
                         4u * (unsigned)(min(kKC, p.cin - c0) * TAPS) * (unsigned)p.cout);
#pragma unroll
            for (int i = 0; i < WPT; ++i) wv[i] = buf_load4(rw, wg[i]);
        };
        auto pro_sc = [&](int c) { return c < kMaxProC ? lsc[c] : p.in_scale[c]; };
        auto pro_sh = [&](int c) { return c < kMaxProC ? lsc[kMaxProC + c] : p.in_shift[c]; };
        auto store_chunk = [&](int c0) {
            if (!kPrefetchW) load_weights(c0);
#pragma unroll
            for (int i = 0; i < IPT; ++i) {
                if (tid + i * kThreads < NVI) {
                    const int kc = (int)(pl[i] >> 16);
                    float4 v = pv[i];
                    if (pro && (okmask >> i & 1u)) {
                        const float sc = pro_sc(c0 + kc), sh = pro_sh(c0 + kc);
                        v.x = pro_apply(v.x, sc, sh, p.in_relu);
                        v.y = pro_apply(v.y, sc, sh, p.in_relu);
                        v.z = pro_apply(v.z, sc, sh, p.in_relu);
                        v.w = pro_apply(v.w, sc, sh, p.in_relu);
                    }
                    float* dst = lp + (pl[i] & 0xffffu);
                    {
                        dst[0] = v.x;
                        dst[1] = v.y;
                        dst[2] = v.z;
                        dst[3] = v.w;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < HPT; ++i) {
                if (tid + i * kThreads < NHI) {
                    const int kc = (int)(hl[i] >> 16);
                    float v = ph[i];
                    if (pro && (okmask >> (16 + i) & 1u))
                        v = pro_apply(v, pro_sc(c0 + kc), pro_sh(c0 + kc), p.in_relu);
                    lp[hl[i] & 0xffffu] = v;
                }
            }
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                const int e = tid + i * kThreads;
                if (e < NWI) reinterpret_cast<float4*>(lw)[e] = wv[i];  // lw[row*CT + col]
            }
        };
        load_patch(0);
        if (kPrefetchW) load_weights(0);
        for (int ch = 0; ch < nchunks; ++ch) {
            __syncthreads();  // previous chunk's LDS reads are done
            store_chunk(ch * kKC);
            __syncthreads();
            if (ch + 1 < nchunks) {  // in flight during the MFMAs
                load_patch((ch + 1) * kKC);
                if (kPrefetchW) load_weights((ch + 1) * kKC);
            }
            compute_chunk();
        }
    } else {
        // ---------------- scalar staging (ragged shapes / partial tiles) ----------------
        // a thread owns one patch position (two when the patch has more than 256) and walks
        // the chunk's channels: every load is base + c*H*W
        constexpr int SLOTS = PP <= 64 ? 64 : (PP <= 128 ? 128 : 256);
        constexpr int G = kThreads / SLOTS;
        constexpr int ROUNDS = (PP + SLOTS - 1) / SLOTS;
        const int slot = tid % SLOTS, grp = tid / SLOTS;
        constexpr int WROWS = kKC * TAPS, RPP = kThreads / CT;
        const int wcol = tid % CT, wrow0 = tid / CT;
        const bool wcol_ok = co0 + wcol < p.cout;
        for (int ch = 0; ch < nchunks; ++ch) {
            const int c0 = ch * kKC;
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const int pos = slot + r * SLOTS;
                if (pos < PP) {
                    const int py = pos / PW, px = pos - py * PW;
                    const int gy = ty0 + py - HALO, gx = tx0 + px - HALO;
                    const bool inb = gy >= 0 && gy < p.h && gx >= 0 && gx < p.wd;
                    const unsigned goff = inb ? (unsigned)gy * (unsigned)p.wd + (unsigned)gx : 0u;
#pragma unroll 4
                    for (int kc = grp; kc < kKC; kc += G) {
                        const int c = c0 + kc;
                        float v = 0.f;
                        if (inb && c < p.cin) {
                            v = xin[(unsigned)c * uhw + goff];
                            if (pro) v = pro_apply(v, p.in_scale[c], p.in_shift[c], p.in_relu);
                        }
                        lp[kc * PP + pos] = v;
                    }
                }
            }
            const int wvalid = (p.cin - c0) * TAPS;
#pragma unroll 4
            for (int row = wrow0; row < WROWS; row += RPP) {
                float v = 0.f;
                if (wcol_ok && row < wvalid)
                    v = p.w[((unsigned)c0 * TAPS + row) * (unsigned)p.cout + (unsigned)(co0 + wcol)];
                lw[row * CT + wcol] = v;
            }
            __syncthreads();
            compute_chunk();
        }
    }

    // epilogue: D[row = co][col = pixel]; row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // Processed in groups of RG accumulator rows: the group's read-modify-write operands
    // (accumulate) and BatchNorm-backward mask values are loaded unconditionally from clamped
    // addresses first, so RG*NB loads are in flight together, then stored / reduced.
    constexpr int RG = TAPS == 1 ? (NB <= 2 ? 4 : 1) : (NB <= 2 ? 16 : 4);
    float* yout = p.y + (size_t)n * p.cout * hw;
    const bool stats = p.stat_part != nullptr, masked = p.stat_mask_y != nullptr;
    bool pix_ok[NB];
    unsigned pixc[NB];  // pixel offset, 0 when outside the image
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int f = (wave_px * NB + nb) * 32 + j;
        const int r = f / TW, ox = tx0 + f % TW;
        const bool in_a = r < ra;
        const int oy = in_a ? gyA0 + r : r - ra;
        pix_ok[nb] = (in_a ? imgA_ok : imgB_ok) && oy < p.h && ox < p.wd;
        pixc[nb] = pix_ok[nb] ? (unsigned)((in_a ? imgA : imgA + 1) * p.cout) * uhw +
                                    (unsigned)oy * (unsigned)p.wd + (unsigned)ox
                              : 0u;
    }
    float* red = lds;  // [WPX][CT][2] statistics scratch
    static_assert(WPX * CT * 2 <= PATCH + WSZ, "stat scratch must fit the staging LDS");
    if (stats) __syncthreads();  // every wave is done with the staging LDS
    const float* my = masked ? p.stat_mask_y + (size_t)n * p.cout * hw : nullptr;
#pragma unroll
    for (int m = 0; m < MB; ++m) {
#pragma unroll
        for (int rg = 0; rg < 16; rg += RG) {
            float oldv[RG][NB], yv[RG][NB];
            if (p.accumulate) {
#pragma unroll
                for (int rr = 0; rr < RG; ++rr) {
                    const int r = rg + rr;
                    const int co = co0 + (wave_co * MB + m) * 32 + 4 * khalf + (r & 3) + 8 * (r >> 2);
                    const float* src = yout + (size_t)min(co, p.cout - 1) * hw;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) oldv[rr][nb] = src[pixc[nb]];
                }
            }
            if (masked) {
#pragma unroll
                for (int rr = 0; rr < RG; ++rr) {
                    const int r = rg + rr;
                    const int co = co0 + (wave_co * MB + m) * 32 + 4 * khalf + (r & 3) + 8 * (r >> 2);
                    const float* src = my + (size_t)min(co, p.cout - 1) * hw;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) yv[rr][nb] = src[pixc[nb]];
                }
            }
#pragma unroll
            for (int rr = 0; rr < RG; ++rr) {
                const int r = rg + rr;
                const int cl = (wave_co * MB + m) * 32 + 4 * khalf + (r & 3) + 8 * (r >> 2);
                const int co = co0 + cl;
                const bool co_ok = co < p.cout;
                float* dst = yout + (size_t)co * hw;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    if (p.accumulate) acc[m][nb][r] += oldv[rr][nb];  // the statistics see the sum
                    if (co_ok && pix_ok[nb])
                        dst[pixc[nb]] = acc[m][nb][r];
                }
                if (!stats) continue;
                float s1 = 0.f, s2 = 0.f;
                if (!masked) {  // forward statistics about the pivot
                    const float pv = (p.stat_pivot != nullptr && co_ok) ? p.stat_pivot[co] : 0.f;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const float d = pix_ok[nb] ? acc[m][nb][r] - pv : 0.f;
                        s1 += d;
                        s2 = fmaf(d, d, s2);
                    }
                } else {  // backward sums of the BatchNorm this gradient feeds
                    const float msc = p.mask_scale[min(co, p.cout - 1)], msh = p.mask_shift[min(co, p.cout - 1)];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const bool on = co_ok && pix_ok[nb] &&
                                        (!p.mask_relu || fmaf(yv[rr][nb], msc, msh) > 0.f);
                        const float d = on ? acc[m][nb][r] : 0.f;
                        s1 += d;
                        s2 = fmaf(d, yv[rr][nb], s2);
                    }
                }
                s1 = half_sum32(s1);
                s2 = half_sum32(s2);
                if (j == 31) {
                    red[(wave_px * CT + cl) * 2] = s1;
                    red[(wave_px * CT + cl) * 2 + 1] = s2;
                }
            }
        }
    }
    if (stats) {
        __syncthreads();
        const long long tg = (long long)bz * (p.tiles_x * p.tiles_y) + tile;
        for (int c = tid; c < CT; c += kThreads) {
            if (co0 + c >= p.cout) continue;
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int wp = 0; wp < WPX; ++wp) {
                a += red[(wp * CT + c) * 2];
                b += red[(wp * CT + c) * 2 + 1];
            }
            float* dst = p.stat_part + ((size_t)(co0 + c) * (size_t)p.stat_tiles + (size_t)tg) * 2;
            dst[0] = a;
            dst[1] = b;
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
    int vec_ok;
    // optional: dY is the BatchNorm backward of `dy` (= upstream g), formed while staging
    //   dz = (g*alpha[n][co] + add[n][co]) * [bn_y*coef0[co] + coef1[co] > 0 or !bn_relu]
    //   dY = coef2[co]*dz + coef3[co]*bn_y + coef4[co]
    // and written to dy_out by the ci-block-0 workgroups (each element exactly once)
    const float* bn_y;     // null = plain dY
    const float* bn_alpha;
    const float* bn_add;
    const float* bn_coef;  // [5][Cout]
    float* dy_out;
    int bn_relu;
};

__device__ __forceinline__ float bn_dy1(float g, float y, float al, float ad, float c0, float c1,
                                        float c2, float c3, float c4, int relu) {
    float dz = fmaf(g, al, ad);
    if (relu && !(fmaf(y, c0, c1) > 0.f)) dz = 0.f;
    return fmaf(c2, dz, fmaf(c3, y, c4));
}

template <int TAPS, int TW, int TH, int WCI, int WCO, int KSPL>
__global__ __launch_bounds__(kThreads, (TAPS == 9 ? 2 : 4)) void wgrad_mfma_kernel(WgradArgs p) {
    static_assert(WCI * WCO * KSPL == 4, "wave decomposition");
    static_assert(TH % KSPL == 0 && TW % 4 == 0, "rows split across waves, float4 rows");
    constexpr int HALO = TAPS == 9 ? 1 : 0;
    constexpr int PW = TW + 2 * HALO, PH = TH + 2 * HALO;
    constexpr int PP = (PW * PH) | 1;  // odd plane pitch: 32 lanes on 32 channels hit 32 banks
    constexpr int DP = (TW * TH) | 1;
    constexpr int CI_T = 32 * WCI, CO_T = 32 * WCO;
    constexpr int XSZ = CI_T * PP, DSZ = CO_T * DP;
    constexpr int RED = KSPL > 1 ? TAPS * 1024 : 0;  // one wave's accumulators
    constexpr int LDSF = XSZ + DSZ > RED ? XSZ + DSZ : RED;
    constexpr int TW4 = TW / 4;
    constexpr int NXI = CI_T * PH * TW4, XPT = (NXI + kThreads - 1) / kThreads;      // interior float4 items
    constexpr int NHI = CI_T * PH * 2 * HALO, HPT = (NHI + kThreads - 1) / kThreads;  // halo scalars
    constexpr int NDI = CO_T * TH * TW4, DPT = (NDI + kThreads - 1) / kThreads;
    constexpr int kMaxProC = 512;
    __shared__ float lds[LDSF];
    float* lx = lds;
    float* ld = lds + XSZ;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int w_ci = wid % WCI, w_co = (wid / WCI) % WCO, w_k = wid / (WCI * WCO);
    const int ci0 = blockIdx.y * CI_T, co0 = blockIdx.z * CO_T;
    const int khalf = lane >> 5, j = lane & 31;
    const size_t hw = (size_t)p.h * p.wd;
    const unsigned uhw = (unsigned)hw;

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int abase = (w_ci * 32 + j) * PP + khalf;  // + row*PW + x + tap offset
    const int bbase = (w_co * 32 + j) * DP + khalf;
    const bool pro = p.in_scale != nullptr;

    auto compute_item = [&]() {
        constexpr int ROWS = TH / KSPL;
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            const int row = w_k * ROWS + rr;
            // 9 MFMAs (576 cycles) per pixel pair: a shallow unroll already hides the LDS
            // latency and keeps the 144 accumulators + staging registers under 256
#pragma unroll 2
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
    };

    const int first = blockIdx.x * p.items_per_split;
    const int last = min(first + p.items_per_split, p.items);
    const int tiles = p.tiles_x * p.tiles_y;

    if (p.vec_ok) {  // every tile is full in x (host guarantees W % TW == 0 for this path)
        float4 xv[XPT], dv[DPT];
        float xh[HPT > 0 ? HPT : 1];
        unsigned xok = 0;
        // producer BatchNorm scale/shift of this workgroup's CI_T channels, staged once in LDS
        __shared__ float lsc[kMaxProC / 2];
        if (pro) {
            for (int c = tid; c < CI_T; c += kThreads) {
                const int gc = ci0 + c;
                lsc[c] = gc < p.cin ? p.in_scale[gc] : 1.f;
                lsc[kMaxProC / 4 + c] = gc < p.cin ? p.in_shift[gc] : 0.f;
            }
        }
        // optional: dY = BatchNorm backward of p.dy, formed while staging (see WgradArgs)
        const bool bn = p.bn_y != nullptr;
        float4 yv[DPT];
        float dal[DPT], dad[DPT];
        unsigned dok = 0;
        __shared__ float lbn[5 * CO_T];
        if (bn) {
            for (int e = tid; e < 5 * CO_T; e += kThreads) {
                const int kk = e / CO_T, gc = co0 + (e - kk * CO_T);
                lbn[e] = gc < p.cout ? p.bn_coef[(size_t)kk * p.cout + gc] : 0.f;
            }
        }
        auto load_x = [&](int item, auto i0c, auto i1c) {
            constexpr int I0 = decltype(i0c)::value, I1 = decltype(i1c)::value;
            const int n = item / tiles, t = item - n * tiles;
            const int tx0 = (t % p.tiles_x) * TW, ty0 = (t / p.tiles_x) * TH;
            const float* xin = p.x + (size_t)n * p.cin * hw;
            if (I0 == 0) xok = 0;
#pragma unroll
            for (int i = I0; i < I1; ++i) {
                const int e = tid + i * kThreads;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                const int c = e / (PH * TW4), rem = e - c * (PH * TW4);
                const int py = rem / TW4, slot = rem - py * TW4;
                const int gc = ci0 + c, gy = ty0 + py - HALO;
                if (e < NXI && gc < p.cin && gy >= 0 && gy < p.h) {
                    v = *reinterpret_cast<const float4*>(xin + (unsigned)gc * uhw +
                                                         (unsigned)gy * (unsigned)p.wd + tx0 + 4 * slot);
                    xok |= 1u << i;
                }
                xv[i] = v;
            }
            if (I1 == XPT) {
#pragma unroll
                for (int i = 0; i < HPT; ++i) {
                    const int e = tid + i * kThreads;
                    float v = 0.f;
                    const int c = e / (PH * 2), rem = e - c * (PH * 2);
                    const int py = rem >> 1, side = rem & 1;
                    const int gc = ci0 + c, gy = ty0 + py - HALO, gx = side ? tx0 + TW : tx0 - 1;
                    if (e < NHI && gc < p.cin && gy >= 0 && gy < p.h && gx >= 0 && gx < p.wd) {
                        v = xin[(unsigned)gc * uhw + (unsigned)gy * (unsigned)p.wd + (unsigned)gx];
                        xok |= 1u << (16 + i);
                    }
                    xh[i] = v;
                }
            }
        };
        auto load_d = [&](int item) {
            const int n = item / tiles, t = item - n * tiles;
            const int tx0 = (t % p.tiles_x) * TW, ty0 = (t / p.tiles_x) * TH;
            const float* din = p.dy + (size_t)n * p.cout * hw;
            dok = 0;
#pragma unroll
            for (int i = 0; i < DPT; ++i) {
                const int e = tid + i * kThreads;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                const int c = e / (TH * TW4), rem = e - c * (TH * TW4);
                const int py = rem / TW4, slot = rem - py * TW4;
                const int gc = co0 + c, gy = ty0 + py;
                if (e < NDI && gc < p.cout && gy < p.h) {
                    v = *reinterpret_cast<const float4*>(din + (unsigned)gc * uhw +
                                                         (unsigned)gy * (unsigned)p.wd + tx0 + 4 * slot);
                    dok |= 1u << i;
                }
                dv[i] = v;
            }
            if (bn) {
                const float* yin = p.bn_y + (size_t)n * p.cout * hw;
#pragma unroll
                for (int i = 0; i < DPT; ++i) {
                    const int e = tid + i * kThreads;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    const int c = e / (TH * TW4), rem = e - c * (TH * TW4);
                    const int py = rem / TW4, slot = rem - py * TW4;
                    if (dok >> i & 1u)
                        v = *reinterpret_cast<const float4*>(yin + (unsigned)(co0 + c) * uhw +
                                                             (unsigned)(ty0 + py) * (unsigned)p.wd + tx0 + 4 * slot);
                    yv[i] = v;
                }
#pragma unroll
                for (int i = 0; i < DPT; ++i) {
                    const int gc = min(co0 + (tid + i * kThreads) / (TH * TW4), p.cout - 1);
                    dal[i] = p.bn_alpha != nullptr ? p.bn_alpha[(size_t)n * p.cout + gc] : 1.f;
                    dad[i] = p.bn_add != nullptr ? p.bn_add[(size_t)n * p.cout + gc] : 0.f;
                }
            }
        };
        auto store_x = [&](auto i0c, auto i1c) {
            constexpr int I0 = decltype(i0c)::value, I1 = decltype(i1c)::value;
#pragma unroll
            for (int i = I0; i < I1; ++i) {
                const int e = tid + i * kThreads;
                if (e < NXI) {
                    const int c = e / (PH * TW4), rem = e - c * (PH * TW4);
                    const int py = rem / TW4, slot = rem - py * TW4;
                    float4 v = xv[i];
                    if (pro && (xok >> i & 1u)) {
                        const float sc = lsc[c], sh = lsc[kMaxProC / 4 + c];
                        v.x = pro_apply(v.x, sc, sh, p.in_relu);
                        v.y = pro_apply(v.y, sc, sh, p.in_relu);
                        v.z = pro_apply(v.z, sc, sh, p.in_relu);
                        v.w = pro_apply(v.w, sc, sh, p.in_relu);
                    }
                    float* dst = lx + c * PP + py * PW + HALO + 4 * slot;
                    dst[0] = v.x;
                    dst[1] = v.y;
                    dst[2] = v.z;
                    dst[3] = v.w;
                }
            }
            if (I1 == XPT) {
#pragma unroll
                for (int i = 0; i < HPT; ++i) {
                    const int e = tid + i * kThreads;
                    if (e < NHI) {
                        const int c = e / (PH * 2), rem = e - c * (PH * 2);
                        const int py = rem >> 1, side = rem & 1;
                        float v = xh[i];
                        if (pro && (xok >> (16 + i) & 1u))
                            v = pro_apply(v, lsc[c], lsc[kMaxProC / 4 + c], p.in_relu);
                        lx[c * PP + py * PW + (side ? PW - 1 : 0)] = v;
                    }
                }
            }
        };
        auto store_d = [&](int item) {
            const int sn = item / tiles, st = item - sn * tiles;
            const int stx0 = (st % p.tiles_x) * TW, sty0 = (st / p.tiles_x) * TH;
#pragma unroll
            for (int i = 0; i < DPT; ++i) {
                const int e = tid + i * kThreads;
                if (e < NDI) {
                    const int c = e / (TH * TW4), rem = e - c * (TH * TW4);
                    float4 v = dv[i];
                    if (bn && (dok >> i & 1u)) {
                        const float c0 = lbn[c], c1 = lbn[CO_T + c], c2 = lbn[2 * CO_T + c],
                                    c3 = lbn[3 * CO_T + c], c4 = lbn[4 * CO_T + c];
                        v.x = bn_dy1(v.x, yv[i].x, dal[i], dad[i], c0, c1, c2, c3, c4, p.bn_relu);
                        v.y = bn_dy1(v.y, yv[i].y, dal[i], dad[i], c0, c1, c2, c3, c4, p.bn_relu);
                        v.z = bn_dy1(v.z, yv[i].z, dal[i], dad[i], c0, c1, c2, c3, c4, p.bn_relu);
                        v.w = bn_dy1(v.w, yv[i].w, dal[i], dad[i], c0, c1, c2, c3, c4, p.bn_relu);
                        if (blockIdx.y == 0 && p.dy_out != nullptr) {
                            const int py = rem / TW4, slot = rem - py * TW4;
                            *reinterpret_cast<float4*>(p.dy_out + ((size_t)sn * p.cout + co0 + c) * hw +
                                                       (size_t)(sty0 + py) * p.wd + stx0 + 4 * slot) = v;
                        }
                    }
                    float* dst = ld + c * DP + rem * 4;
                    dst[0] = v.x;
                    dst[1] = v.y;
                    dst[2] = v.z;
                    dst[3] = v.w;
                }
            }
        };
        if (TAPS == 1) {  // few accumulators: next item's loads stay in registers over the MFMAs
            using Z = std::integral_constant<int, 0>;
            using E = std::integral_constant<int, XPT>;
            if (first < last) {
                load_x(first, Z{}, E{});
                load_d(first);
            }
            for (int item = first; item < last; ++item) {
                __syncthreads();
                store_x(Z{}, E{});
                store_d(item);
                __syncthreads();
                if (item + 1 < last) {
                    load_x(item + 1, Z{}, E{});
                    load_d(item + 1);
                }
                compute_item();
            }
        } else {
            // 144 accumulators leave ~100 registers: stage X in two halves (XPT/2 float4 in
            // flight per thread), then dY
            using Z = std::integral_constant<int, 0>;
            using H = std::integral_constant<int, (XPT + 1) / 2>;
            using E = std::integral_constant<int, XPT>;
            for (int item = first; item < last; ++item) {
                __syncthreads();
                load_x(item, Z{}, H{});
                store_x(Z{}, H{});
                load_x(item, H{}, E{});
                store_x(H{}, E{});
                load_d(item);
                store_d(item);
                __syncthreads();
                compute_item();
            }
        }
    } else {
        // scalar staging: thread -> one tile position, walking channels
        constexpr int XPOS = PW * PH, DPOS = TW * TH;
        constexpr int XSLOTS = XPOS <= 64 ? 64 : (XPOS <= 128 ? 128 : 256);
        constexpr int DSLOTS = DPOS <= 64 ? 64 : (DPOS <= 128 ? 128 : 256);
        constexpr int XG = kThreads / XSLOTS, DG = kThreads / DSLOTS;
        static_assert(XPOS <= 256 && DPOS <= 256, "wgrad tiles are at most 256 positions");
        const int xslot = tid % XSLOTS, xgrp = tid / XSLOTS;
        const int dslot = tid % DSLOTS, dgrp = tid / DSLOTS;
        const int xpy = xslot / PW, xpx = xslot - xpy * PW;
        const int dpy = dslot / TW, dpx = dslot - dpy * TW;
        for (int item = first; item < last; ++item) {
            const int n = item / tiles, t = item - n * tiles;
            const int tx0 = (t % p.tiles_x) * TW, ty0 = (t / p.tiles_x) * TH;
            const float* xin = p.x + (size_t)n * p.cin * hw;
            const float* din = p.dy + (size_t)n * p.cout * hw;
            __syncthreads();
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
                        if (pro) v = pro_apply(v, p.in_scale[gc], p.in_shift[gc], p.in_relu);
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
            __syncthreads();
            compute_item();
        }
    }

    // K-split waves fold their accumulators into wave k = 0 through LDS, one (ci,co) block
    // at a time, in a fixed order
    if (KSPL > 1) {
        const int q = w_co * WCI + w_ci;
#pragma unroll 1
        for (int k = 1; k < KSPL; ++k) {
#pragma unroll 1
            for (int qq = 0; qq < WCI * WCO; ++qq) {
                __syncthreads();
                if (w_k == k && q == qq) {
#pragma unroll
                    for (int t = 0; t < TAPS; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) lds[t * 1024 + r * 64 + lane] = acc[t][r];
                }
                __syncthreads();
                if (w_k == 0 && q == qq) {
#pragma unroll
                    for (int t = 0; t < TAPS; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[t][r] += lds[t * 1024 + r * 64 + lane];
                }
            }
        }
    }
    if (w_k == 0) {
        float* out = p.part + (size_t)blockIdx.x * p.cin * TAPS * p.cout;
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


// 3x3 wgrad with the filter rows spread over waves: a workgroup of 3*WCI*WCO*KSPL waves owns a
// (32*WCI ci) x (32*WCO co) weight block; wave (q, tr, k) accumulates the three taps of filter
// row tr (48 accumulators) over its K-split share of each tile's rows.  With so few
// accumulators the next item's tiles are prefetched into registers during the MFMAs.
// floats of (dynamic) LDS: two staging buffers (X patch + dY tile each), or the K-split scratch
template <int TW, int TH, int WCI, int WCO, int KSPL>
constexpr int wgrad3_lds_floats() {
    const int pp = ((TW + 2) * (TH + 2)) | 1, dp = (TW * TH) | 1;
    const int buf = 32 * WCI * pp + 32 * WCO * dp;
    const int red = KSPL > 1 ? 3 * 3 * 1024 : 0;
    return 2 * buf > red ? 2 * buf : red;
}

template <int TW, int TH, int WCI, int WCO, int KSPL>
__global__ __launch_bounds__(64 * 3 * WCI * WCO * KSPL, 3) void wgrad3_kernel(WgradArgs p) {
    constexpr int NT = 64 * 3 * WCI * WCO * KSPL;
    static_assert(TH % KSPL == 0 && TW % 4 == 0, "rows split across waves, float4 rows");
    constexpr int PW = TW + 2, PH = TH + 2;
    constexpr int PP = (PW * PH) | 1;  // odd plane pitch: 32 lanes on 32 channels hit 32 banks
    constexpr int DP = (TW * TH) | 1;
    constexpr int CI_T = 32 * WCI, CO_T = 32 * WCO, NQ = WCI * WCO;
    constexpr int XSZ = CI_T * PP, DSZ = CO_T * DP;
    constexpr int BUF = XSZ + DSZ;  // one staging buffer; the kernel double-buffers
    static_assert(wgrad3_lds_floats<TW, TH, WCI, WCO, KSPL>() >= 2 * BUF, "LDS sizing");
    constexpr int TW4 = TW / 4;
    constexpr int NXI = CI_T * PH * TW4, XPT = (NXI + NT - 1) / NT;      // interior float4 items
    constexpr int NHI = CI_T * PH * 2 * 1, HPT = (NHI + NT - 1) / NT;  // halo scalars
    constexpr int NDI = CO_T * TH * TW4, DPT = (NDI + NT - 1) / NT;
    constexpr int kMaxProC = 512;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tr = wid % 3, rest = wid / 3;
    const int w_ci = rest % WCI, w_co = (rest / WCI) % WCO, w_k = rest / NQ;
    const int ci0 = blockIdx.y * CI_T, co0 = blockIdx.z * CO_T;
    const int khalf = lane >> 5, j = lane & 31;
    const size_t hw = (size_t)p.h * p.wd;
    const unsigned uhw = (unsigned)hw;

    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int abase = (w_ci * 32 + j) * PP + tr * PW + khalf;
    const int bbase = (w_co * 32 + j) * DP + khalf;
    const bool pro = p.in_scale != nullptr;

    // quarter q (0..3) of an item's pixel pairs from staging buffer `buf`
    constexpr int ROWS = TH / KSPL, NK = ROWS * (TW / 2);
    auto compute_quarter = [&](auto qc, int buf) {
        constexpr int Q = decltype(qc)::value;
        constexpr int K0 = NK * Q / 4, K1 = NK * (Q + 1) / 4;
        const float* lx = lds + buf * BUF;
        const float* ld = lx + XSZ;
#pragma unroll
        for (int kk = K0; kk < K1; ++kk) {
            const int row = w_k * ROWS + kk / (TW / 2), xx = 2 * (kk % (TW / 2));
            const float b = ld[bbase + row * TW + xx];
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float a = lx[abase + row * PW + xx + dx];
                acc[dx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[dx], 0, 0, 0);
            }
        }
    };
    using Q0 = std::integral_constant<int, 0>;
    using Q1 = std::integral_constant<int, 1>;
    using Q2 = std::integral_constant<int, 2>;
    using Q3 = std::integral_constant<int, 3>;

    const int first = blockIdx.x * p.items_per_split;
    const int last = min(first + p.items_per_split, p.items);
    const int tiles = p.tiles_x * p.tiles_y;

    if (p.vec_ok) {  // every tile is full in x (host guarantees W % TW == 0 for this path)
        float4 xv[XPT], dv[DPT], yv[DPT];
        float xh[HPT], dal[DPT], dad[DPT];
        unsigned xok = 0, dok = 0;
        const bool bn = p.bn_y != nullptr;
        // producer BatchNorm scale/shift of this workgroup's CI_T channels, staged once in LDS
        __shared__ float lsc[kMaxProC / 2];
        if (pro) {
            for (int c = tid; c < CI_T; c += NT) {
                const int gc = ci0 + c;
                lsc[c] = gc < p.cin ? p.in_scale[gc] : 1.f;
                lsc[kMaxProC / 4 + c] = gc < p.cin ? p.in_shift[gc] : 0.f;
            }
        }
        // BatchNorm-backward coefficients of this workgroup's CO_T channels
        __shared__ float lbn[5 * CO_T];
        if (bn) {
            for (int e = tid; e < 5 * CO_T; e += NT) {
                const int kk = e / CO_T, gc = co0 + (e - kk * CO_T);
                lbn[e] = gc < p.cout ? p.bn_coef[(size_t)kk * p.cout + gc] : 0.f;
            }
        }
        // Per-thread staging invariants, decoded once: an item only moves the tile origin.
        //   *_g: BYTE offset of the element from the patch origin (image n, channel block, row
        //        ty0-1, column tx0-1 for X; row ty0, column tx0 for dY), kBufOob when the
        //        thread's slot / channel does not exist;
        //   *_l: LDS float index | (patch row << 16)
        unsigned xg[XPT], hg[HPT], dg[DPT];
        unsigned xl[XPT], hl[HPT], dl[DPT];
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int e = tid + i * NT;
            const int c = e / (PH * TW4), rem = e - c * (PH * TW4);
            const int py = rem / TW4, slot = rem - py * TW4;
            const bool ok = e < NXI && ci0 + c < p.cin;
            xg[i] = ok ? 4u * ((unsigned)c * uhw + (unsigned)py * (unsigned)p.wd + 1u + 4u * (unsigned)slot) : kBufOob;
            xl[i] = (unsigned)(c * PP + py * PW + 1 + 4 * slot) | ((unsigned)py << 16);
        }
#pragma unroll
        for (int i = 0; i < HPT; ++i) {
            const int e = tid + i * NT;
            const int c = e / (PH * 2), rem = e - c * (PH * 2);
            const int py = rem >> 1, side = rem & 1;
            const bool ok = e < NHI && ci0 + c < p.cin;
            hg[i] = ok ? 4u * ((unsigned)c * uhw + (unsigned)py * (unsigned)p.wd + (side ? TW + 1u : 0u)) : kBufOob;
            hl[i] = (unsigned)(c * PP + py * PW + (side ? PW - 1 : 0)) | ((unsigned)py << 16) |
                    ((unsigned)side << 30);
        }
#pragma unroll
        for (int i = 0; i < DPT; ++i) {
            const int e = tid + i * NT;
            const int c = e / (TH * TW4), rem = e - c * (TH * TW4);
            const int py = rem / TW4, slot = rem - py * TW4;
            const bool ok = e < NDI && co0 + c < p.cout;
            dg[i] = ok ? 4u * ((unsigned)c * uhw + (unsigned)py * (unsigned)p.wd + 4u * (unsigned)slot) : kBufOob;
            dl[i] = (unsigned)(c * DP + rem * 4) | ((unsigned)py << 16);
        }
        const unsigned xbytes = 4u * ((unsigned)min(CI_T, p.cin - ci0) * uhw + (unsigned)p.wd + 1u);
        const unsigned dbytes = 4u * (unsigned)min(CO_T, p.cout - co0) * uhw;
        auto load_item = [&](int item) {
            const int n = item / tiles, t = item - n * tiles;
            const int tx0 = (t % p.tiles_x) * TW, ty0 = (t / p.tiles_x) * TH;
            const size_t torg = (size_t)ty0 * p.wd + tx0;
            const size_t nl = (size_t)n;
            // patch origin = one row up, one column left of the tile (never dereferenced there:
            // rows / columns outside the image get the out-of-range offset)
            const __amdgpu_buffer_rsrc_t rx =
                buf_rsrc(p.x + (nl * p.cin + ci0) * hw + torg - (size_t)p.wd - 1, xbytes);
            const __amdgpu_buffer_rsrc_t rd = buf_rsrc(p.dy + (nl * p.cout + co0) * hw + torg, dbytes);
            xok = 0;
#pragma unroll
            for (int i = 0; i < XPT; ++i) {
                const unsigned gy = (unsigned)(ty0 + (int)((xl[i] >> 16) & 0xffu) - 1);
                const bool ok = xg[i] != kBufOob && gy < (unsigned)p.h;
                xv[i] = buf_load4(rx, ok ? xg[i] : kBufOob);
                xok |= (ok ? 1u : 0u) << i;
            }
#pragma unroll
            for (int i = 0; i < HPT; ++i) {
                const unsigned gy = (unsigned)(ty0 + (int)((hl[i] >> 16) & 0xffu) - 1);
                const bool side = (hl[i] >> 30) & 1u;
                const bool ok = hg[i] != kBufOob && gy < (unsigned)p.h && (side ? tx0 + TW < p.wd : tx0 > 0);
                xh[i] = buf_load1(rx, ok ? hg[i] : kBufOob);
                xok |= (ok ? 1u : 0u) << (16 + i);
            }
            dok = 0;
#pragma unroll
            for (int i = 0; i < DPT; ++i) {
                const unsigned gy = (unsigned)(ty0 + (int)((dl[i] >> 16) & 0xffu));
                const bool ok = dg[i] != kBufOob && gy < (unsigned)p.h;
                dv[i] = buf_load4(rd, ok ? dg[i] : kBufOob);
                dok |= (ok ? 1u : 0u) << i;
            }
            if (bn) {
                const __amdgpu_buffer_rsrc_t ry =
                    buf_rsrc(p.bn_y + ((size_t)n * p.cout + co0) * hw + torg, dbytes);
#pragma unroll
                for (int i = 0; i < DPT; ++i) yv[i] = buf_load4(ry, (dok >> i & 1u) ? dg[i] : kBufOob);
                if (p.bn_alpha != nullptr) {
#pragma unroll
                    for (int i = 0; i < DPT; ++i) {
                        const int e = tid + i * NT;
                        const int gc = min(co0 + e / (TH * TW4), p.cout - 1);
                        dal[i] = p.bn_alpha[(size_t)n * p.cout + gc];
                        dad[i] = p.bn_add != nullptr ? p.bn_add[(size_t)n * p.cout + gc] : 0.f;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < DPT; ++i) {
                        dal[i] = 1.f;
                        dad[i] = 0.f;
                    }
                }
            }
        };
        auto store_item = [&](int item, int buf) {
            float* lx = lds + buf * BUF;
            float* ld = lx + XSZ;
#pragma unroll
            for (int i = 0; i < XPT; ++i) {
                if (tid + i * NT < NXI) {
                    const int c = (tid + i * NT) / (PH * TW4);
                    float4 v = xv[i];  // padding was fetched as zeros
                    if (pro && (xok >> i & 1u)) {
                        const float sc = lsc[c], sh = lsc[kMaxProC / 4 + c];
                        v.x = pro_apply(v.x, sc, sh, p.in_relu);
                        v.y = pro_apply(v.y, sc, sh, p.in_relu);
                        v.z = pro_apply(v.z, sc, sh, p.in_relu);
                        v.w = pro_apply(v.w, sc, sh, p.in_relu);
                    }
                    float* dst = lx + (xl[i] & 0xffffu);
                    {
                        dst[0] = v.x;
                        dst[1] = v.y;
                        dst[2] = v.z;
                        dst[3] = v.w;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < HPT; ++i) {
                if (tid + i * NT < NHI) {
                    const int c = (tid + i * NT) / (PH * 2);
                    float v = xh[i];
                    if (pro && (xok >> (16 + i) & 1u))
                        v = pro_apply(v, lsc[c], lsc[kMaxProC / 4 + c], p.in_relu);
                    lx[hl[i] & 0xffffu] = v;
                }
            }
            const int sn = item / tiles, st = item - sn * tiles;
            const int stx0 = (st % p.tiles_x) * TW, sty0 = (st / p.tiles_x) * TH;
            float* dyo = bn ? p.dy_out + ((size_t)sn * p.cout + co0) * hw + (size_t)sty0 * p.wd + stx0 : nullptr;
#pragma unroll
            for (int i = 0; i < DPT; ++i) {
                if (tid + i * NT < NDI) {
                    const int c = (tid + i * NT) / (TH * TW4);
                    float4 v = dv[i];
                    if (bn && (dok >> i & 1u)) {
                        const float c0 = lbn[c], c1 = lbn[CO_T + c], c2 = lbn[2 * CO_T + c],
                                    c3 = lbn[3 * CO_T + c], c4 = lbn[4 * CO_T + c];
                        v.x = bn_dy1(v.x, yv[i].x, dal[i], dad[i], c0, c1, c2, c3, c4, p.bn_relu);
                        v.y = bn_dy1(v.y, yv[i].y, dal[i], dad[i], c0, c1, c2, c3, c4, p.bn_relu);
                        v.z = bn_dy1(v.z, yv[i].z, dal[i], dad[i], c0, c1, c2, c3, c4, p.bn_relu);
                        v.w = bn_dy1(v.w, yv[i].w, dal[i], dad[i], c0, c1, c2, c3, c4, p.bn_relu);
                        if (blockIdx.y == 0 && p.dy_out != nullptr)
                            *reinterpret_cast<float4*>(dyo + (dg[i] >> 2)) = v;
                    }
                    float* dst = ld + (dl[i] & 0xffffu);
                    {
                        dst[0] = v.x;
                        dst[1] = v.y;
                        dst[2] = v.z;
                        dst[3] = v.w;
                    }
                }
            }
        };
        // double-buffered staging, one barrier per item: while an item's MFMAs run from one
        // buffer the next item (loaded an iteration earlier) is written into the other, and
        // the loads of the item after that are issued
        int cur = 0;
        const int count = last - first;
        auto nth = [&](int idx) { return first + idx; };
        if (count > 0) {
            load_item(nth(0));
            __syncthreads();  // lsc / lbn are staged
            store_item(nth(0), 0);
            if (count > 1) load_item(nth(1));
        }
        __syncthreads();
        // The waves sharing a SIMD (wid, wid+4, wid+8) move through their MFMAs in lock step,
        // so each stages after a different quarter: while one writes LDS the other two keep
        // the MFMA pipe busy.
        const int stage_q = (wid >> 2) % 3;
        for (int idx = 0; idx < count; ++idx) {
            const bool more = idx + 1 < count;
            auto stage = [&]() {
                store_item(nth(idx + 1), cur ^ 1);
                if (idx + 2 < count) load_item(nth(idx + 2));
            };
            compute_quarter(Q0{}, cur);
            if (more && stage_q == 0) stage();
            compute_quarter(Q1{}, cur);
            if (more && stage_q == 1) stage();
            compute_quarter(Q2{}, cur);
            if (more && stage_q == 2) stage();
            compute_quarter(Q3{}, cur);
            __syncthreads();
            cur ^= 1;
        }
    } else {
        float* lx = lds;
        float* ld = lds + XSZ;
        // scalar staging for ragged shapes
        for (int item = first; item < last; ++item) {
            const int n = item / tiles, t = item - n * tiles;
            const int tx0 = (t % p.tiles_x) * TW, ty0 = (t / p.tiles_x) * TH;
            const float* xin = p.x + (size_t)n * p.cin * hw;
            const float* din = p.dy + (size_t)n * p.cout * hw;
            __syncthreads();
            for (int e = tid; e < CI_T * PW * PH; e += NT) {
                const int c = e / (PW * PH), rem = e - c * (PW * PH);
                const int py = rem / PW, px = rem - py * PW;
                const int gc = ci0 + c, gy = ty0 + py - 1, gx = tx0 + px - 1;
                float v = 0.f;
                if (gc < p.cin && gy >= 0 && gy < p.h && gx >= 0 && gx < p.wd) {
                    v = xin[(unsigned)gc * uhw + (unsigned)gy * (unsigned)p.wd + (unsigned)gx];
                    if (pro) v = pro_apply(v, p.in_scale[gc], p.in_shift[gc], p.in_relu);
                }
                lx[c * PP + rem] = v;
            }
            for (int e = tid; e < CO_T * TW * TH; e += NT) {
                const int c = e / (TW * TH), rem = e - c * (TW * TH);
                const int py = rem / TW, px = rem - py * TW;
                const int gc = co0 + c, gy = ty0 + py, gx = tx0 + px;
                float v = 0.f;
                if (gc < p.cout && gy < p.h && gx < p.wd)
                    v = din[(unsigned)gc * uhw + (unsigned)gy * (unsigned)p.wd + (unsigned)gx];
                ld[c * DP + rem] = v;
            }
            __syncthreads();
            compute_quarter(Q0{}, 0);
            compute_quarter(Q1{}, 0);
            compute_quarter(Q2{}, 0);
            compute_quarter(Q3{}, 0);
        }
    }

    // K-split partner waves fold into k = 0 through LDS, one (ci,co) block per round (fixed order)
    const int q = w_co * WCI + w_ci;
#pragma unroll 1
    for (int k = 1; k < KSPL; ++k) {
#pragma unroll 1
        for (int qq = 0; qq < NQ; ++qq) {
            __syncthreads();
            if (w_k == k && q == qq) {
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) lds[(tr * 3 + t) * 1024 + r * 64 + lane] = acc[t][r];
            }
            __syncthreads();
            if (w_k == 0 && q == qq) {
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] += lds[(tr * 3 + t) * 1024 + r * 64 + lane];
            }
        }
    }
    if (w_k == 0) {
        float* out = p.part + (size_t)blockIdx.x * p.cin * 9 * p.cout;
        const int co = co0 + w_co * 32 + j;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + w_ci * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (ci < p.cin && co < p.cout)
                    out[((size_t)ci * 9 + tr * 3 + t) * p.cout + co] = acc[t][r];
            }
    }
}

// Small-Cin 3x3 wgrad (the stem, Cin*9 <= 32): D[(ci,tap)][co] — the 27 (ci, tap) pairs ride
// the MFMA's M dimension, so a pixel pair costs ONE MFMA per 32 output channels instead of 9.
// Tile 32 x 8; 4 waves split the 8 rows; X patch [cin][10][34] (+ one zero plane for the
// unused M rows), dY tile [32][256].  grid = (splits, 1, cout/32 blocks).
template <int TW, int TH>
__global__ __launch_bounds__(kThreads, 4) void wgrad_smallcin_kernel(WgradArgs p) {
    constexpr int PW = TW + 2, PH = TH + 2, PP = PW * PH;
    constexpr int DP = (TW * TH) | 1;
    constexpr int MAXC = 3;
    constexpr int TW4 = TW / 4;
    __shared__ float lds[(MAXC + 1) * PP + 32 * DP];
    float* lx = lds;
    float* ld = lds + (MAXC + 1) * PP;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int khalf = lane >> 5, j = lane & 31;
    const int co0 = blockIdx.z * 32;
    const size_t hw = (size_t)p.h * p.wd;
    const unsigned uhw = (unsigned)hw;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // lane j < cin*9 reads channel j/9 at tap j%9; other lanes read the zero plane
    const int ktot = p.cin * 9;
    const int aci = j < ktot ? j / 9 : MAXC, atap = j < ktot ? j % 9 : 0;
    const int abase = aci * PP + (atap / 3) * PW + (atap % 3) + khalf;
    const int bbase = j * DP + khalf;
    for (int i = tid; i < PP; i += kThreads) lx[MAXC * PP + i] = 0.f;
    const int first = blockIdx.x * p.items_per_split;
    const int last = min(first + p.items_per_split, p.items);
    const int tiles = p.tiles_x * p.tiles_y;
    const bool pro = p.in_scale != nullptr;
    const bool bn = p.bn_y != nullptr;  // dY = BatchNorm backward of p.dy, formed while staging
    __shared__ float lbn[5 * 32];
    if (bn) {
        for (int e = tid; e < 5 * 32; e += kThreads) {
            const int kk = e / 32, gc = co0 + (e - kk * 32);
            lbn[e] = gc < p.cout ? p.bn_coef[(size_t)kk * p.cout + gc] : 0.f;
        }
    }
    constexpr int ROWS = TH / 4;
    for (int item = first; item < last; ++item) {
        const int n = item / tiles, t = item - n * tiles;
        const int tx0 = (t % p.tiles_x) * TW, ty0 = (t / p.tiles_x) * TH;
        const float* xin = p.x + (size_t)n * p.cin * hw;
        const float* din = p.dy + (size_t)n * p.cout * hw;
        __syncthreads();
        for (int e = tid; e < p.cin * PP; e += kThreads) {
            const int c = e / PP, rem = e - c * PP;
            const int py = rem / PW, px = rem - py * PW;
            const int gy = ty0 + py - 1, gx = tx0 + px - 1;
            float v = 0.f;
            if (gy >= 0 && gy < p.h && gx >= 0 && gx < p.wd) {
                v = xin[(unsigned)c * uhw + (unsigned)gy * (unsigned)p.wd + (unsigned)gx];
                if (pro) v = pro_apply(v, p.in_scale[c], p.in_shift[c], p.in_relu);
            }
            lx[e] = v;
        }
        if (p.vec_ok) {
#pragma unroll
            for (int i = 0; i < 32 * TH * TW4 / kThreads; ++i) {
                const int e = tid + i * kThreads;
                const int c = e / (TH * TW4), rem = e - c * (TH * TW4);
                const int py = rem / TW4, slot = rem - py * TW4;
                const int gc = co0 + c, gy = ty0 + py;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gc < p.cout && gy < p.h) {
                    const size_t off = (size_t)gc * uhw + (size_t)gy * p.wd + tx0 + 4 * slot;
                    v = *reinterpret_cast<const float4*>(din + off);
                    if (bn) {
                        const float4 yv = *reinterpret_cast<const float4*>(
                            p.bn_y + (size_t)n * p.cout * hw + off);
                        const float al = p.bn_alpha ? p.bn_alpha[(size_t)n * p.cout + gc] : 1.f;
                        const float ad = p.bn_add ? p.bn_add[(size_t)n * p.cout + gc] : 0.f;
                        const float c0 = lbn[c], c1 = lbn[32 + c], c2 = lbn[64 + c], c3 = lbn[96 + c],
                                    c4 = lbn[128 + c];
                        v.x = bn_dy1(v.x, yv.x, al, ad, c0, c1, c2, c3, c4, p.bn_relu);
                        v.y = bn_dy1(v.y, yv.y, al, ad, c0, c1, c2, c3, c4, p.bn_relu);
                        v.z = bn_dy1(v.z, yv.z, al, ad, c0, c1, c2, c3, c4, p.bn_relu);
                        v.w = bn_dy1(v.w, yv.w, al, ad, c0, c1, c2, c3, c4, p.bn_relu);
                        if (blockIdx.y == 0 && p.dy_out != nullptr)
                            *reinterpret_cast<float4*>(p.dy_out + (size_t)n * p.cout * hw + off) = v;
                    }
                }
                float* dst = ld + c * DP + rem * 4;
                dst[0] = v.x;
                dst[1] = v.y;
                dst[2] = v.z;
                dst[3] = v.w;
            }
        } else {
            for (int e = tid; e < 32 * TW * TH; e += kThreads) {
                const int c = e / (TW * TH), rem = e - c * (TW * TH);
                const int py = rem / TW, px = rem - py * TW;
                const int gc = co0 + c, gy = ty0 + py, gx = tx0 + px;
                float v = 0.f;
                if (gc < p.cout && gy < p.h && gx < p.wd)
                    v = din[(unsigned)gc * uhw + (unsigned)gy * (unsigned)p.wd + (unsigned)gx];
                ld[c * DP + rem] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            const int row = wid * ROWS + rr;
#pragma unroll 8
            for (int xx = 0; xx < TW; xx += 2) {
                const float a = lx[abase + row * PW + xx];
                const float b = ld[bbase + row * TW + xx];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
        }
    }
    // fold the 4 row-split waves through LDS
#pragma unroll 1
    for (int k = 1; k < 4; ++k) {
        __syncthreads();
        if (wid == k)
#pragma unroll
            for (int r = 0; r < 16; ++r) lds[r * 64 + lane] = acc[r];
        __syncthreads();
        if (wid == 0)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += lds[r * 64 + lane];
    }
    if (wid == 0) {
        float* out = p.part + (size_t)blockIdx.x * p.cin * 9 * p.cout;
        const int co = co0 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kidx = (r & 3) + 8 * (r >> 2) + 4 * khalf;  // = ci*9 + tap
            if (kidx < ktot && co < p.cout) out[(size_t)kidx * p.cout + co] = acc[r];
        }
    }
}

// dst[g][i] = sum over slabs s in group g of part[s][i]; with one group and dw given this is
// the final dw = beta*dw + sum.  Fixed order -> deterministic.
__global__ __launch_bounds__(kThreads) void slab_reduce_kernel(const float* __restrict__ part,
                                                               float* __restrict__ dst, size_t count,
                                                               int nslabs, int per_group, float beta,
                                                               int final_pass) {
    const int g = blockIdx.y;
    const int s0 = g * per_group, s1 = min(s0 + per_group, nslabs);
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < count;
         i += (size_t)gridDim.x * kThreads) {
        float s = 0.f;
        for (int k = s0; k < s1; ++k) s += part[(size_t)k * count + i];
        float* o = dst + (size_t)g * count + i;
        *o = (final_pass && beta != 0.f) ? fmaf(beta, *o, s) : s;
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
constexpr FwdVariant kFwdVariants[] = {{32, 8, 32}, {32, 8, 64}, {16, 16, 32}, {16, 16, 64},
                                       {28, 8, 128}, {32, 8, 128}, {56, 8, 64}};
constexpr int kNumFwd = sizeof(kFwdVariants) / sizeof(kFwdVariants[0]);

inline long long padded_work(const FwdVariant& v, int h, int w, int cout) {
    const long long tx = (w + v.tw - 1) / v.tw, ty = (h + v.th - 1) / v.th,
                    tc = (cout + v.ct - 1) / v.ct;
    return tx * v.tw * ty * v.th * tc * v.ct;
}

// K-chunk of 8 input channels everywhere (16 measured slower: more prefetch registers, fewer resident waves)
template <int TAPS>
int launch_fwd(int variant, const ConvArgs& a, dim3 grid, hipStream_t s) {
    constexpr int KS = TAPS == 9 ? LF_KC_SMALL : 8;
    if (TAPS == 9 && variant == 0 && a.cin <= 4) {
        // the stem (Cin = 3): a 4-channel K-chunk instead of 8 halves the MFMAs spent on zeros
        conv_mfma_kernel<9, 32, 8, 1, 1, 4, 2, 4><<<grid, kThreads, 0, s>>>(a);
        return LF_OK;
    }
    switch (variant) {
        case 0: conv_mfma_kernel<TAPS, 32, 8, 1, 1, 4, 2, KS><<<grid, kThreads, 0, s>>>(a); break;
        case 1: conv_mfma_kernel<TAPS, 32, 8, 1, 2, 4, 2, KS><<<grid, kThreads, 0, s>>>(a); break;
        case 2: conv_mfma_kernel<TAPS, 16, 16, 1, 1, 4, 2, KS><<<grid, kThreads, 0, s>>>(a); break;
        case 3: conv_mfma_kernel<TAPS, 16, 16, 1, 2, 4, 2, KS><<<grid, kThreads, 0, s>>>(a); break;
        case 4: conv_mfma_kernel<TAPS, 28, 8, 4, 1, 1, 7, 8, true><<<grid, kThreads, 0, s>>>(a); break;
        case 5: conv_mfma_kernel<TAPS, 32, 8, 2, 2, 2, 4, 8><<<grid, kThreads, 0, s>>>(a); break;
        case 6: conv_mfma_kernel<TAPS, 56, 8, 2, 1, 2, 7, 8><<<grid, kThreads, 0, s>>>(a); break;
        default: return LF_ERR_INVALID;
    }
    return LF_OK;
}

struct WgVariant {
    int tw, th, ci_t, co_t, kspl;
};
constexpr WgVariant kWgVariants[] = {{32, 4, 32, 32, 4}, {16, 8, 32, 64, 2}, {16, 4, 64, 64, 1},
                                     {28, 2, 64, 64, 1}, {32, 4, 32, 64, 2}};
constexpr int kWgSmallCin = 5;  // variant id of wgrad_smallcin_kernel<32, 8>

// wgrad3 uses more than the 64 KB of LDS a kernel gets by default: raise the limit once
template <int TW, int TH, int WCI, int WCO, int KSPL>
int launch_wgrad3(const WgradArgs& a, dim3 grid, hipStream_t s) {
    constexpr size_t bytes = (size_t)wgrad3_lds_floats<TW, TH, WCI, WCO, KSPL>() * sizeof(float);
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3_kernel<TW, TH, WCI, WCO, KSPL>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
            lf::set_error("lf_conv2d_wgrad: cannot reserve %zu bytes of LDS", bytes);
            return LF_ERR_LAUNCH;
        }
        raised = true;
    }
    wgrad3_kernel<TW, TH, WCI, WCO, KSPL><<<grid, 64 * 3 * WCI * WCO * KSPL, bytes, s>>>(a);
    return LF_OK;
}

int launch_wgrad(int ksize, int variant, const WgradArgs& a, dim3 grid, hipStream_t s) {
    if (ksize == 3) {
        switch (variant) {
            case 0: return launch_wgrad3<32, 4, 1, 1, 4>(a, grid, s);
            case 1: return launch_wgrad3<16, 8, 1, 2, 2>(a, grid, s);
            case 2: return launch_wgrad3<16, 4, 2, 2, 1>(a, grid, s);
            case 3: return launch_wgrad3<28, 2, 2, 2, 1>(a, grid, s);
            case 4: return launch_wgrad3<32, 4, 1, 2, 2>(a, grid, s);
            default: return LF_ERR_INVALID;
        }
    } else {
        switch (variant) {
            case 0: wgrad_mfma_kernel<1, 32, 4, 1, 1, 4><<<grid, kThreads, 0, s>>>(a); break;
            case 1: wgrad_mfma_kernel<1, 16, 8, 1, 2, 2><<<grid, kThreads, 0, s>>>(a); break;
            case 2: wgrad_mfma_kernel<1, 16, 4, 2, 2, 1><<<grid, kThreads, 0, s>>>(a); break;
            case 3: wgrad_mfma_kernel<1, 28, 2, 2, 2, 1><<<grid, kThreads, 0, s>>>(a); break;
            case 4: wgrad_mfma_kernel<1, 32, 4, 1, 2, 2><<<grid, kThreads, 0, s>>>(a); break;
            default: return LF_ERR_INVALID;
        }
    }
    return LF_OK;
}

struct WgPlan {
    int variant, tw, tiles_x, tiles_y, items, splits, items_per_split, gy, gz;
};

WgPlan plan_wgrad(int n, int cin, int cout, int h, int w, int ksize) {
    WgPlan best{};
    if (ksize == 3 && cin * 9 <= 32) {
        best.variant = kWgSmallCin;
        best.tw = 32;
        best.tiles_x = (w + 31) / 32;
        best.tiles_y = (h + 7) / 8;
        best.gy = 1;
        best.gz = (cout + 31) / 32;
    } else {
        long long best_cost = -1;
        for (int v = 0; v < (int)(sizeof(kWgVariants) / sizeof(kWgVariants[0])); ++v) {
            const WgVariant& k = kWgVariants[v];
            const long long tx = (w + k.tw - 1) / k.tw, ty = (h + k.th - 1) / k.th;
            const long long gy = (cin + k.ci_t - 1) / k.ci_t, gz = (cout + k.co_t - 1) / k.co_t;
            const long long cost = tx * k.tw * ty * k.th * gy * k.ci_t * gz * k.co_t;
            if (best_cost < 0 || cost < best_cost) {
                best_cost = cost;
                best.variant = v;
                best.tw = k.tw;
                best.tiles_x = (int)tx;
                best.tiles_y = (int)ty;
                best.gy = (int)gy;
                best.gz = (int)gz;
            }
        }
    }
    best.items = n * best.tiles_x * best.tiles_y;
    // every split gets the same number of items
    // resident workgroups per CU: the 12-wave 3x3 kernels 1 (x2 rounds), the others ~3
    int splits = (256 * (ksize == 3 && best.variant != kWgSmallCin ? 2 : 3)) / (best.gy * best.gz);
    if (splits < 1) splits = 1;
    if (splits > best.items) splits = best.items;
    best.items_per_split = (best.items + splits - 1) / splits;
    best.splits = (best.items + best.items_per_split - 1) / best.items_per_split;
    return best;
}

constexpr int kReduceGroup = 32;  // slabs summed per first-stage workgroup row

inline bool aligned16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

// Images per workgroup strip: 2 when the 28x8 tile would leave the last tile of every image
// half empty (H = 28: 3.5 tiles) and the shape meets what the strip path needs (full-width
// vector tiles, Cin a whole number of K-chunks, two images within 32-bit byte offsets).
inline int conv_stack(int variant, int n, int cin, int h, int wd, int cout) {
    const FwdVariant& v = kFwdVariants[variant];
    if (variant != 4 || n < 2 || v.th > h) return 1;
    if (h % v.th == 0 || (2 * h) % v.th != 0) return 1;
    if (wd % v.tw != 0 || wd % 4 != 0 || cout % 4 != 0 || cin % 8 != 0) return 1;
    if ((size_t)2 * cin * h * wd >= (1ull << 29) || (size_t)2 * cout * h * wd >= (1ull << 30)) return 1;
    return 2;
}

}  // namespace

extern "C" {

int lf_conv2d_wgrad_bn_supported(int n, int cin, int h, int wd, int cout, int ksize);

int lf_conv2d_variant(int h, int wd, int cout, int ksize) {
    int best = 0;
    long long bw = -1;
    for (int v = 0; v < kNumFwd; ++v) {
        const long long c = padded_work(kFwdVariants[v], h, wd, cout);
        // 1x1 convolutions are bandwidth-bound: among equally padded tilings take the widest
        // cout tile, so that the input tile is read once instead of once per cout tile
        const bool better = bw < 0 || c < bw ||
                            (ksize == 1 && c == bw && kFwdVariants[v].ct > kFwdVariants[best].ct);
        if (better) {
            bw = c;
            best = v;
        }
    }
    return best;
}

int lf_conv2d_wgrad_variant(int n, int cin, int h, int wd, int cout, int ksize) {
    if (n <= 0 || cin <= 0 || cout <= 0 || h <= 0 || wd <= 0) return -1;
    return plan_wgrad(n, cin, cout, h, wd, ksize).variant;
}

static int conv2d_launch(const char* who, const float* x, const float* w, float* y, int n, int cin,
                         int h, int wd, int cout, int ksize, const float* in_scale,
                         const float* in_shift, int in_relu, int accumulate, float* stat_part,
                         const float* stat_pivot, const float* stat_mask_y, const float* mask_scale,
                         const float* mask_shift, int mask_relu, lf_stream_t stream) {
    LF_REQUIRE(x && w && y, "%s: null buffer", who);
    LF_REQUIRE(n > 0 && cin > 0 && cout > 0 && h > 0 && wd > 0,
               "%s: bad dims n=%d cin=%d cout=%d h=%d w=%d", who, n, cin, cout, h, wd);
    LF_REQUIRE(ksize == 3 || ksize == 1, "%s: ksize must be 1 or 3 (got %d)", who, ksize);
    LF_REQUIRE((in_scale == nullptr) == (in_shift == nullptr),
               "%s: in_scale/in_shift must both be set", who);
    LF_REQUIRE(n <= 65535, "%s: batch too large for grid.z", who);
    LF_REQUIRE((size_t)cin * h * wd < (1ull << 30) && (size_t)cin * ksize * ksize * cout < (1ull << 30),
               "%s: per-image tensor too large for 32-bit offsets", who);
    const int best = lf_conv2d_variant(h, wd, cout, ksize);
    const FwdVariant& v = kFwdVariants[best];
    ConvArgs a;
    a.x = x; a.w = w; a.y = y; a.in_scale = in_scale; a.in_shift = in_shift;
    a.n = n; a.cin = cin; a.cout = cout; a.h = h; a.wd = wd;
    a.in_relu = in_relu;
    a.accumulate = accumulate;
    a.vec_ok = (wd % 4 == 0) && (cout % 4 == 0) && aligned16(x) && aligned16(w);
    a.stack = conv_stack(best, n, cin, h, wd, cout);
    if (a.stack > 1 && !a.vec_ok) {
        if (stat_part != nullptr) {  // the tile count the caller sized its buffers for assumes it
            lf::set_error("%s: the statistics epilogue needs 16-byte aligned x / w", who);
            return LF_ERR_INVALID;
        }
        a.stack = 1;
    }
    a.tiles_x = (wd + v.tw - 1) / v.tw;
    a.tiles_y = (a.stack * h + v.th - 1) / v.th;
    a.stat_part = stat_part;
    a.stat_pivot = stat_pivot;
    a.stat_mask_y = stat_mask_y; a.mask_scale = mask_scale; a.mask_shift = mask_shift;
    a.mask_relu = mask_relu;
    const int gz = (n + a.stack - 1) / a.stack;
    a.stat_tiles = (long long)gz * a.tiles_x * a.tiles_y;
    dim3 grid(a.tiles_x * a.tiles_y, (cout + v.ct - 1) / v.ct, gz);
    hipStream_t s = lf::as_stream(stream);
    const int rc = ksize == 3 ? launch_fwd<9>(best, a, grid, s) : launch_fwd<1>(best, a, grid, s);
    if (rc != LF_OK) return rc;
    return lf::check_launch(who);
}

int lf_conv2d_f32(const float* x, const float* w, float* y, int n, int cin, int h, int wd, int cout,
                  int ksize, const float* in_scale, const float* in_shift, int in_relu,
                  int accumulate, lf_stream_t stream) {
    return conv2d_launch("lf_conv2d", x, w, y, n, cin, h, wd, cout, ksize, in_scale, in_shift, in_relu,
                         accumulate, nullptr, nullptr, nullptr, nullptr, nullptr, 0, stream);
}

long long lf_conv2d_stats_tiles(int n, int cin, int h, int wd, int cout, int ksize) {
    if (n <= 0 || cin <= 0 || h <= 0 || wd <= 0 || cout <= 0) return 0;
    const int best = lf_conv2d_variant(h, wd, cout, ksize);
    const FwdVariant& v = kFwdVariants[best];
    const int stack = conv_stack(best, n, cin, h, wd, cout);
    return (long long)((n + stack - 1) / stack) * ((wd + v.tw - 1) / v.tw) * ((stack * h + v.th - 1) / v.th);
}

int lf_conv2d_stats_f32(const float* x, const float* w, float* y, int n, int cin, int h, int wd,
                        int cout, int ksize, const float* in_scale, const float* in_shift,
                        int in_relu, const float* pivot, float* tile_part, size_t tile_part_bytes,
                        lf_stream_t stream) {
    LF_REQUIRE(tile_part, "lf_conv2d_stats: null tile_part");
    const long long tiles = lf_conv2d_stats_tiles(n, cin, h, wd, cout, ksize);
    if (tile_part_bytes < (size_t)tiles * (size_t)(cout > 0 ? cout : 0) * 2 * sizeof(float)) {
        lf::set_error("lf_conv2d_stats: tile_part %zu bytes < %lld tiles x %d channels x 8",
                      tile_part_bytes, tiles, cout);
        return LF_ERR_WORKSPACE;
    }
    return conv2d_launch("lf_conv2d_stats", x, w, y, n, cin, h, wd, cout, ksize, in_scale, in_shift,
                         in_relu, 0, tile_part, pivot, nullptr, nullptr, nullptr, 0, stream);
}

int lf_conv2d_bnbwd_f32(const float* x, const float* w, float* y, int n, int cin, int h, int wd,
                        int cout, int ksize, int accumulate, const float* mask_y,
                        const float* mask_scale, const float* mask_shift, int mask_relu,
                        float* tile_part, size_t tile_part_bytes, lf_stream_t stream) {
    LF_REQUIRE(tile_part && mask_y && mask_scale && mask_shift, "lf_conv2d_bnbwd: null buffer");
    const long long tiles = lf_conv2d_stats_tiles(n, cin, h, wd, cout, ksize);
    if (tile_part_bytes < (size_t)tiles * (size_t)(cout > 0 ? cout : 0) * 2 * sizeof(float)) {
        lf::set_error("lf_conv2d_bnbwd: tile_part %zu bytes < %lld tiles x %d channels x 8",
                      tile_part_bytes, tiles, cout);
        return LF_ERR_WORKSPACE;
    }
    return conv2d_launch("lf_conv2d_bnbwd", x, w, y, n, cin, h, wd, cout, ksize, nullptr, nullptr, 0,
                         accumulate, tile_part, nullptr, mask_y, mask_scale, mask_shift, mask_relu,
                         stream);
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
    const WgPlan pl = plan_wgrad(n, cin, cout, h, wd, ksize);
    const size_t count = (size_t)cin * ksize * ksize * cout;
    const size_t groups = (pl.splits + kReduceGroup - 1) / kReduceGroup;
    return ((size_t)pl.splits + groups) * count * sizeof(float);
}

static int wgrad_launch(const char* who, const float* x, const float* dy, int n, int cin, int h,
                        int wd, int cout, int ksize, const float* in_scale, const float* in_shift,
                        int in_relu, const float* bn_y, const float* bn_alpha, const float* bn_add,
                        const float* bn_coef, int bn_relu, float* dy_out, void* workspace,
                        size_t ws_bytes, lf_stream_t stream) {
    LF_REQUIRE(x && dy && workspace, "%s: null buffer", who);
    LF_REQUIRE(n > 0 && cin > 0 && cout > 0 && h > 0 && wd > 0,
               "%s: bad dims n=%d cin=%d cout=%d h=%d w=%d", who, n, cin, cout, h, wd);
    LF_REQUIRE(ksize == 3 || ksize == 1, "%s: ksize must be 1 or 3 (got %d)", who, ksize);
    LF_REQUIRE((in_scale == nullptr) == (in_shift == nullptr),
               "%s: in_scale/in_shift must both be set", who);
    LF_REQUIRE((size_t)cin * h * wd < (1ull << 30) && (size_t)cout * h * wd < (1ull << 30),
               "%s: per-image tensor too large for 32-bit offsets", who);
    const WgPlan pl = plan_wgrad(n, cin, cout, h, wd, ksize);
    if (ws_bytes < lf_conv2d_wgrad_workspace(n, cin, h, wd, cout, ksize)) {
        lf::set_error("%s: workspace %zu < %zu bytes", who, ws_bytes,
                      lf_conv2d_wgrad_workspace(n, cin, h, wd, cout, ksize));
        return LF_ERR_WORKSPACE;
    }
    WgradArgs a;
    a.x = x; a.dy = dy; a.part = static_cast<float*>(workspace);
    a.in_scale = in_scale; a.in_shift = in_shift;
    a.n = n; a.cin = cin; a.cout = cout; a.h = h; a.wd = wd;
    a.tiles_x = pl.tiles_x; a.tiles_y = pl.tiles_y; a.items = pl.items;
    a.items_per_split = pl.items_per_split;
    a.in_relu = in_relu;
    a.vec_ok = (wd % pl.tw == 0) && (wd % 4 == 0) && aligned16(x) && aligned16(dy);
    a.bn_y = bn_y; a.bn_alpha = bn_alpha; a.bn_add = bn_add; a.bn_coef = bn_coef;
    a.dy_out = dy_out; a.bn_relu = bn_relu;
    if (bn_y != nullptr) {
        LF_REQUIRE(bn_coef, "%s: bn_coef missing", who);
        LF_REQUIRE(bn_add == nullptr || bn_alpha != nullptr, "%s: bn_add needs bn_alpha", who);
        LF_REQUIRE(lf_conv2d_wgrad_bn_supported(n, cin, h, wd, cout, ksize) && a.vec_ok &&
                       aligned16(bn_y) && (dy_out == nullptr || aligned16(dy_out)),
                   "%s: shape/alignment not supported by the fused BatchNorm-backward path "
                   "(ask lf_conv2d_wgrad_bn_supported first)", who);
    }
    dim3 grid(pl.splits, pl.gy, pl.gz);
    hipStream_t s = lf::as_stream(stream);
    int rc = LF_OK;
    if (pl.variant == kWgSmallCin)
        wgrad_smallcin_kernel<32, 8><<<grid, kThreads, 0, s>>>(a);
    else
        rc = launch_wgrad(ksize, pl.variant, a, grid, s);
    if (rc != LF_OK) return rc;
    return lf::check_launch(who);
}

int lf_conv2d_wgrad_f32(const float* x, const float* dy, int n, int cin, int h, int wd, int cout,
                        int ksize, const float* in_scale, const float* in_shift, int in_relu,
                        void* workspace, size_t ws_bytes, lf_stream_t stream) {
    return wgrad_launch("lf_conv2d_wgrad", x, dy, n, cin, h, wd, cout, ksize, in_scale, in_shift,
                        in_relu, nullptr, nullptr, nullptr, nullptr, 0, nullptr, workspace, ws_bytes,
                        stream);
}

int lf_conv2d_wgrad_bn_supported(int n, int cin, int h, int wd, int cout, int ksize) {
    if (n <= 0 || cin <= 0 || cout <= 0 || h <= 0 || wd <= 0 || (ksize != 3 && ksize != 1)) return 0;
    const WgPlan pl = plan_wgrad(n, cin, cout, h, wd, ksize);
    return (wd % pl.tw == 0) && (wd % 4 == 0);
}

int lf_conv2d_wgrad_bn_f32(const float* x, const float* g, const float* bn_y,
                           const float* alpha_nc, const float* add_nc, const float* coef,
                           int bn_relu, float* dy_out, int n, int cin, int h, int wd, int cout,
                           int ksize, const float* in_scale, const float* in_shift, int in_relu,
                           void* workspace, size_t ws_bytes, lf_stream_t stream) {
    LF_REQUIRE(bn_y, "lf_conv2d_wgrad_bn: null bn_y");
    return wgrad_launch("lf_conv2d_wgrad_bn", x, g, n, cin, h, wd, cout, ksize, in_scale, in_shift,
                        in_relu, bn_y, alpha_nc, add_nc, coef, bn_relu, dy_out, workspace, ws_bytes,
                        stream);
}

int lf_conv2d_wgrad_reduce_f32(void* workspace, float* dw, int n, int cin, int h, int wd, int cout,
                               int ksize, float beta, lf_stream_t stream) {
    LF_REQUIRE(workspace && dw, "lf_conv2d_wgrad_reduce: null buffer");
    LF_REQUIRE(n > 0 && cin > 0 && cout > 0 && h > 0 && wd > 0 && (ksize == 1 || ksize == 3),
               "lf_conv2d_wgrad_reduce: bad dims");
    const WgPlan pl = plan_wgrad(n, cin, cout, h, wd, ksize);
    const size_t count = (size_t)cin * ksize * ksize * cout;
    float* part = static_cast<float*>(workspace);
    hipStream_t s = lf::as_stream(stream);
    const unsigned gx = lf::stream_grid(count, kThreads, 1024);
    if (pl.splits <= kReduceGroup) {
        slab_reduce_kernel<<<dim3(gx, 1), kThreads, 0, s>>>(part, dw, count, pl.splits, pl.splits, beta, 1);
    } else {
        const int groups = (pl.splits + kReduceGroup - 1) / kReduceGroup;
        float* stage = part + (size_t)pl.splits * count;
        slab_reduce_kernel<<<dim3(gx, groups), kThreads, 0, s>>>(part, stage, count, pl.splits,
                                                                 kReduceGroup, 0.f, 0);
        slab_reduce_kernel<<<dim3(gx, 1), kThreads, 0, s>>>(stage, dw, count, groups, groups, beta, 1);
    }
    return lf::check_launch("lf_conv2d_wgrad_reduce");
}

}  // extern "C"
