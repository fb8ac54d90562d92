// libleafhip — forward convolution with bf16 operands and fp32 accumulation (inference only).
//
// The reference trains and predicts under Keras' `mixed_float16` policy unless told otherwise
// (train.py:53-117, `--no-mixed-precision`); BASELINE configs[4] asks for reduced-precision
// inference.  This is that mode for the conv stack: activations and weights stay fp32 in HBM (the
// BatchNorm / SE / residual-tail kernels are shared with the fp32 path), the convolution rounds its
// two operands to bf16 while staging them and multiplies on `v_mfma_f32_32x32x16_bf16` with fp32
// accumulators — 16x the matrix rate of the fp32 path, so the kernel is bound by staging and HBM,
// not by the MFMA pipe.
//
// Implicit GEMM D[co][pixel] = sum_k W[co][k] X[k][pixel], K = (input channel, tap).  Workgroup =
// 32x8 output pixels x 64 output channels (32x16 x 32 when cout is an odd multiple of 32); per 16-channel chunk the input patch sits in LDS as
// eight planes of channel PAIRS (one dword = bf16(ci), bf16(ci+1) of one pixel), so the B operand of
// a lane (one pixel, eight consecutive channels) is four conflict-free ds_read_b32 and staging is
// one ds_write_b128 per four pixels of a pair plane; weights are pre-packed once per model to
// [chunk][tap][cout][16] bf16 so the A operand (one output channel, eight channels) is one 16-byte
// LDS read.
#include "lf_common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kTW = 32;            // output tile width; height = 4 waves x NB rows
constexpr int kPW = 40;           // patch row pitch in dwords: columns x0-4 .. x0+35
constexpr int kKC = 2;             // 16-channel k-steps per staged chunk (32 input channels)
constexpr int kMaxPrologueCin = 512;   // input channels a fused prologue (scale, shift) may have

struct Bf16ConvArgs {
    const void* x;   // fp32 or bf16 NCHW (template XBF)
    const uint16_t* wprep;
    void* y;         // fp32 or bf16 NCHW (template YBF)
    int n, cin, h, w, cout, chunks, chunks16;   // staged chunks (32 channels); 16-channel slices of wprep
    const float* in_scale;
    const float* in_shift;
    int in_relu;
    const float* out_scale;  // optional epilogue on the fp32 accumulators: v*out_scale[co]+out_shift[co],
    const float* out_shift;  // then ReLU if out_relu (inference: the layer's folded BatchNorm + ReLU)
    int out_relu;
    // ---- training epilogue (template TR; the output is bf16 and what is summed is the ROUNDED
    // value, i.e. exactly what later kernels read back):
    int accumulate;            // y = bf16(conv + y_old) (input-gradient of a block with two consumers)
    // per (output channel, workgroup tile) sums -> stat_part[(co * stat_tiles + tile) * 2 + {0,1}]:
    //   stat_mask_y == null: BatchNorm FORWARD statistics {sum d, sum d*d}, d = y - stat_pivot[co];
    //   else BatchNorm-BACKWARD sums of the BN this gradient feeds: d = y * [mask_y*mask_scale[co] +
    //   mask_shift[co] > 0 or !mask_relu] -> {sum d, sum d*mask_y}
    float* stat_part;
    const float* stat_pivot;   // may be null (pivot 0)
    long long stat_tiles;      // n * tiles per image
    const uint16_t* stat_mask_y;
    const float* mask_scale;
    const float* mask_shift;
    int mask_relu;
};

// sum over the 32 lanes of each wave half (DPP); the total lands in lane 31 / 63
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

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v;
    v.x = (__bf16)lo;  // round to nearest even
    v.y = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}

__device__ __forceinline__ float bf16_up(unsigned bits16) { return __uint_as_float(bits16 << 16); }

__device__ __forceinline__ uint16_t bf16_down(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }

// (three or four workgroups per CU would need <= 168 / 128 registers: the spills cost more than the
// occupancy brings — 40.4 k and 24.5 k img/s against 47.3 k for the whole forward pass)
template <int TAPS, int NCO, int NB, bool XBF, bool YBF, bool TR = false, bool WIDE = false>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_bf16_kernel(Bf16ConvArgs p) {
    static_assert(!WIDE || (YBF && NCO == 2 && NB == 2), "the 16-byte epilogue: bf16 output, 32x8 tile");
    static_assert(!TR || YBF, "the training epilogue stores bf16");
    constexpr int kTH = 4 * NB;
    constexpr int R = TAPS == 9 ? 1 : 0, PH = kTH + 2 * R, KS = TAPS == 9 ? 3 : 1;
    // one LDS buffer: input patch | weight slice; the training epilogue's transpose buffer (fp32
    // [32 channels][256 pixels]) reuses it once the chunk loop is done
    constexpr int KC = kKC;   // 16-channel k-steps staged per barrier pair
    constexpr int kPatchB = KC * 8 * PH * kPW * 4, kWlB = KC * TAPS * NCO * 32 * 2 * 16;
    constexpr bool kWide = WIDE;   // bf16 output, 32x8 tile, w % 8 == 0: 16-byte epilogue through LDS
    constexpr int kSmemB = (kWide && kPatchB + kWlB < 32 * 256 * 4) ? 32 * 256 * 4 : kPatchB + kWlB;
    __shared__ __attribute__((aligned(16))) unsigned char smem[kSmemB];
    uint32_t (*patch)[PH][kPW] = reinterpret_cast<uint32_t (*)[PH][kPW]>(smem);
    lf::u32x4* wl = reinterpret_cast<lf::u32x4*>(smem + kPatchB);
    __shared__ float eps[2][NCO * 32];  // epilogue scale / shift of this workgroup's output channels
    // The prologue's per-channel scale / shift, read from LDS while staging: fetched from global memory there
    // (one dependent round trip per staged channel pair, between the two barriers of every chunk) they were
    // what the kernel waited for.
    __shared__ float lsc[2][kMaxPrologueCin];
    // XCD-aware order (see lf_conv.hip): XCD k walks the k-th contiguous share of the
    // (tile, channel group, image) space, so tiles that share halo rows meet in one L2
    const unsigned gxy = gridDim.x * gridDim.y, gtotal = gxy * gridDim.z;
    const unsigned bflat = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned xk = bflat & 7u, xfloor = gtotal >> 3, xrem = gtotal & 7u;
    const unsigned wflat = xk * xfloor + (xk < xrem ? xk : xrem) + (bflat >> 3);
    const int n = (int)(wflat / gxy);
    const int cog = (int)((wflat - (unsigned)n * gxy) / gridDim.x);
    const int tile = (int)(wflat - (unsigned)n * gxy - (unsigned)cog * gridDim.x);
    const int tiles_x = (p.w + kTW - 1) / kTW;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int x0 = tx * kTW, y0 = ty * kTH;
    const int co0 = cog * (NCO * 32);
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, px = lane & 31, half = lane >> 5;
    const size_t hw = (size_t)p.h * p.w;

    f32x16 acc[NB][NCO];
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b < NCO; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    // Staging is software-pipelined through registers: the global loads of chunk c+1 are issued before
    // the MFMAs of chunk c and land in LDS after them.  A chunk is 32 input channels (two MFMA k-steps per
    // tap): measured on 128->128 @56, batch 256 (scripts/microbench/conv_modes.py, ablation builds), the
    // matrix phase alone takes 212 us and the loads + LDS stores alone 190 us, but with 16-channel chunks
    // (31 KB in flight per workgroup, two workgroups per CU, one MFMA phase of ~0.5 us to land in) the two
    // added up to 547 us: the L2 -> CU path needs more bytes in flight, for longer, to run at its rate.
    // The loads are buffer loads: an offset past the end of the resource returns zero WITHOUT a memory
    // access, which gives the image border, the channel padding and the chunk past the last for free and
    // keeps every load unconditional.
    constexpr int kPatchItems = KC * 8 * PH * (kPW / 4);
    constexpr int NP = (kPatchItems + kThreads - 1) / kThreads;
    constexpr int kWItems = KC * TAPS * NCO * 64;
    constexpr int NW = (kWItems + kThreads - 1) / kThreads;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    struct Raw {  // four pixels of one channel as loaded
        f32x4v f;
        u32x2 h;
    };
    Raw ra[NP], rb[NP];
    lf::u32x4 rw[NW];
    constexpr unsigned kEl = XBF ? 2u : 4u;   // bytes per input element
    const unsigned plane_b = (unsigned)hw * kEl;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(static_cast<const unsigned char*>(p.x)) + (size_t)n * p.cin * hw * kEl, 0,
        (unsigned)p.cin * plane_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(p.wprep), 0, (unsigned)((size_t)p.chunks16 * TAPS * p.cout * 32), 0x00020000);

    // Per-slot geometry does not depend on the chunk.  A slot's flat index (pair plane, patch row,
    // column group) is also its LDS position / 4, so only the global side needs a register: the
    // byte offset of its four pixels inside a channel plane, or kOutside (past every resource).
    constexpr unsigned kOutside = 0x80000000u;
    unsigned g_off[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int it = tid + k * kThreads;
        const int pl = it / (PH * (kPW / 4)), rem = it - pl * (PH * (kPW / 4));
        const int row = rem / (kPW / 4), q = rem - row * (kPW / 4);
        const int gy = y0 - R + row, gx = x0 - 4 + 4 * q;
        // w % 4 == 0: a group of four columns is inside the image or outside it as a whole
        const bool inside = it < kPatchItems && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
        g_off[k] = inside ? ((unsigned)gy * (unsigned)p.w + (unsigned)gx) * kEl : kOutside;
    }
    auto plane_of = [&](int k) { return (tid + k * kThreads) / (PH * (kPW / 4)); };
    auto load_raw = [&](Raw& r, unsigned off) {
        if (XBF)
            r.h = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(xr, off, 0, 0));
        else
            r.f = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
    };
    const unsigned wchunk_b = (unsigned)(TAPS * p.cout * 32);   // one 16-channel slice of the packed weights
    auto issue = [&](int c) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            // channel >= cin (padding of the last chunk, or the chunk past the last): past the resource
            const unsigned o = (unsigned)(c * (16 * KC) + 2 * plane_of(k)) * plane_b + g_off[k];
            load_raw(ra[k], g_off[k] == kOutside ? kOutside : o);
            load_raw(rb[k], g_off[k] == kOutside ? kOutside : o + plane_b);
        }
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int it = tid + k * kThreads;
            const int kk = it / (TAPS * NCO * 64), r1 = it - kk * (TAPS * NCO * 64);
            const int tap = r1 / (NCO * 64), rem = r1 - tap * (NCO * 64);
            const unsigned o = (unsigned)(c * KC + kk) * wchunk_b +
                               (unsigned)((tap * p.cout + co0 + (rem >> 1)) * 32 + 16 * (rem & 1));
            rw[k] = __builtin_bit_cast(lf::u32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, it < kWItems ? o : kOutside, 0, 0));
        }
    };
    auto widen = [&](const Raw& r, int ci) -> f32x4v {  // fp32 values with the fused prologue applied
        f32x4v v;
        if (XBF) {
            v[0] = bf16_up(r.h.x & 0xffffu);
            v[1] = bf16_up(r.h.x >> 16);
            v[2] = bf16_up(r.h.y & 0xffffu);
            v[3] = bf16_up(r.h.y >> 16);
        } else {
            v = r.f;
        }
        if (p.in_scale) {
            const float sc = lsc[0][ci], sh = lsc[1][ci];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], sc, sh);
        }
        if (p.in_relu)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
        return v;
    };
    const bool passthrough = XBF && p.in_scale == nullptr && !p.in_relu;  // bf16 in, no prologue
    uint32_t* patch_flat = &patch[0][0][0];
    auto commit = [&](int c) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            if (tid + k * kThreads >= kPatchItems) continue;
            const int ci0 = c * (16 * KC) + 2 * plane_of(k);
            lf::u32x4 o;
            if (passthrough) {
                // the stored values are the operands (what was not loaded is exactly zero): interleave the
                // two channels' bf16 pixels
                const u32x2 a = ra[k].h, b = rb[k].h;
                o.x = (a.x & 0xffffu) | (b.x << 16);
                o.y = (a.x >> 16) | (b.x & 0xffff0000u);
                o.z = (a.y & 0xffffu) | (b.y << 16);
                o.w = (a.y >> 16) | (b.y & 0xffff0000u);
            } else {
                const bool inside = g_off[k] != kOutside;   // the prologue must not touch the padding
                f32x4v a = {0.0f, 0.0f, 0.0f, 0.0f}, b = {0.0f, 0.0f, 0.0f, 0.0f};
                if (inside && ci0 < p.cin) a = widen(ra[k], ci0);
                if (inside && ci0 + 1 < p.cin) b = widen(rb[k], ci0 + 1);
                o.x = pack_bf16(a[0], b[0]);
                o.y = pack_bf16(a[1], b[1]);
                o.z = pack_bf16(a[2], b[2]);
                o.w = pack_bf16(a[3], b[3]);
            }
            *reinterpret_cast<lf::u32x4*>(patch_flat + 4 * (tid + k * kThreads)) = o;
        }
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int it = tid + k * kThreads;
            if (it >= kWItems) break;
            wl[it] = rw[k];
        }
    };

    if (p.in_scale)   // read back after the first barrier of the chunk loop
        for (int c = tid; c < p.cin; c += kThreads) {
            lsc[0][c] = p.in_scale[c];
            lsc[1][c] = p.in_shift[c];
        }
    if (p.out_scale && tid < NCO * 32) {  // read back after the chunk loop's barriers
        eps[0][tid] = p.out_scale[co0 + tid];
        eps[1][tid] = p.out_shift[co0 + tid];
    }
    // training, 32x8 tile, 16-byte rows: the epilogue works on (8 consecutive pixels, one channel)
    // per thread — its read-modify-write operands are requested now and arrive during the MFMAs
    typedef unsigned uvec4 __attribute__((ext_vector_type(4)));
    const bool wide = kWide;
    const int eg = tid & 31, ec = tid >> 5;
    const int egy = y0 + (eg >> 2), egx = x0 + 8 * (eg & 3);
    const bool eok = egy < p.h && egx < p.w;
    const size_t epo = eok ? (size_t)egy * p.w + egx : 0;
    uvec4 rold[(kWide && TR) ? NCO : 1][4], rmask[(kWide && TR) ? NCO : 1][4];
    auto request_rmw = [&]() {
        if (!TR) return;  // inference: nothing is read back
        const uint16_t* yold = static_cast<const uint16_t*>(p.y) + (size_t)n * p.cout * hw;
        if (p.accumulate)
#pragma unroll
            for (int cb = 0; cb < ((kWide && TR) ? NCO : 1); ++cb)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    rold[cb][j] = *reinterpret_cast<const uvec4*>(yold + (size_t)(co0 + cb * 32 + 8 * j + ec) * hw + epo);
        if (p.stat_mask_y != nullptr) {
            const uint16_t* ym = p.stat_mask_y + (size_t)n * p.cout * hw;
#pragma unroll
            for (int cb = 0; cb < ((kWide && TR) ? NCO : 1); ++cb)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    rmask[cb][j] = *reinterpret_cast<const uvec4*>(ym + (size_t)(co0 + cb * 32 + 8 * j + ec) * hw + epo);
        }
    };
    if (wide) request_rmw();
    issue(0);
    for (int c = 0; c < p.chunks; ++c) {
        __syncthreads();  // the previous chunk's LDS reads are done
        commit(c);
        issue(c + 1);     // past the last chunk: no memory access
        __syncthreads();
        // One B operand (a patch row shifted by dx) serves every (output row, dy) pair that reads
        // it: NB + KS - 1 LDS fetches per dx instead of NB * KS.
#pragma unroll
        for (int kd = 0; kd < KC * KS; ++kd) {
            const int kk = kd / KS, dx = kd - kk * KS;
            bf16x8 A[KS][NCO];
#pragma unroll
            for (int dy = 0; dy < KS; ++dy)
#pragma unroll
                for (int cb = 0; cb < NCO; ++cb)
                    A[dy][cb] = __builtin_bit_cast(
                        bf16x8, wl[((((kk * KS + dy) * KS + dx) * NCO + cb) * 32 + px) * 2 + half]);
            const int col = 4 + px + dx - R;
#pragma unroll
            for (int prow = 0; prow < NB + KS - 1; ++prow) {
                const int row = NB * wv + prow;
                lf::u32x4 bv;
                bv.x = patch[8 * kk + 4 * half + 0][row][col];
                bv.y = patch[8 * kk + 4 * half + 1][row][col];
                bv.z = patch[8 * kk + 4 * half + 2][row][col];
                bv.w = patch[8 * kk + 4 * half + 3][row][col];
                const bf16x8 B = __builtin_bit_cast(bf16x8, bv);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int dy = prow - nb;
                    if (dy < 0 || dy >= KS) continue;
#pragma unroll
                    for (int cb = 0; cb < NCO; ++cb)
                        acc[nb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[dy][cb], B, acc[nb][cb], 0, 0, 0);
                }
            }
        }
    }
    // D[row = output channel][col = pixel]: register r of a lane is channel 8*(r/4) + 4*half + r%4
    float* yn = static_cast<float*>(p.y) + (YBF ? 0 : (size_t)n * p.cout * hw);
    uint16_t* yb = static_cast<uint16_t*>(p.y) + (YBF ? (size_t)n * p.cout * hw : 0);
    const int gx = x0 + px;
    const bool stats = p.stat_part != nullptr, masked = p.stat_mask_y != nullptr;
    if (wide) {
        // through LDS: lane (pixel, 16 channels) -> thread (8 pixels, one channel): 16-byte stores,
        // and a channel's whole tile sits in one half-wave (no cross-wave reduction)
        float* le = reinterpret_cast<float*>(smem);
        uint16_t* yout = static_cast<uint16_t*>(p.y) + (size_t)n * p.cout * hw;
        const long long tg = (long long)n * gridDim.x + tile;
#pragma unroll
        for (int cb = 0; cb < NCO; ++cb) {
            __syncthreads();  // every wave is done with the staging LDS / with the previous block
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cl = 8 * (r >> 2) + 4 * half + (r & 3);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) le[cl * 256 + (NB * wv + nb) * 32 + px] = acc[nb][cb][r];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cl = 8 * j + ec, co = co0 + cb * 32 + cl;
                const lf::f32x4 a0 = *reinterpret_cast<const lf::f32x4*>(le + cl * 256 + 8 * eg);
                const lf::f32x4 a1 = *reinterpret_cast<const lf::f32x4*>(le + cl * 256 + 8 * eg + 4);
                float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                if (TR && p.accumulate)
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        v[e] += bf16_up(rold[TR ? cb : 0][j][e / 2] & 0xffffu);
                        v[e + 1] += bf16_up(rold[TR ? cb : 0][j][e / 2] >> 16);
                    }
                if (p.out_scale) {
                    const float osc = eps[0][cb * 32 + cl], osh = eps[1][cb * 32 + cl];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], osc, osh);
                }
                if (p.out_relu)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.0f);
                uvec4 o;
#pragma unroll
                for (int e = 0; e < 8; e += 2) o[e / 2] = pack_bf16(v[e], v[e + 1]);
                if (eok) *reinterpret_cast<uvec4*>(yout + (size_t)co * hw + epo) = o;
                if (!TR || !stats) continue;
                float a = 0.f, b = 0.f;
                if (eok) {
                    if (!masked) {
                        const float pv = p.stat_pivot != nullptr ? p.stat_pivot[co] : 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float d = bf16_up((e & 1) ? o[e / 2] >> 16 : o[e / 2] & 0xffffu) - pv;
                            a += d;
                            b = fmaf(d, d, b);
                        }
                    } else {
                        const float msc = p.mask_scale[co], msh = p.mask_shift[co];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float rv = bf16_up((e & 1) ? o[e / 2] >> 16 : o[e / 2] & 0xffffu);
                            const unsigned mw = rmask[TR ? cb : 0][j][e / 2];
                            const float yv = bf16_up((e & 1) ? mw >> 16 : mw & 0xffffu);
                            const float d = (!p.mask_relu || fmaf(yv, msc, msh) > 0.f) ? rv : 0.f;
                            a += d;
                            b = fmaf(d, yv, b);
                        }
                    }
                }
                a = half_sum32(a);
                b = half_sum32(b);
                if (eg == 31) {
                    float* dst = p.stat_part + ((size_t)co * (size_t)p.stat_tiles + (size_t)tg) * 2;
                    dst[0] = a;
                    dst[1] = b;
                }
            }
        }
        return;
    }
    if (!TR) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int gy = y0 + NB * wv + nb;
            if (gy >= p.h || gx >= p.w) continue;
#pragma unroll
            for (int cb = 0; cb < NCO; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int col = cb * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
                    const int co = co0 + col;
                    const size_t o = (size_t)co * hw + (size_t)gy * p.w + gx;
                    float v = acc[nb][cb][r];
                    if (p.out_scale) v = fmaf(v, eps[0][col], eps[1][col]);
                    if (p.out_relu) v = fmaxf(v, 0.0f);
                    if (YBF)
                        yb[o] = bf16_down(v);
                    else
                        yn[o] = v;
                }
        }
        return;
    }
    // ---- training: bf16 store (optionally on top of the old value) + per-tile channel sums of
    // the rounded values

    const uint16_t* my = masked ? p.stat_mask_y + (size_t)n * p.cout * hw : nullptr;
    float* red = reinterpret_cast<float*>(smem);  // [4 waves][NCO*32][2]
    static_assert(4 * NCO * 32 * 2 <= 8 * PH * kPW, "statistics scratch must fit the patch LDS");
    if (stats) __syncthreads();  // every wave is done with the staging LDS
    bool ok[NB];
    unsigned po[NB];  // pixel offset inside a channel plane (0 when outside the image)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int gy = y0 + NB * wv + nb;
        ok[nb] = gy < p.h && gx < p.w;
        po[nb] = ok[nb] ? (unsigned)gy * (unsigned)p.w + (unsigned)gx : 0u;
    }
#pragma unroll
    for (int cb = 0; cb < NCO; ++cb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int col = cb * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
            const int co = co0 + col;
            const size_t cbase = (size_t)co * hw;
            float oldv[NB], yv[NB];
            if (p.accumulate)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) oldv[nb] = bf16_up(yb[cbase + po[nb]]);
            if (masked)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) yv[nb] = bf16_up(my[cbase + po[nb]]);
            float s1 = 0.f, s2 = 0.f;
            const float pv = (stats && !masked && p.stat_pivot != nullptr) ? p.stat_pivot[co] : 0.f;
            const float msc = masked ? p.mask_scale[co] : 0.f, msh = masked ? p.mask_shift[co] : 0.f;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                float v = acc[nb][cb][r];
                if (p.accumulate) v += oldv[nb];
                if (p.out_scale) v = fmaf(v, eps[0][col], eps[1][col]);   // inference with sums: the folded BatchNorm
                if (p.out_relu) v = fmaxf(v, 0.0f);
                const uint16_t vb = bf16_down(v);
                if (ok[nb]) yb[cbase + po[nb]] = vb;
                if (!stats) continue;
                const float vr = bf16_up(vb);
                if (!masked) {
                    const float d = ok[nb] ? vr - pv : 0.f;
                    s1 += d;
                    s2 = fmaf(d, d, s2);
                } else {
                    const bool on = ok[nb] && (!p.mask_relu || fmaf(yv[nb], msc, msh) > 0.f);
                    const float d = on ? vr : 0.f;
                    s1 += d;
                    s2 = fmaf(d, yv[nb], s2);
                }
            }
            if (stats) {
                s1 = half_sum32(s1);
                s2 = half_sum32(s2);
                if (px == 31) {
                    red[(wv * (NCO * 32) + col) * 2] = s1;
                    red[(wv * (NCO * 32) + col) * 2 + 1] = s2;
                }
            }
        }
    }
    if (stats) {
        __syncthreads();
        const long long tg = (long long)n * gridDim.x + tile;
        if (tid < NCO * 32) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) {
                a += red[(w4 * (NCO * 32) + tid) * 2];
                b += red[(w4 * (NCO * 32) + tid) * 2 + 1];
            }
            float* dst = p.stat_part + ((size_t)(co0 + tid) * (size_t)p.stat_tiles + (size_t)tg) * 2;
            dst[0] = a;
            dst[1] = b;
        }
    }
}

// --- the plane kernels of the forward pass on bf16 activations (inference only) ---------------
// Same arithmetic as gap_kernel / tail_fwd_kernel of lf_nn.hip (fp32 after widening the operands);
// what differs is the storage type, and that nothing is kept for a backward pass.

// out[n][c] = mean over the plane of relu?(x * scale[c] + shift[c]); one workgroup per plane.
__global__ __launch_bounds__(kThreads) void gap_bf16_kernel(const uint16_t* __restrict__ x,
                                                            float* __restrict__ out, int hw, int c,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int relu) {
    __shared__ float red[kThreads / 64];
    const size_t base = (size_t)blockIdx.x * hw;
    const bool pro = scale != nullptr;
    const float sc = pro ? scale[blockIdx.x % c] : 1.f, sh = pro ? shift[blockIdx.x % c] : 0.f;
    auto one = [&](unsigned bits) {
        float v = bf16_up(bits);
        if (pro) v = fmaf(v, sc, sh);
        return relu ? fmaxf(v, 0.f) : v;
    };
    float acc = 0.f;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2* x4 = reinterpret_cast<const u32x2*>(x + base);  // hw % 4 == 0
    for (int i = threadIdx.x; i < hw / 4; i += kThreads) {
        const u32x2 v = x4[i];
        acc += (one(v.x & 0xffffu) + one(v.x >> 16)) + (one(v.y & 0xffffu) + one(v.y >> 16));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)hw;
}

struct TailBf16Args {
    const uint16_t* y;       // second conv of the block (raw, before its BatchNorm)
    const float* a_scale;    // its folded BatchNorm
    const float* a_shift;
    const float* s;          // SE gate [n][c] or null
    const uint16_t* sc;      // shortcut tensor
    const float* sc_scale;   // its scale / shift (+ReLU) or null when it is final already
    const float* sc_shift;
    int sc_relu;
    int c, h, w;
};

// pooled = maxpool2x2(relu(shortcut + relu(BN(y)) * gate)), bf16 in, bf16 out; w % 4 == 0.
__global__ __launch_bounds__(kThreads) void tail_fwd_bf16_kernel(TailBf16Args t, uint16_t* __restrict__ p) {
    const int plane = blockIdx.x, ch = plane % t.c;
    const float sv = t.s ? t.s[plane] : 1.f;
    const float as = t.a_scale ? t.a_scale[ch] : 1.f, ab = t.a_scale ? t.a_shift[ch] : 0.f;
    const float ks = t.sc_scale ? t.sc_scale[ch] : 1.f, kb = t.sc_scale ? t.sc_shift[ch] : 0.f;
    const int h = t.h, w = t.w, ph = h / 2, pw = w / 2, pw2 = pw / 2;
    const size_t base = (size_t)plane * h * w, pbase = (size_t)plane * ph * pw;
    auto r = [&](unsigned yb, unsigned sb) {
        const float a = t.a_scale ? fmaxf(fmaf(bf16_up(yb), as, ab), 0.f) : bf16_up(yb);
        float shv = bf16_up(sb);
        if (t.sc_scale) {
            shv = fmaf(shv, ks, kb);
            if (t.sc_relu) shv = fmaxf(shv, 0.f);
        }
        return fmaxf(shv + a * sv, 0.f);
    };
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    for (int q = blockIdx.y * kThreads + threadIdx.x; q < ph * pw2; q += gridDim.y * kThreads) {
        const int py = q / pw2, px2 = q - py * pw2;
        float m0 = 0.f, m1 = 0.f;  // the block's output is >= 0: starting the max at 0 is exact
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const size_t o = base + (size_t)(2 * py + dy) * w + 4 * px2;
            const u32x2 yv = *reinterpret_cast<const u32x2*>(t.y + o);
            const u32x2 sv4 = *reinterpret_cast<const u32x2*>(t.sc + o);
            m0 = fmaxf(m0, fmaxf(r(yv.x & 0xffffu, sv4.x & 0xffffu), r(yv.x >> 16, sv4.x >> 16)));
            m1 = fmaxf(m1, fmaxf(r(yv.y & 0xffffu, sv4.y & 0xffffu), r(yv.y >> 16, sv4.y >> 16)));
        }
        *reinterpret_cast<unsigned*>(p + pbase + (size_t)py * pw + 2 * px2) =
            (unsigned)bf16_down(m0) | (unsigned)bf16_down(m1) << 16;
    }
}

// fp32 [cin][taps][cout] -> bf16 [chunk][tap][cout][16] (channels past cin are zero)
__global__ void prep_weights_bf16_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int cin,
                                         int taps, int cout, int chunks) {
    const size_t total = (size_t)chunks * taps * cout * 16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 15);
        const size_t rest = i >> 4;
        const int co = (int)(rest % cout);
        const size_t rest2 = rest / cout;
        const int tap = (int)(rest2 % taps), c = (int)(rest2 / taps);
        const int ci = c * 16 + j;
        const float v = ci < cin ? w[((size_t)ci * taps + tap) * cout + co] : 0.0f;
        out[i] = __builtin_bit_cast(uint16_t, (__bf16)v);
    }
}

}  // namespace

namespace {

// cout % 64 == 0: two 32-channel blocks per workgroup share one staged 32x8 patch (four measured
// the same); cout == 32 (+64k): one block, and a 32x16 tile instead so that a staged patch
// still feeds 16 accumulator tiles per wave
inline int bf16_nco(int cout) { return cout % 64 == 0 ? 2 : 1; }
inline int bf16_tiles(int h, int w, int cout) {
    const int th = bf16_nco(cout) == 2 ? 8 : 16;
    return ((w + kTW - 1) / kTW) * ((h + th - 1) / th);
}

template <bool XBF, bool YBF, bool TR>
void launch_conv_bf16(const Bf16ConvArgs& a, int ksize, hipStream_t s) {
    const int nco = bf16_nco(a.cout);
    dim3 grid(bf16_tiles(a.h, a.w, a.cout), a.cout / (32 * nco), a.n);
    constexpr bool W = YBF;
    const bool wide = YBF && nco == 2 && a.w % 8 == 0;  // rows of 16 bytes: the epilogue through LDS
    if (ksize == 3) {
        if (nco == 2) {
            if (wide) conv_bf16_kernel<9, 2, 2, XBF, YBF, TR, W><<<grid, kThreads, 0, s>>>(a);
            else conv_bf16_kernel<9, 2, 2, XBF, YBF, TR, false><<<grid, kThreads, 0, s>>>(a);
        } else {
            conv_bf16_kernel<9, 1, 4, XBF, YBF, TR, false><<<grid, kThreads, 0, s>>>(a);
        }
    } else {
        if (nco == 2) {
            if (wide) conv_bf16_kernel<1, 2, 2, XBF, YBF, TR, W><<<grid, kThreads, 0, s>>>(a);
            else conv_bf16_kernel<1, 2, 2, XBF, YBF, TR, false><<<grid, kThreads, 0, s>>>(a);
        } else {
            conv_bf16_kernel<1, 1, 4, XBF, YBF, TR, false><<<grid, kThreads, 0, s>>>(a);
        }
    }
}

// means[n][c] = scale * sum over the image's partial sums: `unit` layout part[(n * units + u) * c_total + c] (the
// streaming kernel's segments) or `tile` layout part[((c * n_total + n) * units + u) * 2] (the K-chunked kernel's
// per-tile statistics, sum in element 0).  Fixed order: deterministic.
__global__ void partial_sums_mean_kernel(const float* __restrict__ part, float* __restrict__ means, int n, int c,
                                         int units, int tile_layout, float scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * c) return;
    const int im = i / c, ch = i - im * c;
    float acc = 0.f;
    for (int u = 0; u < units; ++u)
        acc += tile_layout ? part[(((size_t)ch * n + im) * units + u) * 2] : part[((size_t)im * units + u) * c + ch];
    means[i] = acc * scale;
}

}  // namespace

extern "C" {

size_t lf_conv2d_bf16_weight_elems(int cin, int cout, int ksize) {
    if (cin <= 0 || cout <= 0 || (ksize != 1 && ksize != 3)) return 0;
    return (size_t)((cin + 15) / 16) * ksize * ksize * cout * 16;
}

int lf_conv2d_bf16_prep_weights(const float* w_iko, uint16_t* wprep, int cin, int cout, int ksize,
                                lf_stream_t stream) {
    LF_REQUIRE(w_iko && wprep, "lf_conv2d_bf16_prep_weights: null buffer");
    LF_REQUIRE(cin > 0 && cout > 0 && (ksize == 1 || ksize == 3), "lf_conv2d_bf16_prep_weights: bad dims");
    const int chunks = (cin + 15) / 16;
    const size_t total = lf_conv2d_bf16_weight_elems(cin, cout, ksize);
    prep_weights_bf16_kernel<<<lf::stream_grid(total, 256), 256, 0, lf::as_stream(stream)>>>(
        w_iko, wprep, cin, ksize * ksize, cout, chunks);
    return lf::check_launch("lf_conv2d_bf16_prep_weights");
}

int lf_conv2d_bf16_act(const void* x, int x_bf16, const uint16_t* wprep, void* y, int y_bf16, int n, int cin,
                       int h, int w, int cout, int ksize, const float* in_scale, const float* in_shift,
                       int in_relu, const float* out_scale, const float* out_shift, int out_relu,
                       lf_stream_t stream) {
    LF_REQUIRE(x && wprep && y, "lf_conv2d_bf16: null buffer");
    LF_REQUIRE(n > 0 && cin > 0 && h > 0 && w > 0 && cout > 0, "lf_conv2d_bf16: bad dims");
    LF_REQUIRE(ksize == 1 || ksize == 3, "lf_conv2d_bf16: ksize must be 1 or 3");
    LF_REQUIRE(w % 4 == 0, "lf_conv2d_bf16: width must be a multiple of 4 (got %d)", w);
    LF_REQUIRE(cout % 32 == 0, "lf_conv2d_bf16: cout must be a multiple of 32 (got %d)", cout);
    LF_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "lf_conv2d_bf16: scale/shift must both be set");
    LF_REQUIRE(in_scale == nullptr || cin <= kMaxPrologueCin,
               "lf_conv2d_bf16: a fused prologue takes at most %d input channels (got %d)", kMaxPrologueCin, cin);
    LF_REQUIRE((out_scale == nullptr) == (out_shift == nullptr),
               "lf_conv2d_bf16: out_scale/out_shift must both be set");
    // 32-bit buffer offsets: one image's input (plus the two staged chunks past its end that the pipeline asks
    // for and gets zeros back) and the packed weights must each stay below 2 GiB
    LF_REQUIRE((size_t)(cin + 2 * 16 * kKC) * h * w * (x_bf16 ? 2 : 4) < ((size_t)1 << 31) &&
                   lf_conv2d_bf16_weight_elems(cin, cout, ksize) * 2 < ((size_t)1 << 31),
               "lf_conv2d_bf16: image or weights too large for 32-bit buffer offsets");
    LF_REQUIRE(n <= 65535, "lf_conv2d_bf16: batch too large for grid.z");
    LF_REQUIRE(((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(wprep)) & 15) == 0,
               "lf_conv2d_bf16: x and wprep must be 16-byte aligned");
    hipStream_t s = lf::as_stream(stream);
    if (y_bf16 && cout == 32 && lf::conv_bf16s_parts(n, cin, h, w, cout, ksize, x_bf16) > 0) {
        // the 224x224 stage (stem, 32->32): the streaming kernel (resident filter bank, 16-byte accesses).
        // The 64-channel layers of the 112x112 stage stay on the K-chunked kernel: it was well ahead there
        // without a read-modify-write epilogue (1.7 ms against 3.2 ms at 64->64, batch 1,024) and is level with
        // the streaming kernel since that one got 64x4 tiles and per-XCD tile rows (whole forward pass
        // 52.4 k img/s as routed here, 52.7 k with every covered layer on the streaming kernel).
        lf::ConvBf16TrainArgs t{};
        t.x = x; t.wprep = wprep; t.y = static_cast<uint16_t*>(y); t.n = n; t.cin = cin; t.h = h; t.w = w; t.cout = cout;
        t.in_scale = in_scale; t.in_shift = in_shift; t.in_relu = in_relu;
        t.out_scale = out_scale; t.out_shift = out_shift; t.out_relu = out_relu;
        const int rc = lf::conv_bf16s_launch(t, ksize, x_bf16, s);
        if (rc != LF_OK) return rc;
        return lf::check_launch("lf_conv2d_bf16");
    }
    Bf16ConvArgs a{};
    a.x = x; a.wprep = wprep; a.y = y; a.n = n; a.cin = cin; a.h = h; a.w = w; a.cout = cout;
    a.chunks16 = (cin + 15) / 16;
    a.chunks = (a.chunks16 + kKC - 1) / kKC;
    a.in_scale = in_scale; a.in_shift = in_shift; a.in_relu = in_relu;
    a.out_scale = out_scale; a.out_shift = out_shift; a.out_relu = out_relu;
    if (x_bf16) {
        if (y_bf16) launch_conv_bf16<true, true, false>(a, ksize, s); else launch_conv_bf16<true, false, false>(a, ksize, s);
    } else {
        if (y_bf16) launch_conv_bf16<false, true, false>(a, ksize, s); else launch_conv_bf16<false, false, false>(a, ksize, s);
    }
    return lf::check_launch("lf_conv2d_bf16");
}

long long lf_conv2d_bf16_stats_tiles(int n, int cin, int h, int w, int cout, int ksize, int x_bf16) {
    if (n <= 0 || cin <= 0 || h <= 0 || w <= 0 || cout <= 0) return 0;
    const long long parts = lf::conv_bf16s_parts(n, cin, h, w, cout, ksize, x_bf16);
    return parts > 0 ? parts : (long long)n * bf16_tiles(h, w, cout);
}

int lf_conv2d_bf16_train(const void* x, int x_bf16, const uint16_t* wprep, uint16_t* y, int n, int cin, int h,
                         int w, int cout, int ksize, const float* in_scale, const float* in_shift, int in_relu,
                         int accumulate, float* tile_part, size_t tile_part_bytes, const float* pivot,
                         const uint16_t* mask_y, const float* mask_scale, const float* mask_shift,
                         int mask_relu, lf_stream_t stream) {
    LF_REQUIRE(x && wprep && y, "lf_conv2d_bf16_train: null buffer");
    LF_REQUIRE(n > 0 && cin > 0 && h > 0 && w > 0 && cout > 0, "lf_conv2d_bf16_train: bad dims");
    LF_REQUIRE(ksize == 1 || ksize == 3, "lf_conv2d_bf16_train: ksize must be 1 or 3");
    LF_REQUIRE(w % 4 == 0, "lf_conv2d_bf16_train: width must be a multiple of 4 (got %d)", w);
    LF_REQUIRE(cout % 32 == 0, "lf_conv2d_bf16_train: cout must be a multiple of 32 (got %d)", cout);
    LF_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "lf_conv2d_bf16_train: scale/shift must both be set");
    LF_REQUIRE(in_scale == nullptr || cin <= kMaxPrologueCin,
               "lf_conv2d_bf16_train: a fused prologue takes at most %d input channels (got %d)", kMaxPrologueCin, cin);
    LF_REQUIRE((size_t)(cin + 2 * 16 * kKC) * h * w * (x_bf16 ? 2 : 4) < ((size_t)1 << 31) &&
                   lf_conv2d_bf16_weight_elems(cin, cout, ksize) * 2 < ((size_t)1 << 31),
               "lf_conv2d_bf16_train: image or weights too large for 32-bit buffer offsets");
    LF_REQUIRE(n <= 65535, "lf_conv2d_bf16_train: batch too large for grid.z");
    LF_REQUIRE(((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(wprep)) & 15) == 0,
               "lf_conv2d_bf16_train: x and wprep must be 16-byte aligned");
    LF_REQUIRE(mask_y == nullptr || (tile_part && mask_scale && mask_shift),
               "lf_conv2d_bf16_train: mask_y needs tile_part and mask_scale / mask_shift");
    const long long tiles = lf_conv2d_bf16_stats_tiles(n, cin, h, w, cout, ksize, x_bf16);
    if (tile_part != nullptr) {
        const size_t need = (size_t)tiles * (size_t)cout * 2 * sizeof(float);
        if (tile_part_bytes < need) {
            lf::set_error("lf_conv2d_bf16_train: tile_part %zu bytes < %zu", tile_part_bytes, need);
            return LF_ERR_WORKSPACE;
        }
    }
    hipStream_t s = lf::as_stream(stream);
    if (lf::conv_bf16s_parts(n, cin, h, w, cout, ksize, x_bf16) > 0) {
        lf::ConvBf16TrainArgs t{};
        t.x = x; t.wprep = wprep; t.y = y; t.n = n; t.cin = cin; t.h = h; t.w = w; t.cout = cout;
        t.in_scale = in_scale; t.in_shift = in_shift; t.in_relu = in_relu; t.accumulate = accumulate;
        t.stat_part = tile_part; t.stat_pivot = pivot; t.stat_mask_y = mask_y;
        t.mask_scale = mask_scale; t.mask_shift = mask_shift; t.mask_relu = mask_relu;
        const int rc = lf::conv_bf16s_launch(t, ksize, x_bf16, s);
        if (rc != LF_OK) return rc;
        return lf::check_launch("lf_conv2d_bf16_train");
    }
    Bf16ConvArgs a{};
    a.x = x; a.wprep = wprep; a.y = y; a.n = n; a.cin = cin; a.h = h; a.w = w; a.cout = cout;
    a.chunks16 = (cin + 15) / 16;
    a.chunks = (a.chunks16 + kKC - 1) / kKC;
    a.in_scale = in_scale; a.in_shift = in_shift; a.in_relu = in_relu;
    a.accumulate = accumulate;
    a.stat_part = tile_part; a.stat_pivot = pivot;
    a.stat_tiles = tiles;
    a.stat_mask_y = mask_y; a.mask_scale = mask_scale; a.mask_shift = mask_shift; a.mask_relu = mask_relu;
    if (x_bf16) launch_conv_bf16<true, true, true>(a, ksize, s); else launch_conv_bf16<false, true, true>(a, ksize, s);
    return lf::check_launch("lf_conv2d_bf16_train");
}

size_t lf_conv2d_bf16_act_mean_workspace(int n, int cin, int h, int w, int cout, int ksize, int x_bf16) {
    if (n <= 0 || cin <= 0 || h <= 0 || w <= 0 || cout <= 0) return 0;
    // (the same routing as lf_conv2d_bf16_act: the streaming kernel for the 32-channel layers only, so that the
    // stored activation is bit-equal with and without the means)
    const int units = cout == 32 ? lf::conv_bf16s_units_per_image(n, cin, h, w, cout, ksize, x_bf16) : 0;
    if (units > 0) return (size_t)n * units * cout * sizeof(float);
    return (size_t)n * bf16_tiles(h, w, cout) * cout * 2 * sizeof(float);
}

int lf_conv2d_bf16_act_mean(const void* x, int x_bf16, const uint16_t* wprep, uint16_t* y, int n, int cin, int h, int w,
                            int cout, int ksize, const float* in_scale, const float* in_shift, int in_relu,
                            const float* out_scale, const float* out_shift, int out_relu, float* means,
                            void* workspace, size_t ws_bytes, lf_stream_t stream) {
    LF_REQUIRE(x && wprep && y && means && workspace, "lf_conv2d_bf16_act_mean: null buffer");
    LF_REQUIRE(n > 0 && cin > 0 && h > 0 && w > 0 && cout > 0, "lf_conv2d_bf16_act_mean: bad dims");
    LF_REQUIRE(ksize == 1 || ksize == 3, "lf_conv2d_bf16_act_mean: ksize must be 1 or 3");
    LF_REQUIRE(w % 4 == 0, "lf_conv2d_bf16_act_mean: width must be a multiple of 4 (got %d)", w);
    LF_REQUIRE(cout % 32 == 0, "lf_conv2d_bf16_act_mean: cout must be a multiple of 32 (got %d)", cout);
    LF_REQUIRE((in_scale == nullptr) == (in_shift == nullptr) && (out_scale == nullptr) == (out_shift == nullptr),
               "lf_conv2d_bf16_act_mean: scale/shift must both be set");
    LF_REQUIRE(in_scale == nullptr || cin <= kMaxPrologueCin,
               "lf_conv2d_bf16_act_mean: a fused prologue takes at most %d input channels (got %d)", kMaxPrologueCin, cin);
    LF_REQUIRE((size_t)(cin + 2 * 16 * kKC) * h * w * (x_bf16 ? 2 : 4) < ((size_t)1 << 31) &&
                   lf_conv2d_bf16_weight_elems(cin, cout, ksize) * 2 < ((size_t)1 << 31),
               "lf_conv2d_bf16_act_mean: image or weights too large for 32-bit buffer offsets");
    LF_REQUIRE(n <= 65535, "lf_conv2d_bf16_act_mean: batch too large for grid.z");
    LF_REQUIRE(((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(wprep)) & 15) == 0,
               "lf_conv2d_bf16_act_mean: x and wprep must be 16-byte aligned");
    const size_t need = lf_conv2d_bf16_act_mean_workspace(n, cin, h, w, cout, ksize, x_bf16);
    if (ws_bytes < need) {
        lf::set_error("lf_conv2d_bf16_act_mean: workspace %zu < %zu bytes", ws_bytes, need);
        return LF_ERR_WORKSPACE;
    }
    hipStream_t s = lf::as_stream(stream);
    float* part = static_cast<float*>(workspace);
    const float inv = 1.0f / (float)((size_t)h * w);
    const int units = cout == 32 ? lf::conv_bf16s_units_per_image(n, cin, h, w, cout, ksize, x_bf16) : 0;
    if (units > 0) {
        lf::ConvBf16TrainArgs t{};
        t.x = x; t.wprep = wprep; t.y = y; t.n = n; t.cin = cin; t.h = h; t.w = w; t.cout = cout;
        t.in_scale = in_scale; t.in_shift = in_shift; t.in_relu = in_relu;
        t.out_scale = out_scale; t.out_shift = out_shift; t.out_relu = out_relu;
        t.unit_sums = part;
        const int rc = lf::conv_bf16s_launch(t, ksize, x_bf16, s);
        if (rc != LF_OK) return rc;
        partial_sums_mean_kernel<<<(n * cout + 255) / 256, 256, 0, s>>>(part, means, n, cout, units, 0, inv);
        return lf::check_launch("lf_conv2d_bf16_act_mean");
    }
    Bf16ConvArgs a{};
    a.x = x; a.wprep = wprep; a.y = y; a.n = n; a.cin = cin; a.h = h; a.w = w; a.cout = cout;
    a.chunks16 = (cin + 15) / 16;
    a.chunks = (a.chunks16 + kKC - 1) / kKC;
    a.in_scale = in_scale; a.in_shift = in_shift; a.in_relu = in_relu;
    a.out_scale = out_scale; a.out_shift = out_shift; a.out_relu = out_relu;
    const int tiles = bf16_tiles(h, w, cout);
    a.stat_part = part; a.stat_pivot = nullptr; a.stat_tiles = (long long)n * tiles;
    if (x_bf16) launch_conv_bf16<true, true, true>(a, ksize, s); else launch_conv_bf16<false, true, true>(a, ksize, s);
    partial_sums_mean_kernel<<<(n * cout + 255) / 256, 256, 0, s>>>(part, means, n, cout, tiles, 1, inv);
    return lf::check_launch("lf_conv2d_bf16_act_mean");
}

int lf_gap_bf16(const uint16_t* x, float* out, int n, int c, int hw, const float* scale, const float* shift,
                int relu, lf_stream_t stream) {
    LF_REQUIRE(x && out, "lf_gap_bf16: null buffer");
    LF_REQUIRE(n > 0 && c > 0 && hw > 0 && hw % 4 == 0, "lf_gap_bf16: bad dims n=%d c=%d hw=%d (hw %% 4 == 0)", n, c,
               hw);
    LF_REQUIRE((scale == nullptr) == (shift == nullptr), "lf_gap_bf16: scale/shift must both be set");
    LF_REQUIRE((reinterpret_cast<size_t>(x) & 7) == 0, "lf_gap_bf16: x must be 8-byte aligned");
    gap_bf16_kernel<<<n * c, kThreads, 0, lf::as_stream(stream)>>>(x, out, hw, c, scale, shift, relu);
    return lf::check_launch("lf_gap_bf16");
}

int lf_block_tail_fwd_bf16(const uint16_t* y, const float* a_scale, const float* a_shift, const float* s,
                           const uint16_t* sc, const float* sc_scale, const float* sc_shift, int sc_relu,
                           uint16_t* pooled, int n, int c, int h, int w, lf_stream_t stream) {
    LF_REQUIRE(y && sc && pooled, "lf_block_tail_fwd_bf16: null buffer");
    LF_REQUIRE((a_scale == nullptr) == (a_shift == nullptr), "lf_block_tail_fwd_bf16: a_scale/a_shift must both be set");
    LF_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 4 == 0,
               "lf_block_tail_fwd_bf16: bad dims n=%d c=%d h=%d w=%d (h even, w %% 4 == 0)", n, c, h, w);
    LF_REQUIRE((sc_scale == nullptr) == (sc_shift == nullptr), "lf_block_tail_fwd_bf16: scale/shift must both be set");
    LF_REQUIRE(((reinterpret_cast<size_t>(y) | reinterpret_cast<size_t>(sc)) & 7) == 0 &&
                   (reinterpret_cast<size_t>(pooled) & 3) == 0,
               "lf_block_tail_fwd_bf16: buffers must be 8-byte (inputs) / 4-byte (output) aligned");
    TailBf16Args t{y, a_scale, a_shift, s, sc, sc_scale, sc_shift, sc_relu, c, h, w};
    const int per_plane = (h / 2) * (w / 4);
    dim3 grid(n * c, (per_plane + kThreads - 1) / kThreads > 8 ? 8 : (per_plane + kThreads - 1) / kThreads);
    tail_fwd_bf16_kernel<<<grid, kThreads, 0, lf::as_stream(stream)>>>(t, pooled);
    return lf::check_launch("lf_block_tail_fwd_bf16");
}

int lf_conv2d_bf16_f32(const float* x, const uint16_t* wprep, float* y, int n, int cin, int h, int w,
                       int cout, int ksize, const float* in_scale, const float* in_shift, int in_relu,
                       lf_stream_t stream) {
    return lf_conv2d_bf16_act(x, 0, wprep, y, 0, n, cin, h, w, cout, ksize, in_scale, in_shift, in_relu, nullptr,
                              nullptr, 0, stream);
}

}  // extern "C"
