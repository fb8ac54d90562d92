// libleafhip — 3x3 weight gradient on bf16 tensors for the 224x224 / 112x112 stages (Cin = 32,
// Cout = 32 or 64), where the tensors are largest and the kernel is bound by HBM latency, not by
// arithmetic: the same transposed-LDS / ds_read_b64_tr_b16 scheme as lf_wgrad_bf16.hip, organised
// as a PRODUCER / CONSUMER workgroup so that the memory pipeline runs two tiles ahead.
//
//   waves 0-3 (consumers): operand fetch + MFMA only; each owns all nine taps of one
//              (sub-block, K share): 144 accumulator registers and no staging state;
//   waves 4-7 (producers): global loads, BatchNorm-backward / BatchNorm+ReLU arithmetic in fp32,
//              f32 -> bf16 packing (channel pairs of one pixel = the transposition), LDS stores and
//              the dY side output.  No accumulators, so a producer thread can hold the raw loads
//              of TWO tiles: the loads of tile i+3 are issued while tile i+1 is transformed, and
//              every load has two whole tile periods to arrive.
//
// One barrier per tile; LDS holds two operand images (the consumers read one while the producers
// fill the other).  With one depth of prefetch the same kernel delivered one tile per HBM round
// trip (2.1 TB/s at 32->32 / 224x224); see DESIGN.md for the measured steps.
// Replaces the weight gradient of Conv2D + the BatchNormalization backward that precedes it
// (srcs/model/cnn.py:27-33 under mixed_float16, train.py:179-190).
#include "lf_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned uvec __attribute__((ext_vector_type(4)));  // eight bf16

constexpr int kNT = 512, kPT = 256;  // threads: 4 consumer + 4 producer waves
constexpr int G = 8;                 // pixels per staging unit (16-byte accesses)

__device__ __forceinline__ float up(unsigned bits16) { return __uint_as_float(bits16 << 16); }
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v;
    v.x = (__bf16)lo;
    v.y = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}
// byte offset of (pixel, channel quad) inside one [pixel][32 channels] image (see lf_wgrad_bf16.hip)
__device__ __forceinline__ unsigned img_off(unsigned pixel, unsigned quad) {
    return pixel * 64u + ((quad ^ ((pixel >> 2) & 7u)) << 3);
}

template <int TW, int TH, int COB>
struct PcShape {
    static constexpr int PW = TW + 2, PH = TH + 2;
    static constexpr int XPIX = PW * PH, DPIX = TW * TH;
    static constexpr int XBYTES = XPIX * 64, DBYTES = COB * DPIX * 64;
    static constexpr int BUF = XBYTES + DBYTES;
    static constexpr int RED = 9 * 16 * 64 * 4;
    static constexpr int LDS = 2 * BUF > RED ? 2 * BUF : RED;
    static constexpr int PGS = TW / G;
    static constexpr int NDU = COB * 8 * TH * PGS, DPT = (NDU + kPT - 1) / kPT;
    static constexpr int NXU = 8 * PH * PGS, XPT = (NXU + kPT - 1) / kPT;
    static constexpr int NHU = 8 * PH * 2, HPT = (NHU + kPT - 1) / kPT;
};

// the raw loads of one tile, as one producer thread holds them
template <int DPT, int XPT, int HPT>
struct Raw {
    uvec g[DPT][4], y[DPT][4], x[XPT][4];
    unsigned h[HPT][2];
    unsigned dmask, xmask, hmask;
};

template <int TW, int TH, int COB>
__global__ __launch_bounds__(kNT, 1) void wgrad_bf16_pc_kernel(lf::WgradBf16Args p) {
    using S = PcShape<TW, TH, COB>;
    static_assert(TW % G == 0 && (TW * TH) % 16 == 0, "tile: whole 8-pixel groups, whole 16-pixel k-steps");
    constexpr int PW = S::PW, PGS = S::PGS;
    constexpr int NQ = COB, KSPL = 4 / NQ;
    constexpr int NS = TW * TH / 16;
    constexpr int DPT = S::DPT, XPT = S::XPT, HPT = S::HPT;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ float lbn[5 * 32 * COB];
    __shared__ float lsc[2 * 32];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool producer = wv >= 4;
    const int ptid = tid - kPT;            // producer thread index 0..255
    const int q = wv % NQ, ks = (wv & 3) / NQ;  // consumer: sub-block (= output block), K share
    const int co0 = blockIdx.z * (32 * COB);
    const size_t hw = (size_t)p.h * p.w;
    const bool bn = p.bn_y != nullptr;
    const bool pro = p.in_scale != nullptr;

    if (bn)
        for (int e = tid; e < 5 * 32 * COB; e += kNT) {
            const int kk = e / (32 * COB), c = e - kk * (32 * COB);
            lbn[e] = p.bn_coef[(size_t)kk * p.cout + co0 + c];
        }
    if (pro)
        for (int c = tid; c < 32; c += kNT) {
            lsc[c] = c < p.cin ? p.in_scale[c] : 1.f;
            lsc[32 + c] = c < p.cin ? p.in_shift[c] : 0.f;
        }

    const int first = blockIdx.x * p.items_per_split;
    const int last = min(first + p.items_per_split, p.items);
    const int count = last - first;
    const int tiles = p.tiles_x * p.tiles_y;
    auto tile_of = [&](int item, int& n, int& tx0, int& ty0) {
        n = item / tiles;
        const int t = item - n * tiles;
        tx0 = (t % p.tiles_x) * TW;
        ty0 = (t / p.tiles_x) * TH;
    };

    typedef Raw<DPT, XPT, HPT> RawT;
    auto issue = [&](int item, RawT& r) {
        int n, tx0, ty0;
        tile_of(item, n, tx0, ty0);
        r.dmask = r.xmask = r.hmask = 0;
        const uint16_t* gn = p.g + (size_t)n * p.cout * hw;
        const uint16_t* yn = bn ? p.bn_y + (size_t)n * p.cout * hw : nullptr;
        const uint16_t* xn = p.x + (size_t)n * p.cin * hw;
#pragma unroll
        for (int k = 0; k < DPT; ++k) {
            const int u = ptid + k * kPT;
            const int pg = u % PGS, t1 = u / PGS, quad = t1 % (8 * COB), row = t1 / (8 * COB);
            const int gy = ty0 + row, gx = tx0 + G * pg;
            const bool ok = u < S::NDU && gy < p.h && gx < p.w;
            r.dmask |= (ok ? 1u : 0u) << k;
            if (!ok) continue;
            const size_t o = (size_t)(co0 + 4 * quad) * hw + (size_t)gy * p.w + gx;
#pragma unroll
            for (int i = 0; i < 4; ++i) r.g[k][i] = *reinterpret_cast<const uvec*>(gn + o + (size_t)i * hw);
            if (bn)
#pragma unroll
                for (int i = 0; i < 4; ++i) r.y[k][i] = *reinterpret_cast<const uvec*>(yn + o + (size_t)i * hw);
        }
#pragma unroll
        for (int k = 0; k < XPT; ++k) {
            const int u = ptid + k * kPT;
            const int pg = u % PGS, t1 = u / PGS, quad = t1 % 8, pr = t1 / 8;
            const int gy = ty0 - 1 + pr, gx = tx0 + G * pg;
            const bool ok = u < S::NXU && gy >= 0 && gy < p.h && gx < p.w && 4 * quad < p.cin;
            r.xmask |= (ok ? 1u : 0u) << k;
            if (!ok) continue;
            const size_t o = (size_t)(4 * quad) * hw + (size_t)gy * p.w + gx;
#pragma unroll
            for (int i = 0; i < 4; ++i) r.x[k][i] = *reinterpret_cast<const uvec*>(xn + o + (size_t)i * hw);
        }
#pragma unroll
        for (int k = 0; k < HPT; ++k) {
            const int u = ptid + k * kPT;
            const int side = u & 1, t1 = u >> 1, quad = t1 % 8, pr = t1 / 8;
            const int gy = ty0 - 1 + pr, gx = side ? tx0 + TW : tx0 - 1;
            const bool ok = u < S::NHU && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w && 4 * quad < p.cin;
            r.hmask |= (ok ? 1u : 0u) << k;
            if (!ok) continue;
            const size_t o = (size_t)(4 * quad) * hw + (size_t)gy * p.w + gx;
            r.h[k][0] = (unsigned)xn[o] | (unsigned)xn[o + hw] << 16;
            r.h[k][1] = (unsigned)xn[o + 2 * hw] | (unsigned)xn[o + 3 * hw] << 16;
        }
    };

    auto commit = [&](int item, int buf, const RawT& r) {
        unsigned char* lx = lds + buf * S::BUF;
        unsigned char* ld = lx + S::XBYTES;
        int n, tx0, ty0;
        tile_of(item, n, tx0, ty0);
        // ---- dY (optionally the BatchNorm backward of g), channel by channel; dy_out on the side
#pragma unroll
        for (int k = 0; k < DPT; ++k) {
            const int u = ptid + k * kPT;
            if (u >= S::NDU) continue;
            const int pg = u % PGS, t1 = u / PGS, quad = t1 % (8 * COB), row = t1 / (8 * COB);
            const bool ok = r.dmask >> k & 1u;
            unsigned char* img = ld + (quad >> 3) * (S::DPIX * 64);
            const unsigned pd = (unsigned)(row * TW + G * pg);
            float prev[G];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = 4 * quad + i;
                float gv[G];
#pragma unroll
                for (int e = 0; e < G; ++e) gv[e] = 0.f;
                if (ok) {
#pragma unroll
                    for (int e = 0; e < G; e += 2) {
                        gv[e] = up(r.g[k][i][e / 2] & 0xffffu);
                        gv[e + 1] = up(r.g[k][i][e / 2] >> 16);
                    }
                    if (bn) {
                        const float c0 = lbn[c], c1 = lbn[32 * COB + c], c2 = lbn[64 * COB + c],
                                    c3 = lbn[96 * COB + c], c4 = lbn[128 * COB + c];
                        const float al = p.bn_alpha ? p.bn_alpha[(size_t)n * p.cout + co0 + c] : 1.f;
                        const float ad = p.bn_add ? p.bn_add[(size_t)n * p.cout + co0 + c] : 0.f;
#pragma unroll
                        for (int e = 0; e < G; ++e) {
                            const unsigned yw = r.y[k][i][e / 2];
                            const float yv = up((e & 1) ? yw >> 16 : yw & 0xffffu);
                            float dz = fmaf(gv[e], al, ad);
                            if (p.bn_relu && !(fmaf(yv, c0, c1) > 0.f)) dz = 0.f;
                            gv[e] = fmaf(c2, dz, fmaf(c3, yv, c4));
                        }
                        if (p.dy_out != nullptr) {
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
        // ---- A = relu?(x*scale+shift), zero padding stays zero
#pragma unroll
        for (int k = 0; k < XPT; ++k) {
            const int u = ptid + k * kPT;
            if (u >= S::NXU) continue;
            const int pg = u % PGS, t1 = u / PGS, quad = t1 % 8, pr = t1 / 8;
            const bool ok = r.xmask >> k & 1u;
            const unsigned pi = (unsigned)(pr * PW + 1 + G * pg);
            float prev[G];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v[G];
#pragma unroll
                for (int e = 0; e < G; ++e) v[e] = 0.f;
                if (ok) {
#pragma unroll
                    for (int e = 0; e < G; e += 2) {
                        v[e] = up(r.x[k][i][e / 2] & 0xffffu);
                        v[e + 1] = up(r.x[k][i][e / 2] >> 16);
                    }
                    if (pro) {
                        const float sc = lsc[4 * quad + i], sh = lsc[32 + 4 * quad + i];
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
                        *reinterpret_cast<unsigned*>(lx + img_off(pi + e, quad) + 4 * (i >> 1)) = pack2(prev[e], v[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < G; ++e) prev[e] = v[e];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < HPT; ++k) {
            const int u = ptid + k * kPT;
            if (u >= S::NHU) continue;
            const int side = u & 1, t1 = u >> 1, quad = t1 % 8, pr = t1 / 8;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (r.hmask >> k & 1u) {
                v[0] = up(r.h[k][0] & 0xffffu);
                v[1] = up(r.h[k][0] >> 16);
                v[2] = up(r.h[k][1] & 0xffffu);
                v[3] = up(r.h[k][1] >> 16);
                if (pro)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[i] = fmaf(v[i], lsc[4 * quad + i], lsc[32 + 4 * quad + i]);
                        if (p.in_relu) v[i] = fmaxf(v[i], 0.f);
                    }
            }
            u32x2 o;
            o.x = pack2(v[0], v[1]);
            o.y = pack2(v[2], v[3]);
            *reinterpret_cast<u32x2*>(lx + img_off((unsigned)(pr * PW + (side ? PW - 1 : 0)), quad)) = o;
        }
    };

    // ---- the pipeline.  Tile j's loads are issued two iterations before they are transformed:
    //   prologue: issue(0 -> A), issue(1 -> B); commit(0 from A -> image 0); issue(2 -> A)
    //   iteration idx: consumers compute image idx&1 | producers commit(idx+1 -> image (idx+1)&1) from
    //   the register set of parity (idx+1)&1, then refill that set with tile idx+3; barrier.
    // The two roles are separate code paths with the same number of barriers, so that the register
    // allocation of one (two raw tiles) does not add to the other's (144 accumulators).
    __syncthreads();  // lbn / lsc staged
    if (producer) {
        RawT ra, rb;
        if (count > 0) {
            issue(first, ra);
            if (count > 1) issue(first + 1, rb);
            commit(first, 0, ra);
            if (count > 2) issue(first + 2, ra);
        }
        __syncthreads();
        for (int idx = 0; idx < count; idx += 2) {
            if (idx + 1 < count) {  // even iteration: tile idx+1 goes from set B into image 1
                commit(first + idx + 1, 1, rb);
                if (idx + 3 < count) issue(first + idx + 3, rb);
            }
            __syncthreads();
            if (idx + 1 >= count) break;
            if (idx + 2 < count) {  // odd iteration: tile idx+2 goes from set A into image 0
                commit(first + idx + 2, 0, ra);
                if (idx + 4 < count) issue(first + idx + 4, ra);
            }
            __syncthreads();
        }
        for (int k = 1; k < KSPL; ++k)      // the consumers' K-split reduction rounds
            for (int qq = 0; qq < NQ; ++qq) {
                __syncthreads();
                __syncthreads();
            }
        return;
    }

    // ---- consumer side: operand fetch (see lf_wgrad_bf16.hip for the lane maps) + MFMA
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
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
    auto compute = [&](int buf) {
        const unsigned char* xb = lds + buf * S::BUF;
        const unsigned char* db = xb + S::XBYTES + q * (S::DPIX * 64);
#pragma unroll 1
        for (int s = ks; s < NS; s += KSPL) {
            const int f0 = 16 * s + 8 * kh, f1 = f0 + 4;
            const int r0 = f0 / TW, c0 = f0 - r0 * TW, r1 = f1 / TW, c1 = f1 - r1 * TW;
            const bf16x8 B = frag(db, (unsigned)f0, (unsigned)f1);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3, dx = t % 3;
                const bf16x8 A = frag(xb, (unsigned)((r0 + dy) * PW + c0 + dx), (unsigned)((r1 + dy) * PW + c1 + dx));
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[t], 0, 0, 0);
            }
        }
    };
    __syncthreads();
    for (int idx = 0; idx < count; idx += 2) {
        compute(0);
        __syncthreads();
        if (idx + 1 >= count) break;
        compute(1);
        __syncthreads();
    }
    // K-split consumer waves fold into k = 0 through LDS, one sub-block per round (fixed order)
    float* red = reinterpret_cast<float*>(lds);
#pragma unroll 1
    for (int k = 1; k < KSPL; ++k) {
#pragma unroll 1
        for (int qq = 0; qq < NQ; ++qq) {
            __syncthreads();
            if (ks == k && q == qq) {
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[(t * 16 + r) * 64 + lane] = acc[t][r];
            }
            __syncthreads();
            if (ks == 0 && q == qq) {
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] += red[(t * 16 + r) * 64 + lane];
            }
        }
    }
    if (ks == 0) {
        float* out = p.part + (size_t)blockIdx.x * p.cin * 9 * p.cout;
        const int co = co0 + q * 32 + (lane & 31);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (ci < p.cin && co < p.cout) out[((size_t)ci * 9 + t) * p.cout + co] = acc[t][r];
            }
    }
}

template <int TW, int TH, int COB>
int launch_pc(const lf::WgradBf16Args& a, dim3 grid, hipStream_t s) {
    using S = PcShape<TW, TH, COB>;
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16_pc_kernel<TW, TH, COB>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, S::LDS) != hipSuccess) {
            lf::set_error("lf_conv2d_wgrad_bf16: cannot reserve %d bytes of LDS", S::LDS);
            return LF_ERR_LAUNCH;
        }
        raised = true;
    }
    wgrad_bf16_pc_kernel<TW, TH, COB><<<grid, kNT, S::LDS, s>>>(a);
    return LF_OK;
}

}  // namespace

namespace lf {

bool wgrad_bf16_pc_covers(int cin, int cout, int w, int ksize, int tw, int th) {
    // (Cout = 64 would need 196 registers of raw loads per producer thread for two tiles: it spills, and
    // stays on the general kernel)
    return ksize == 3 && cin == 32 && cout == 32 && w % 8 == 0 &&
           ((tw == 56 && th == 4) || (tw == 32 && th == 8));
}

int wgrad_bf16_pc_launch(const WgradBf16Args& a, int tw, int cob, dim3 grid, hipStream_t s) {
    if (tw == 56) return cob == 1 ? launch_pc<56, 4, 1>(a, grid, s) : launch_pc<56, 4, 2>(a, grid, s);
    return cob == 1 ? launch_pc<32, 8, 1>(a, grid, s) : launch_pc<32, 8, 2>(a, grid, s);
}

}  // namespace lf
