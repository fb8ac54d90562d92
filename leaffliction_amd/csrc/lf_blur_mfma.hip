// Gaussian blur (uint8, OpenCV fixed-point semantics) on the i8 matrix cores.
//
// Both passes of the separable filter are banded (Toeplitz) matrix products, and with Q8.8 taps that
// fit a signed byte they are exact in v_mfma_i32_32x32x32_i8:
//
//   horizontal  mid[y][b] = sum_k  in[y][b0 - PADL + k] * Th[k][b]       (bytes of an HWC row; the taps
//               sit CH bytes apart in Th, so the interleaved channels never have to be separated)
//   vertical    out[y][b] = (sum_y' Tv[y][y'] * mid[y'][b] + 2^15) >> 16
//
// Pixels are staged as x ^ 0x80 (= x - 128 as a signed byte).  The horizontal accumulator then holds
// v = mid - 128 * (sum of taps), a signed 16-bit number, which splits into two signed bytes
// (byte 1 of v; byte 0 of v ^ 0x80, which leaves a constant 128 behind); the constants that undo all
// the offsets, and the rounding 2^15, ride in the initial value of the vertical accumulator.  The
// horizontal product is laid out with the image ROW on the A operand's lane, so its 32x32 result has
// the output byte on the lane and 16 rows in the registers: already an operand of a product that sums
// over rows, with no lane movement and no LDS in between (guide: "an accumulator tile as the next
// MFMA's operand").  It is used as the A operand (out^T = mid^T * Tv^T), which puts an output row on
// the lane and four neighbouring bytes of it in consecutive registers: four 4-byte LDS writes per lane.
// BORDER_REFLECT_101 is folded in: to the left and right by staging reflected pixels, at the top and
// bottom by one extra banded matrix each.
//
// A wave owns a 32-byte column of the image and walks down it in 32-row blocks, keeping the last
// three horizontal results in registers; a workgroup is `nwv` neighbouring columns, so a block's rows
// are read and written as runs of 32*nwv bytes, 16 bytes per lane, and it walks on from the bottom of
// one column group to the top of its next one without draining its pipeline.  Measured (rocprofv3,
// 4096 x 224x224x3, 15x15): 146 vector instructions and 9 MFMAs per wave per 1 KiB of output, where the
// dot-product kernel (lf_augment.hip) spends ~400 vector instructions; 332 us, 3.7 TB/s of image bytes.
// What bounds it now is not the memory system: each wave issues 36 % of its cycles and waits the rest
// (dependent MFMA chains, one workgroup barrier per 32 rows, 14 waves per CU at 124 registers).
// Ablations on the same batch (development builds, results wrong by construction): without the workgroup barrier
// 400 us (slower), without the vertical pass 338 us, without global loads 278 us, without global stores 293 us,
// without either 271 us, without either and without the vertical pass 119 us: the arithmetic alone (271 us) and
// the barrier-paced memory traffic alone (~330 us) each nearly fill the 348 us — both have to shrink for the
// kernel to move.
// Nontemporal loads / stores measured slower (2.9 against 3.5 TB/s): neighbouring column groups share their
// 32-byte margins through the L2.
#include "lf_common.h"

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// row of a 32x32 accumulator tile held in register j of a lane of half h
__device__ __forceinline__ int acc_row(int h, int j) { return (j & 3) + 8 * (j >> 2) + 4 * h; }

constexpr int kStageRounds = 2;
constexpr int kTabs = 11;  // banded-matrix fragments: 3 horizontal k-steps, 3 vertical blocks, top and bottom reflection

template <int CH, int KSTEPS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void blur_mfma_kernel(
    const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int h, int w, lf::BlurTaps taps, int ksize,
    int n_images, int nwv, int colgroups) {
    constexpr int PADL = KSTEPS == 2 ? 16 : 32;
    constexpr int PADR = 32 * KSTEPS - 32 - PADL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int tq[32];
    __shared__ int padtab[PADL + PADR];
    __shared__ __attribute__((aligned(16))) unsigned ctab[kTabs * 64 * 4];
    const int R = ksize / 2;
    const int WB = w * CH;
    const int tile_b = 32 * nwv;
    const int gpr = (PADL + tile_b + PADR) / 16;  // 16-byte groups per staged row
    const int pin = 16 * gpr + 16;                // odd number of 16-byte slots: conflict-free b128 reads
    const int pout = tile_b + 16;
    unsigned char* inbuf = smem;                // [3][32][pin], pixels already ^ 0x80
    unsigned char* obuf = smem + 3 * 32 * pin;  // [2][32][pout]
    const int nthreads = 64 * nwv;
    if (threadIdx.x < 32) tq[threadIdx.x] = (int)threadIdx.x < ksize ? taps.k[threadIdx.x] : 0;
    // where, in a staged row, the pixel that a byte outside the image reflects (BORDER_REFLECT_101) lies:
    // entries [0, PADL) for the workgroup at the left side, [PADL, PADL + PADR) for the one at the right
    if ((int)threadIdx.x < PADL + PADR) {
        const int i = threadIdx.x;
        int off;
        if (i < PADL) {
            const int b = i - PADL;                                  // b < 0
            const int x = -((-b + CH - 1) / CH);                     // floor(b / CH)
            off = -x * CH + (b - x * CH) + PADL;
        } else {
            const int e = i - PADL;                                  // byte WB + e
            off = tile_b + PADL - 2 * CH - (e / CH) * CH + e % CH;
        }
        padtab[i] = off;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, ln = lane & 31, lh = lane >> 5;
    int tapsum = 0;
    for (int q = 0; q < ksize; ++q) tapsum += tq[q];

    // ---- the banded matrices: fragment word rg of lane l of table c at ctab[(c * 64 + l) * 4 + rg]
    for (int idx = threadIdx.x; idx < kTabs * 256; idx += nthreads) {
        const int c = idx >> 8, l = (idx >> 2) & 63, rg = idx & 3;
        const int n = l & 31, hh = l >> 5;
        unsigned v = 0;
        for (int t = 0; t < 4; ++t) {
            const int j = 4 * rg + t;
            const int yi = acc_row(hh, j);
            int q = -1;
            if (c < 3) {                      // horizontal k-step c: window byte k feeds output byte n
                const int e = 32 * c + 16 * hh + j - PADL - n + CH * R;
                if (c < KSTEPS && e >= 0 && e % CH == 0) q = e / CH;
            } else if (c < 6) {               // vertical, block c - 4 relative to the output block
                q = 32 * (c - 4) + yi - n + R;
            } else if (c == 6) {              // top: row -yi of the image is row yi
                if (yi > 0) q = R - n - yi;
            } else {                          // bottom: row 31 + d of the last block is row 31 - d
                if (yi < 31) q = 62 - yi - n + R;
            }
            if (q >= 0 && q < ksize) v |= (unsigned)tq[q] << (8 * t);
        }
        ctab[idx] = v;
    }
    __syncthreads();
    auto frag = [&](int c) { return *reinterpret_cast<const i32x4*>(ctab + (c * 64 + lane) * 4); };
    // The horizontal accumulator holds v = sum (x - 128) t = mid - 128 * tapsum, a signed 16-bit number:
    // v = 256 * (signed byte 1 of v) + (byte 0 of v ^ 0x80, as a signed byte) + 128.  The vertical
    // accumulators start from what that leaves out, plus the rounding 2^15.
    const int vbias = __builtin_amdgcn_readfirstlane(128 * tapsum * (1 + tapsum) + (1 << 15));

    const int nblk = h / 32;
    const int items = n_images * colgroups;
    const int mine = items > (int)blockIdx.x ? (items - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const int steps = mine * nblk;

    // this thread's 16-byte groups of a staged block
    int srow[kStageRounds], sgi[kStageRounds];
#pragma unroll
    for (int r = 0; r < kStageRounds; ++r) {
        const int g = (int)threadIdx.x + r * nthreads;
        srow[r] = g < 32 * gpr ? g / gpr : -1;
        sgi[r] = g - (g / gpr) * gpr;
    }
    const int orow = (int)threadIdx.x / (2 * nwv), ogi = (int)threadIdx.x - orow * 2 * nwv;

    // Four cursors walk the workgroup's blocks (all blocks of its first column group, then of its next ...)
    // one step apart: L loads a block's bytes inside the image into LDS (16 per lane), P adds its
    // reflected pixels (LDS to LDS; only the column groups at the image's sides have any), C makes its
    // horizontal result, O the output block above C's.  One barrier per step covers all four.
    struct Cursor {
        int jb, cg, img;
    };
    const int step_img = (int)gridDim.x / colgroups, step_cg = (int)gridDim.x - step_img * colgroups;
    auto advance = [&](Cursor& c) {
        if (++c.jb == nblk) {   // the workgroup's next item: gridDim.x further on
            c.jb = 0;
            c.cg += step_cg;
            c.img += step_img;
            if (c.cg >= colgroups) {
                c.cg -= colgroups;
                ++c.img;
            }
        }
    };
    const size_t img_bytes = (size_t)h * WB;
    Cursor L{0, (int)blockIdx.x % colgroups, (int)blockIdx.x / colgroups};
    Cursor S = L, P = L, C = L, O = L;

    i32x4 hpp_hi = {0, 0, 0, 0}, hpp_lo = {0, 0, 0, 0}, hp_hi = {0, 0, 0, 0}, hp_lo = {0, 0, 0, 0};
    int slotS = 0, slotP = 0, slotC = 0;
    // one step; `ld` takes the loads issued now, `st` holds the ones issued a step ago (the caller swaps them)
    auto step = [&](const int t, lf::u32x4 (&ld)[kStageRounds], lf::u32x4 (&st)[kStageRounds]) {
        {
            // Every lane loads at every step (a group outside the image, a lane without a group, a step past
            // the workgroup's last block: from a clamped address and for nothing).  A load under a branch
            // comes with its wait, and then nothing is in flight while the step computes.
            const int wgb0 = L.cg * tile_b;
#pragma unroll
            for (int r = 0; r < kStageRounds; ++r) {
                int gb = wgb0 - PADL + 16 * sgi[r];
                gb = gb < 0 ? 0 : (gb + 16 > WB ? WB - 16 : gb);
                const int row = srow[r] < 0 ? 0 : srow[r];
                ld[r] = *reinterpret_cast<const lf::u32x4*>(in + L.img * img_bytes + (size_t)(32 * L.jb + row) * WB + gb);
            }
            if (t + 1 < steps) advance(L);
        }
        i32x4 hc_hi = {0, 0, 0, 0}, hc_lo = {0, 0, 0, 0};
        if (t >= 3 && t < steps + 3) {
            const unsigned char* rowp = inbuf + (slotC * 32 + ln) * pin + 32 * wv + 16 * lh;
            i32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s)
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const i32x4*>(rowp + 32 * s), frag(s), acc, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const unsigned t0 = __builtin_amdgcn_perm((unsigned)acc[4 * g + 1], (unsigned)acc[4 * g], 0x05010400u);
                const unsigned t1 = __builtin_amdgcn_perm((unsigned)acc[4 * g + 3], (unsigned)acc[4 * g + 2], 0x05010400u);
                hc_lo[g] = (int)(__builtin_amdgcn_perm(t1, t0, 0x05040100u) ^ 0x80808080u);
                hc_hi[g] = (int)__builtin_amdgcn_perm(t1, t0, 0x07060302u);
            }
            slotC = slotC == 2 ? 0 : slotC + 1;
            advance(C);
        }
        if (t >= 4 && t < steps + 4) {
            // out^T = mid^T * Tv^T: the horizontal result is the A operand (its byte column is the A row), so
            // the output row lands on the lane and four neighbouring bytes of it in consecutive registers
            i32x16 ahi, alo;
            int vb = vbias;
            asm volatile("" : "+s"(vb));   // 16 moves from a scalar here, not a 16-register constant held (spilled) across the loop
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                ahi[i] = 0;
                alo[i] = vb;
            }
            if (O.jb > 0) {
                const i32x4 tb = frag(3);
                ahi = __builtin_amdgcn_mfma_i32_32x32x32_i8(hpp_hi, tb, ahi, 0, 0, 0);
                alo = __builtin_amdgcn_mfma_i32_32x32x32_i8(hpp_lo, tb, alo, 0, 0, 0);
            } else {
                const i32x4 tb = frag(6);
                ahi = __builtin_amdgcn_mfma_i32_32x32x32_i8(hp_hi, tb, ahi, 0, 0, 0);
                alo = __builtin_amdgcn_mfma_i32_32x32x32_i8(hp_lo, tb, alo, 0, 0, 0);
            }
            {
                const i32x4 tb = frag(4);
                ahi = __builtin_amdgcn_mfma_i32_32x32x32_i8(hp_hi, tb, ahi, 0, 0, 0);
                alo = __builtin_amdgcn_mfma_i32_32x32x32_i8(hp_lo, tb, alo, 0, 0, 0);
            }
            if (O.jb + 1 < nblk) {
                const i32x4 tb = frag(5);
                ahi = __builtin_amdgcn_mfma_i32_32x32x32_i8(hc_hi, tb, ahi, 0, 0, 0);
                alo = __builtin_amdgcn_mfma_i32_32x32x32_i8(hc_lo, tb, alo, 0, 0, 0);
            } else {
                const i32x4 tb = frag(7);
                ahi = __builtin_amdgcn_mfma_i32_32x32x32_i8(hp_hi, tb, ahi, 0, 0, 0);
                alo = __builtin_amdgcn_mfma_i32_32x32x32_i8(hp_lo, tb, alo, 0, 0, 0);
            }
            unsigned char* op = obuf + ((t & 1) * 32 + ln) * pout + 32 * wv + 4 * lh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = ((unsigned)ahi[4 * g + k] << 8) + (unsigned)alo[4 * g + k];
                const unsigned t0 = __builtin_amdgcn_perm(v[1], v[0], 0x00000602u);
                const unsigned t1 = __builtin_amdgcn_perm(v[3], v[2], 0x00000602u);
                *reinterpret_cast<unsigned*>(op + 8 * g) = __builtin_amdgcn_perm(t1, t0, 0x05040100u);
            }
        }
        if (t >= 2 && t < steps + 2) {
            const int lw = P.cg == 0 ? PADL / 4 : 0, rw = P.cg == colgroups - 1 ? PADR / 4 : 0;
            if (lw + rw > 0) {
                unsigned char* blk = inbuf + slotP * 32 * pin;
                const int nw = lw + rw, sh = __builtin_ctz(nw);   // 4, 8 or 16 words per row
                for (int u = threadIdx.x; u < 32 * nw; u += nthreads) {
                    const int k = u & (nw - 1);
                    unsigned char* rowp = blk + (u >> sh) * pin;
                    const int kk = k < lw ? 4 * k : 4 * (k - lw);
                    const int* tab = padtab + (k < lw ? kk : PADL + kk);
                    const unsigned v = (unsigned)rowp[tab[0]] | (unsigned)rowp[tab[1]] << 8 |
                                       (unsigned)rowp[tab[2]] << 16 | (unsigned)rowp[tab[3]] << 24;
                    *reinterpret_cast<unsigned*>(rowp + (k < lw ? kk : PADL + tile_b + kk)) = v;
                }
            }
            slotP = slotP == 2 ? 0 : slotP + 1;
            advance(P);
        }
        if (t >= 1 && t < steps + 1) {
            const int wgb0 = S.cg * tile_b;
#pragma unroll
            for (int r = 0; r < kStageRounds; ++r) {
                const int gb = wgb0 - PADL + 16 * sgi[r];
                if (srow[r] >= 0 && gb >= 0 && gb + 16 <= WB) {
                    lf::u32x4 v = st[r];
                    asm volatile("" : "+v"(v));   // the wait for the load belongs here, not up where it was issued
                    *reinterpret_cast<lf::u32x4*>(inbuf + (slotS * 32 + srow[r]) * pin + 16 * sgi[r]) = v ^ 0x80808080u;
                }
            }
            slotS = slotS == 2 ? 0 : slotS + 1;
            advance(S);
        }
        __syncthreads();
        if (t >= 4 && t < steps + 4) {
            const lf::u32x4 v = *reinterpret_cast<const lf::u32x4*>(obuf + ((t & 1) * 32 + orow) * pout + 16 * ogi);
            *reinterpret_cast<lf::u32x4*>(out + O.img * img_bytes + (size_t)(32 * O.jb + orow) * WB + O.cg * tile_b + 16 * ogi) = v;
            advance(O);
        }
        hpp_hi = hp_hi;
        hpp_lo = hp_lo;
        hp_hi = hc_hi;
        hp_lo = hc_lo;
    };
    // A block's bytes are requested at step b, written to LDS at b + 1 (a whole step for them to arrive),
    // completed by reflection at b + 2, multiplied at b + 3; its output is made and stored at b + 4.
    lf::u32x4 sa[kStageRounds], sb[kStageRounds];
#pragma unroll
    for (int r = 0; r < kStageRounds; ++r) sa[r] = sb[r] = lf::u32x4{0u, 0u, 0u, 0u};
    for (int t = 0; t < steps + 4; t += 2) {
        step(t, sa, sb);
        step(t + 1, sb, sa);
    }
}

}  // namespace

namespace lf {

// True when the launch was made (the caller then checks the launch error); false when the shape or the
// taps do not fit this kernel and the dot-product kernels have to serve.
bool blur_mfma_launch(const uint8_t* in, uint8_t* out, int n, int h, int w, int channels,
                      const BlurTaps& taps, int ksize, hipStream_t s) {
    const int wb = w * channels, r = ksize / 2;
    if (ksize < 3 || h % 32 != 0 || wb % 32 != 0 || wb < 48 || channels * r > 32) return false;
    if ((reinterpret_cast<size_t>(in) | reinterpret_cast<size_t>(out)) & 15) return false;
    for (int i = 0; i < ksize; ++i)
        if (taps.k[i] > 127) return false;   // signed-byte operands
    int tapsum = 0;
    for (int i = 0; i < ksize; ++i) tapsum += taps.k[i];
    if (tapsum > 256) return false;          // mid - 128 * tapsum must fit 16 signed bits
    const int nb32 = wb / 32;
    int nwv = 1;
    for (int d = 2; d <= 8; ++d)
        if (nb32 % d == 0) nwv = d;
    const int colgroups = nb32 / nwv;
    const int ksteps = channels * r <= 16 ? 2 : 3;
    if (nwv == 1 && ksteps == 3) return false;   // would need a third staging round
    const int padlr = ksteps == 2 ? 32 : 64;
    const int pin = padlr + 32 * nwv + 16, pout = 32 * nwv + 16;
    const size_t lds = (size_t)3 * 32 * pin + (size_t)2 * 32 * pout;
    const long items = (long)n * colgroups;
    const unsigned grid = (unsigned)(items < 1024 ? items : 1024);
    const unsigned block = 64u * nwv;
#define LF_BLUR_MFMA(CH, KS) \
    blur_mfma_kernel<CH, KS><<<grid, block, lds, s>>>(in, out, h, w, taps, ksize, n, nwv, colgroups)
    if (channels == 3) {
        if (ksteps == 2) LF_BLUR_MFMA(3, 2);
        else LF_BLUR_MFMA(3, 3);
    } else {
        if (ksteps == 2) LF_BLUR_MFMA(1, 2);
        else LF_BLUR_MFMA(1, 3);
    }
#undef LF_BLUR_MFMA
    return true;
}

}  // namespace lf
