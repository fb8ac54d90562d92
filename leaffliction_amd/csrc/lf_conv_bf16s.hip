// libleafhip — streaming bf16 convolution for the layers that dominate the mixed-precision
// training step: the 224x224 and 112x112 stages, where Cin and Cout are at most 64 and the
// tensors are 0.8 - 1.6 GB per batch.  With bf16 storage these convolutions are HBM-bound (72
// FLOP per byte at 32->32 against a ridge of ~310), so the kernel is organised around the
// memory stream, not around the MFMA:
//
//   * the whole filter bank (<= 36 KB of bf16) is copied into LDS ONCE per workgroup and stays
//     there;
//   * a workgroup walks DOWN a column strip of an image (TW pixels wide), tile after tile, and the
//     patch in LDS is a RING of TH + 2 pixel rows: a tile brings in only its TH new rows, the two
//     rows it shares with the tile above are still there (round 3; before, every tile staged its
//     own TH + 2 rows and the input crossed the fabric 1.5x for the halo rows alone — PMC,
//     profiles/pmc_latest_bf16.json of round 2: 2.59 GB per forward launch against 1.64 GB).  The
//     strips of one image go to neighbouring workgroups of one XCD, which walk them at the same
//     pace: the halo columns and the 128-byte lines two strips share are found in that XCD's L2;
//   * the NEW rows of the next tile are in flight in registers while the current tile's MFMAs
//     and epilogue run (all input channels at once: no K-chunk loop, no barrier per chunk);
//   * the patch sits in LDS as [pixel][channel] rows (64 B per pixel at 32 channels) with the
//     16-byte channel groups XOR-swizzled by the pixel index: the B operand of a lane (one pixel,
//     eight consecutive channels) is ONE aligned ds_read_b128 whatever the tap offset, and the
//     A operand (one output channel, eight input channels of one tap) is one ds_read_b128 of the
//     resident weights — 3 LDS reads per 2 MFMAs;
//   * the producer's BatchNorm+ReLU is applied in fp32 while staging (the f32 -> bf16 packing
//     pairs channels of a pixel, which is the transposition);
//   * BatchNorm statistics / BatchNorm-backward sums of the rounded output are accumulated per
//     lane across ALL tiles of the workgroup and reduced once at the end: one partial per
//     workgroup instead of one per tile.
//
// Every global access is 16 bytes per lane: the vector memory pipe moves ~5.3 TB/s with 16-byte
// lanes but only ~3 TB/s with 8-byte and ~1.5 TB/s with 4-byte lanes
// (scripts/microbench/seg_bw.hip, profiles/r02_microbench_seg_bw.txt).  The MFMA leaves a lane with
// ONE pixel column of 16 channels (D[row = output channel][col = pixel]), i.e. 2-byte pieces; the
// accumulators therefore go through LDS once (fp32, [channel][pixel]) and come back as eight
// consecutive pixels of one channel per lane: 16-byte stores, 16-byte loads of the read-modify-
// write operands (old value when accumulating, the BatchNorm input for the backward sums — both
// requested BEFORE the MFMAs), and the statistics of a channel live in one half-wave.
// Replaces Conv2D forward and its input-gradient (srcs/model/cnn.py:27-29 under the mixed_float16
// policy of train.py:179-190).
#include "lf_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int kT = 256;

__device__ __forceinline__ float up(unsigned bits16) { return __uint_as_float(bits16 << 16); }
__device__ __forceinline__ uint16_t down(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v;
    v.x = (__bf16)lo;
    v.y = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
// sum over the 32 lanes of each wave half; the total lands in lane 31 / 63
__device__ __forceinline__ float half_sum32(float v) {
    v += dpp_move<0xB1, 0xf>(v);
    v += dpp_move<0x4E, 0xf>(v);
    v += dpp_move<0x141, 0xf>(v);
    v += dpp_move<0x140, 0xf>(v);
    v += dpp_move<0x142, 0xa>(v);
    return v;
}

template <int TAPS, int CI, int NCO, int TW, int TH>
struct SShape {
    static constexpr int HALO = TAPS == 9 ? 1 : 0;
    static constexpr int PW = TW + 2 * HALO, PH = TH + 2 * HALO, PPIX = PW * PH;
    static constexpr int ROWB = CI * 2;                     // bytes per pixel row of the patch image
    static constexpr int R = 256 / ROWB, C = ROWB / 16;    // rows per 256-byte bank window, 16-byte groups per row
    static constexpr int COUT = 32 * NCO, CH = CI / 16;
    static constexpr int WBYTES = CH * TAPS * 2 * COUT * 16;
    static constexpr int PBYTES = (PPIX * ROWB + 15) / 16 * 16;
    // The epilogue's transpose buffer: fp32 accumulators of HALF a 32-channel block (16 channels x 256 pixels), a
    // quarter of it private to each wave.  It has LDS of its own: with the row ring the patch is never dead (the
    // rows the next tile keeps would be overwritten), and a wave may write its accumulators while slower waves are
    // still reading the patch — no barrier in front of the epilogue.
    static constexpr int EBYTES = 16 * 256 * 4;
    static constexpr int LDS = WBYTES + PBYTES + EBYTES;
};

// RMW = false: a launch with no read-modify-write operand (no accumulate, no BatchNorm-backward mask): the
// 32 registers those operands wait in are free, which is what lets three workgroups share a CU (12 waves at
// <= 168 registers) where the filter bank and the patch are small enough.
template <int TAPS, int CI, int NCO, int TW, int TH, bool XBF, bool RMW>
__global__ __launch_bounds__(kT, (SShape<TAPS, CI, NCO, TW, TH>::LDS <= 52 * 1024 && !RMW)
                                     ? 3 : (SShape<TAPS, CI, NCO, TW, TH>::LDS <= 80 * 1024 ? 2 : 1))
void conv_bf16s_kernel(lf::ConvBf16TrainArgs p) {
    using S = SShape<TAPS, CI, NCO, TW, TH>;
    static_assert(TW * TH == 256 && TW % 8 == 0, "tile = 4 waves x 64 pixels, whole 8-pixel groups");
    static_assert(XBF || CI == 16, "fp32 input: the stem only (3 channels padded to one 16-channel group)");
    constexpr int HALO = S::HALO, PW = S::PW, PH = S::PH, ROWB = S::ROWB, COUT = S::COUT, CH = S::CH;
    constexpr int G = XBF ? 8 : 4;           // pixels per staging unit: 16 bytes of bf16 / of fp32
    constexpr int PGS = TW / G, NB = 2;
    typedef unsigned uvec __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* lw = lds;
    unsigned char* lp = lds + S::WBYTES;
    float* le = reinterpret_cast<float*>(lp + S::PBYTES);  // [16 channels][256 pixels]
    __shared__ float lsc[2 * CI];
    __shared__ float lst[2 * COUT];
    __shared__ float los[2 * COUT];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, px = lane & 31, kh = lane >> 5;
    const size_t hw = (size_t)p.h * p.w;
    const bool pro = p.in_scale != nullptr;
    const bool stats = p.stat_part != nullptr || p.unit_sums != nullptr, masked = RMW && p.stat_mask_y != nullptr;
    const bool accumulate = RMW && p.accumulate;
    // What a workgroup walks: SEGMENTS of column strips (a strip = TW columns of one image; a segment = seg_tiles
    // consecutive tiles of it, top to bottom — whole strips when the launch has enough of them to fill the chip).
    // interleave = 1: image i belongs to XCD i % 8 (workgroups k, k+8, ... run on XCD k), whose workgroups take that
    // XCD's segments in order — the strips of one image by NEIGHBOURING workgroups, started together —, so what two
    // strips share (halo columns, 128-byte lines that straddle a strip boundary) crosses the fabric once.
    // interleave = 0 (small launches): segments dealt round-robin over the grid.
    const int SG = p.tiles_x, UI = SG * p.segs;   // strips, segments per image
    int unit_rem = 0;                             // the current segment's index inside its image
    const int xk = blockIdx.x & 7, xj = blockIdx.x >> 3, xw = gridDim.x >> 3;
    const int my_total = (p.interleave & 1) ? (p.n > xk ? (p.n - xk + 7) / 8 : 0) * UI : p.n * UI;
    const int my_first = (p.interleave & 1) ? xj : (int)blockIdx.x, my_step = (p.interleave & 1) ? xw : (int)gridDim.x;
    const int my_units = my_total > my_first ? (my_total - my_first + my_step - 1) / my_step : 0;
    auto unit_of = [&](int ui, int& n, int& tx0, int& t_first, int& t_count) {
        const int q = my_first + ui * my_step;
        const int im = q / UI, rem = q - im * UI, seg = rem / SG;
        unit_rem = rem;
        n = (p.interleave & 1) ? im * 8 + xk : im;
        tx0 = (rem - seg * SG) * TW;
        t_first = seg * p.seg_tiles;
        t_count = min(p.seg_tiles, p.tiles_y - t_first);
    };

    // ---- one-time: filter bank -> LDS as [chunk][tap][k half][cout][8 channels] (16 B per entry)
    {
        constexpr int NW = CH * TAPS * COUT * 2;
        for (int e = tid; e < NW; e += kT) {
            const int h2 = e & 1, co = (e >> 1) % COUT, ct = (e >> 1) / COUT;  // ct = chunk * TAPS + tap
            const lf::u32x4 v = *reinterpret_cast<const lf::u32x4*>(p.wprep + ((size_t)ct * p.cout + co) * 16 + 8 * h2);
            *reinterpret_cast<lf::u32x4*>(lw + ((ct * 2 + h2) * COUT + co) * 16) = v;
        }
        if (pro)
            for (int c = tid; c < CI; c += kT) {
                lsc[c] = c < p.cin ? p.in_scale[c] : 1.f;
                lsc[CI + c] = c < p.cin ? p.in_shift[c] : 0.f;
            }
        for (int c = tid; c < COUT; c += kT) {  // per-channel epilogue constants: pivot | mask scale, mask shift
            lst[c] = masked ? p.mask_scale[c] : ((stats && p.stat_pivot != nullptr) ? p.stat_pivot[c] : 0.f);
            lst[COUT + c] = masked ? p.mask_shift[c] : 0.f;
            los[c] = p.out_scale != nullptr ? p.out_scale[c] : 1.f;
            los[COUT + c] = p.out_scale != nullptr ? p.out_shift[c] : 0.f;
        }
        if (!XBF)  // channels 4..15 of the stem's rows are never written by the staging: zero them once
            for (int e = tid; e < S::PBYTES / 16; e += kT)
                *reinterpret_cast<lf::u32x4*>(lp + 16 * e) = lf::u32x4{0u, 0u, 0u, 0u};
    }

    auto poff = [&](unsigned pp, unsigned c16) -> unsigned {  // byte offset of 16-byte group c16 of patch pixel pp
        return pp * ROWB + ((c16 ^ ((pp / S::R) % S::C)) << 4);
    };

    // ---- staging units: 4 channels x G pixels (interior: one 16-byte load per channel), 4 channels x 1
    // pixel (halo columns).  One staging pass covers up to TH image rows gy0 .. gy0 + nrows - 1 of the strip; row gy
    // lives in ring slot (gy + HALO) mod PH.
    constexpr int QD = XBF ? CI / 4 : 1;  // channel quads that carry data
    constexpr int NXU = QD * TH * PGS, XPT = (NXU + kT - 1) / kT;
    constexpr int NHU = QD * TH * 2 * HALO, HPT = (NHU + kT - 1) / kT;
    static_assert(2 * HALO <= TH, "the two rows that prime a strip go through the same staging registers");
    uvec rx[XPT][XBF ? 4 : 3];
    unsigned rh[HPT > 0 ? HPT : 1][XBF ? 4 : 3];   // raw loads: combining them here would wait for them here
    unsigned xmask = 0, hmask = 0;
    int st_slot = 0, st_rows = 0;   // of the pass in flight: ring slot of its first row, its row count

    auto issue = [&](int n, int tx0, int gy0, int nrows) {
        xmask = hmask = 0;
        st_rows = nrows;
        st_slot = (gy0 + HALO) % PH;   // gy0 >= -HALO
        const uint16_t* xb = static_cast<const uint16_t*>(p.x) + (XBF ? (size_t)n * p.cin * hw : 0);
        const float* xf = static_cast<const float*>(p.x) + (XBF ? 0 : (size_t)n * p.cin * hw);
#pragma unroll
        for (int k = 0; k < XPT; ++k) {
            const int u = tid + k * kT;
            const int pg = u % PGS, t1 = u / PGS, quad = t1 % QD, pr = t1 / QD;
            const int gy = gy0 + pr, gx = tx0 + G * pg;
            const bool ok = pr < nrows && gy >= 0 && gy < p.h && gx < p.w && 4 * quad < p.cin;
            xmask |= (ok ? 1u : 0u) << k;
            if (!ok) continue;
            const size_t o = (size_t)(4 * quad) * hw + (size_t)gy * p.w + gx;
            if (XBF) {
#pragma unroll
                for (int i = 0; i < 4; ++i) rx[k][i] = *reinterpret_cast<const uvec*>(xb + o + (size_t)i * hw);
            } else {
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    if (i < p.cin) rx[k][i] = *reinterpret_cast<const uvec*>(xf + o + (size_t)i * hw);
            }
        }
#pragma unroll
        for (int k = 0; k < HPT; ++k) {
            const int u = tid + k * kT;
            const int side = u & 1, t1 = u >> 1, quad = t1 % QD, pr = t1 / QD;
            const int gy = gy0 + pr, gx = side ? tx0 + TW : tx0 - 1;
            const bool ok = pr < nrows && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w && 4 * quad < p.cin;
            hmask |= (ok ? 1u : 0u) << k;
            if (!ok) continue;
            const size_t o = (size_t)(4 * quad) * hw + (size_t)gy * p.w + gx;
            if (XBF) {
#pragma unroll
                for (int i = 0; i < 4; ++i) rh[k][i] = xb[o + (size_t)i * hw];
            } else {
#pragma unroll
                for (int i = 0; i < 3; ++i) rh[k][i] = i < p.cin ? __float_as_uint(xf[o + (size_t)i * hw]) : 0u;
            }
        }
    };

    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < XPT; ++k) {
            const int u = tid + k * kT;
            const int pg = u % PGS, t1 = u / PGS, quad = t1 % QD, pr = t1 / QD;
            if (pr >= st_rows) continue;
            const bool ok = xmask >> k & 1u;
            int slot = st_slot + pr;
            slot = slot >= PH ? slot - PH : slot;
            const unsigned pi = (unsigned)(slot * PW + HALO + G * pg);
            float prev[G];
            // channel by channel; every second channel the pair goes to LDS, one dword per pixel
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v[G];
#pragma unroll
                for (int e = 0; e < G; ++e) v[e] = 0.f;  // zero padding stays exactly zero
                if (ok && (XBF || (i < 3 && i < p.cin))) {
                    if (XBF) {
#pragma unroll
                        for (int e = 0; e < G; e += 2) {
                            v[e] = up(rx[k][i][e / 2] & 0xffffu);
                            v[e + 1] = up(rx[k][i][e / 2] >> 16);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < G; ++e) v[e] = __uint_as_float(rx[k][i < 3 ? i : 0][e]);
                    }
                    if (pro) {
                        const float sc = lsc[4 * quad + i], sh = lsc[CI + 4 * quad + i];
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
                        *reinterpret_cast<unsigned*>(lp + poff(pi + e, (unsigned)quad >> 1) + 8 * (quad & 1) +
                                                     4 * (i >> 1)) = pack2(prev[e], v[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < G; ++e) prev[e] = v[e];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < HPT; ++k) {
            const int u = tid + k * kT;
            const int side = u & 1, t1 = u >> 1, quad = t1 % QD, pr = t1 / QD;
            if (pr >= st_rows) continue;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (hmask >> k & 1u) {
                if (XBF) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = up(rh[k][i]);
                } else {
#pragma unroll
                    for (int i = 0; i < 3; ++i) v[i] = __uint_as_float(rh[k][i]);
                }
                if (pro)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[i] = fmaf(v[i], lsc[4 * quad + i], lsc[CI + 4 * quad + i]);
                        if (p.in_relu) v[i] = fmaxf(v[i], 0.f);
                    }
                if (!XBF) v[3] = 0.f;
            }
            u32x2 o;
            o.x = pack2(v[0], v[1]);
            o.y = pack2(v[2], v[3]);
            int slot = st_slot + pr;
            slot = slot >= PH ? slot - PH : slot;
            const unsigned pi = (unsigned)(slot * PW + (side ? PW - 1 : 0));
            *reinterpret_cast<u32x2*>(lp + poff(pi, (unsigned)quad >> 1) + 8 * (quad & 1)) = o;
        }
    };

    // MFMA geometry: the wave owns 64 consecutive flat tile positions; lane px holds positions 2*px
    // (block 0) and 2*px + 1 (block 1): the same row, adjacent columns
    const int f0 = 64 * wv + 2 * px;
    const int prow = f0 / TW, pcol = f0 - prow * TW;
    // epilogue geometry: thread = (pixel group eg of 8 consecutive flat positions, channel ec + 8*j), the groups
    // of a wave being the 64 positions whose accumulators the wave itself holds: what a wave reads back from `le`
    // it has written itself, so the transposition needs no workgroup barrier
    const int eg = 8 * wv + (lane & 7), ec = lane >> 3;
    const int erow = (8 * eg) / TW, ecol = (8 * eg) - erow * TW;

    f32x16 acc[NB][NCO];
    float s1[NCO][4], s2[NCO][4];  // running sums of this thread's channels over all its pixels and tiles
#pragma unroll
    for (int cb = 0; cb < NCO; ++cb)
#pragma unroll
        for (int j = 0; j < 4; ++j) s1[cb][j] = s2[cb][j] = 0.f;
    uvec rold[NCO][4], rmask[NCO][4];  // eight pixels of one channel each, as stored

    // A tile's new rows are gy = ty * TH + HALO .. + TH - 1; a segment's first tile also needs the 2 * HALO rows above
    // them (the "prime" pass: staged when the segment starts, its latency exposed once per segment).
    int sn = 0, stx0 = 0, stf = 0, stc = 0;
    if (my_units > 0) {
        unit_of(0, sn, stx0, stf, stc);
        issue(sn, stx0, stf * TH + HALO, TH);
    }
    for (int ui = 0; ui < my_units; ++ui) {
    int n, tx0, t_first, t_count;
    unit_of(ui, n, tx0, t_first, t_count);
    const int cur_rem = unit_rem;   // (unit_of is called again for the next segment's prefetch)
    for (int tt = 0; tt < t_count; ++tt) {
        __syncthreads();  // the previous tile's operand reads are done (first pass: weights / lsc / lst staged)
        const int ty0 = (t_first + tt) * TH;
        commit();
        if (HALO > 0 && tt == 0) {
            issue(n, tx0, ty0 - HALO, 2 * HALO);
            commit();
        }
        // the next tile's new rows: in flight during the MFMAs and the epilogue
        if (tt + 1 < t_count) {
            issue(n, tx0, ty0 + TH + HALO, TH);
        } else if (ui + 1 < my_units) {
            unit_of(ui + 1, sn, stx0, stf, stc);
            issue(sn, stx0, stf * TH + HALO, TH);
        }
        // the epilogue's read-modify-write operands: requested now, consumed after the MFMAs
        uint16_t* yb = p.y + (size_t)n * p.cout * hw;
        const int gy = ty0 + erow, gx = tx0 + ecol;
        const bool ok = gy < p.h && gx < p.w;  // the 8-pixel group is inside or outside as a whole (w % 8 == 0)
        const size_t po = ok ? (size_t)gy * p.w + gx : 0;
        if (accumulate) {
#pragma unroll
            for (int cb = 0; cb < NCO; ++cb)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    rold[cb][j] = *reinterpret_cast<const uvec*>(yb + (size_t)(cb * 32 + 8 * j + ec) * hw + po);
        }
        if (masked) {
            const uint16_t* my = p.stat_mask_y + (size_t)n * p.cout * hw;
#pragma unroll
            for (int cb = 0; cb < NCO; ++cb)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    rmask[cb][j] = *reinterpret_cast<const uvec*>(my + (size_t)(cb * 32 + 8 * j + ec) * hw + po);
        }
        __syncthreads();
        // patch row r of this tile (image row ty0 - HALO + r) sits in ring slot (ty0 + r) mod PH
        unsigned rowpp[TAPS == 9 ? 3 : 1];
        {
            int slot = (ty0 + prow) % PH;
#pragma unroll
            for (int dy = 0; dy < (TAPS == 9 ? 3 : 1); ++dy) {
                rowpp[dy] = (unsigned)(slot * PW + pcol);
                slot = slot + 1 >= PH ? 0 : slot + 1;
            }
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int cb = 0; cb < NCO; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nb][cb][r] = 0.f;
        // One GROUP = one 16-channel chunk of one filter row: its three taps read four neighbouring patch pixels between
        // them (dx + nb = 0 .. 3), so a group is 4 B reads + 3 A reads (per output block) for 6 MFMAs (per output
        // block) — and all of a group's reads are issued before its first MFMA, so that they are in flight beside the
        // PREVIOUS group's MFMAs (the compiler kept two or three reads in flight, every MFMA behind an operand
        // requested one MFMA earlier).  Measured: 32->32 @224 forward 593-606 -> 582 us, input gradient with accumulate
        // 622-630 -> 606 us (-3 %); the same software pipelining of the weight-gradient kernel's k-steps changed
        // nothing and was not kept.  The order of the contributions to each accumulator — (chunk, filter row, tap) —
        // is unchanged: same bits.
        constexpr int ROWS = TAPS == 9 ? 3 : 1, TPR = TAPS / ROWS, NBP = TPR + NB - 1, NG = CH * ROWS;
        // With one output block per workgroup the registers allow TWO groups' operands: group g + 1 is requested,
        // whole, before group g's MFMAs are issued (scheduling barriers keep the compiler from sinking the reads back
        // to their uses).  Two output blocks (NCO = 2) are at their register budget and keep the compiler's order.
        constexpr bool DEEP = NCO == 1 && CI >= 32;   // (the stem variant runs three workgroups per CU at 168 registers)
        bf16x8 A[DEEP ? 2 : 1][TPR][NCO], B[DEEP ? 2 : 1][NBP];
        auto fetch = [&](int g, int buf) {
            const int ch = g / ROWS, dy = g - ch * ROWS;
#pragma unroll
            for (int j = 0; j < NBP; ++j)
                B[buf][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const lf::u32x4*>(
                    lp + poff(rowpp[dy] + (unsigned)j, (unsigned)(2 * ch + kh))));
#pragma unroll
            for (int dx = 0; dx < TPR; ++dx)
#pragma unroll
                for (int cb = 0; cb < NCO; ++cb)
                    A[buf][dx][cb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const lf::u32x4*>(
                        lw + ((((ch * TAPS + dy * TPR + dx) * 2 + kh) * COUT) + cb * 32 + px) * 16));
        };
        if (DEEP) fetch(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int buf = DEEP ? (g & 1) : 0;
            if (DEEP) {
                if (g + 1 < NG) fetch(g + 1, buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                fetch(g, 0);
            }
#pragma unroll
            for (int dx = 0; dx < TPR; ++dx)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int cb = 0; cb < NCO; ++cb)
                        acc[nb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[buf][dx][cb], B[buf][dx + nb],
                                                                              acc[nb][cb], 0, 0, 0);
            if (DEEP) __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue, one 32-channel block at a time through LDS: lane (pixel pair, 16 channels)
        // -> thread (8 pixels, 4 channels); then 16-byte stores and the sums of the rounded values
#pragma unroll
        for (int cb = 0; cb < NCO; ++cb)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
            for (int r = 8 * hf; r < 8 * hf + 8; ++r) {
                const int cl = 8 * ((r >> 2) & 1) + 4 * kh + (r & 3);   // channel within the half block
                *reinterpret_cast<float2*>(le + cl * 256 + 64 * wv + 2 * px) = make_float2(acc[0][cb][r], acc[1][cb][r]);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's own LDS writes (no other wave reads them)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 2 * hf; j < 2 * hf + 2; ++j) {
                const int cl = 8 * j + ec, co = cb * 32 + cl;
                const lf::f32x4 a0 = *reinterpret_cast<const lf::f32x4*>(le + (cl & 15) * 256 + 8 * eg);
                const lf::f32x4 a1 = *reinterpret_cast<const lf::f32x4*>(le + (cl & 15) * 256 + 8 * eg + 4);
                float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                if (accumulate)
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        v[e] += up(rold[cb][j][e / 2] & 0xffffu);
                        v[e + 1] += up(rold[cb][j][e / 2] >> 16);
                    }
                if (p.out_scale != nullptr) {
                    const float osc = los[co], osh = los[COUT + co];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], osc, osh);
                }
                if (p.out_relu)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                uvec o;
#pragma unroll
                for (int e = 0; e < 8; e += 2) o[e / 2] = pack2(v[e], v[e + 1]);
                if (ok) *reinterpret_cast<uvec*>(yb + (size_t)co * hw + po) = o;
                if (!stats || !ok) continue;
                float a = 0.f, b = 0.f;
                if (!masked) {
                    const float pv = lst[co];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float d = up((e & 1) ? o[e / 2] >> 16 : o[e / 2] & 0xffffu) - pv;
                        a += d;
                        b = fmaf(d, d, b);
                    }
                } else {
                    const float msc = lst[co], msh = lst[COUT + co];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float rv = up((e & 1) ? o[e / 2] >> 16 : o[e / 2] & 0xffffu);
                        const unsigned mw = rmask[cb][j][e / 2];
                        const float yv = up((e & 1) ? mw >> 16 : mw & 0xffffu);
                        const float d = (!p.mask_relu || fmaf(yv, msc, msh) > 0.f) ? rv : 0.f;
                        a += d;
                        b = fmaf(d, yv, b);
                    }
                }
                s1[cb][j] += a;
                s2[cb][j] += b;
            }
            __builtin_amdgcn_wave_barrier();   // the reads above stay in front of the next block's writes
        }
    }
    if (p.unit_sums != nullptr) {
        // inference: this segment's channel sums of the stored activation (the squeeze of the block's SE gate adds
        // them up per image).  The patch is dead between segments: the reduction scratch lies over it.
        __syncthreads();
        float* red = reinterpret_cast<float*>(lp);   // [4 waves][COUT]
#pragma unroll
        for (int cb = 0; cb < NCO; ++cb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = s1[cb][j];
#pragma unroll
                for (int m = 1; m < 8; m <<= 1) a += __shfl_xor(a, m, 64);
                if ((lane & 7) == 0) red[wv * COUT + cb * 32 + 8 * j + ec] = a;
                s1[cb][j] = s2[cb][j] = 0.f;
            }
        __syncthreads();
        for (int c = tid; c < COUT; c += kT)
            p.unit_sums[((size_t)n * UI + cur_rem) * p.cout + c] =
                (red[c] + red[COUT + c]) + (red[2 * COUT + c] + red[3 * COUT + c]);
    }
    }
    if (p.stat_part != nullptr) {
        // one partial per workgroup: a channel's pixel groups are 8 lanes (lane & 7) in each of the four waves
        __syncthreads();
        float* red = reinterpret_cast<float*>(lp);   // [4 waves][COUT][2], over the patch (dead now)
#pragma unroll
        for (int cb = 0; cb < NCO; ++cb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = s1[cb][j], b = s2[cb][j];
#pragma unroll
                for (int m = 1; m < 8; m <<= 1) {
                    a += __shfl_xor(a, m, 64);
                    b += __shfl_xor(b, m, 64);
                }
                if ((lane & 7) == 0) {
                    red[(wv * COUT + cb * 32 + 8 * j + ec) * 2] = a;
                    red[(wv * COUT + cb * 32 + 8 * j + ec) * 2 + 1] = b;
                }
            }
        __syncthreads();
        for (int c = tid; c < COUT; c += kT) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int v = 0; v < 4; ++v) {   // fixed order: deterministic
                a += red[(v * COUT + c) * 2];
                b += red[(v * COUT + c) * 2 + 1];
            }
            float* dst = p.stat_part + ((size_t)c * (size_t)p.stat_tiles + blockIdx.x) * 2;
            dst[0] = a;
            dst[1] = b;
        }
    }
}

struct SPlan {
    bool ok;
    int ci, nco, tw, th, tiles_x, tiles_y, seg_tiles, segs, wgs, interleave;
};

SPlan plan_s(int n, int cin, int h, int w, int cout, int ksize, int x_bf16) {
    SPlan pl{};
    pl.ok = false;
    if (cout != 32 && cout != 64) return pl;
    if (w % 8 != 0) return pl;                              // 16-byte rows
    if (!x_bf16) {
        if (cin > 3 || ksize != 3 || cout != 32) return pl;
        pl.ci = 16;
    } else {
        if (cin != 32 && cin != 64) return pl;
        pl.ci = cin;
    }
    pl.nco = cout / 32;
    if (ksize == 1 && !(pl.ci == 32 && pl.nco == 2) && !(pl.ci == 64 && pl.nco == 1)) return pl;
    // Rows of a tile are the unit of every global access (64 px = 128 bytes of bf16 per channel row): measured
    // on 32->32 @224, 64x4 tiles against 32x8: forward 687 -> 659 us, input gradient with accumulate + mask +
    // sums 1167 -> 913 us; 64->64 @112 against 16x16: 724 -> 513 and 1043 -> 699 us — although an eighth of
    // the last tile of a 224- or 112-pixel row is empty (scripts/microbench/conv_modes.py).
    if (w >= 64) { pl.tw = 64; pl.th = 4; }
    else if (w % 32 == 0) { pl.tw = 32; pl.th = 8; }
    else { pl.tw = 16; pl.th = 16; }
    pl.tiles_x = (w + pl.tw - 1) / pl.tw;
    pl.tiles_y = (h + pl.th - 1) / pl.th;
    // Column strips are cut into segments only when there are too few of them to fill the chip twice over (small
    // batches): a segment re-stages the two rows above it, so whole strips are what the full-size step gets
    // (batch 256 @224: 1,024 strips of 56 tiles; @112: 512 strips in two segments of 14 tiles).
    const int strips = n * pl.tiles_x;
    int segs = (4 * 256 + strips - 1) / strips;
    const int max_segs = (pl.tiles_y + 3) / 4;        // at least four tiles to a segment
    if (segs > max_segs) segs = max_segs;
    if (segs < 1) segs = 1;
    pl.seg_tiles = (pl.tiles_y + segs - 1) / segs;
    pl.segs = (pl.tiles_y + pl.seg_tiles - 1) / pl.seg_tiles;
    const int units = strips * pl.segs;
    pl.wgs = units < 256 * 2 ? units : 256 * 2;
    pl.interleave = (pl.wgs % 8 == 0 && n >= 8) ? 1 : 0;
    pl.ok = true;
    return pl;
}

template <int TAPS, int CI, int NCO, int TW, int TH, bool XBF, bool RMW>
int launch_s2(const lf::ConvBf16TrainArgs& a, int wgs, hipStream_t s) {
    using S = SShape<TAPS, CI, NCO, TW, TH>;
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16s_kernel<TAPS, CI, NCO, TW, TH, XBF, RMW>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, S::LDS) != hipSuccess) {
            lf::set_error("lf_conv2d_bf16_train: cannot reserve %d bytes of LDS", S::LDS);
            return LF_ERR_LAUNCH;
        }
        raised = true;
    }
    conv_bf16s_kernel<TAPS, CI, NCO, TW, TH, XBF, RMW><<<wgs, kT, S::LDS, s>>>(a);
    return LF_OK;
}

template <int TAPS, int CI, int NCO, int TW, int TH, bool XBF>
int launch_s(const lf::ConvBf16TrainArgs& a, int wgs, hipStream_t s) {
    return (a.accumulate || a.stat_mask_y != nullptr) ? launch_s2<TAPS, CI, NCO, TW, TH, XBF, true>(a, wgs, s)
                                                      : launch_s2<TAPS, CI, NCO, TW, TH, XBF, false>(a, wgs, s);
}

template <int TW, int TH>
int dispatch_s(const SPlan& pl, int ksize, const lf::ConvBf16TrainArgs& a, hipStream_t s) {
    if (pl.ci == 16) return launch_s<9, 16, 1, TW, TH, false>(a, pl.wgs, s);
    if (ksize == 3) {
        if (pl.ci == 32 && pl.nco == 1) return launch_s<9, 32, 1, TW, TH, true>(a, pl.wgs, s);
        if (pl.ci == 32 && pl.nco == 2) return launch_s<9, 32, 2, TW, TH, true>(a, pl.wgs, s);
        if (pl.ci == 64 && pl.nco == 1) return launch_s<9, 64, 1, TW, TH, true>(a, pl.wgs, s);
        if (pl.ci == 64 && pl.nco == 2) return launch_s<9, 64, 2, TW, TH, true>(a, pl.wgs, s);
    } else {
        if (pl.ci == 32 && pl.nco == 2) return launch_s<1, 32, 2, TW, TH, true>(a, pl.wgs, s);
        if (pl.ci == 64 && pl.nco == 1) return launch_s<1, 64, 1, TW, TH, true>(a, pl.wgs, s);
    }
    lf::set_error("lf_conv2d_bf16_train: no streaming kernel for cin %d nco %d ksize %d", pl.ci, pl.nco, ksize);
    return LF_ERR_INVALID;
}

}  // namespace

namespace lf {

long long conv_bf16s_parts(int n, int cin, int h, int w, int cout, int ksize, int x_bf16) {
    const SPlan pl = plan_s(n, cin, h, w, cout, ksize, x_bf16);
    return pl.ok ? pl.wgs : 0;
}

int conv_bf16s_units_per_image(int n, int cin, int h, int w, int cout, int ksize, int x_bf16) {
    const SPlan pl = plan_s(n, cin, h, w, cout, ksize, x_bf16);
    return pl.ok ? pl.tiles_x * pl.segs : 0;
}

int conv_bf16s_launch(ConvBf16TrainArgs a, int ksize, int x_bf16, hipStream_t s) {
    const SPlan pl = plan_s(a.n, a.cin, a.h, a.w, a.cout, ksize, x_bf16);
    if (!pl.ok) {
        set_error("lf_conv2d_bf16_train: shape not covered by the streaming kernel");
        return LF_ERR_INVALID;
    }
    a.tiles_x = pl.tiles_x; a.tiles_y = pl.tiles_y; a.seg_tiles = pl.seg_tiles; a.segs = pl.segs;
    a.interleave = pl.interleave;
    a.stat_tiles = pl.wgs;
    if (pl.tw == 64) return dispatch_s<64, 4>(pl, ksize, a, s);
    return pl.tw == 32 ? dispatch_s<32, 8>(pl, ksize, a, s) : dispatch_s<16, 16>(pl, ksize, a, s);
}

}  // namespace lf
