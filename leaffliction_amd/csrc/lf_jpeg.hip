// libleafhip — the pixel half of a baseline JPEG encoder (4:2:0), as libjpeg(-turbo) computes it for
// Pillow's Image.save(path, quality=q) — the reference's ImageLoader.save_pil_image
// (srcs/utils/image_utils.py:49-56): jccolor.c rgb_ycc_convert (16-bit fixed point), jcsample.c
// h2v2_downsample (bias 1, 2, 1, 2 ...), jfdctint.c (CONST_BITS 13, PASS1_BITS 2), jcdctmgr.c
// quantisation of the 8x-scaled coefficients (round half away from zero).  All integer: the quantised
// coefficients equal libjpeg's bit for bit, so the host's entropy coder (lf_jpeg_host.cpp) writes the
// very file Pillow writes (tests/test_jpeg_gpu.py compares the bytes).
//
// Output: coef[image][MCU][6 blocks: Y00 Y01 Y10 Y11 Cb Cr][64] int16 in ZIGZAG order — the order the
// scan is coded in, 768 contiguous bytes per 16x16 MCU, as many bytes as the RGB pixels it replaces.
// A workgroup takes four MCUs of a row at a time (a 64x16 pixel strip: 12-byte pixel groups in, 3 KiB of
// coefficients out as 16 bytes per lane); HBM-bound byte work, nothing GEMM-shaped.
#include "lf_common.h"

namespace {

constexpr int kT = 256, kGroup = 4;   // MCUs per workgroup pass

struct JpegQuant {
    uint16_t div[2][64];   // (Q << 3) of the luminance / chrominance table, row-major
    uint8_t pos[64];       // row-major index -> zigzag position
};

constexpr int kCB = 13, kP1 = 2;
#define LF_FIX(x) ((int)((x) * (1 << kCB) + 0.5))

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// jpeg_fdct_islow, one 1-D pass over d[0..7]; FIRST: the row pass (results scaled up by 2^PASS1_BITS)
template <bool FIRST>
__device__ __forceinline__ void fdct8(int* d) {
    const int t0 = d[0] + d[7], t7 = d[0] - d[7], t1 = d[1] + d[6], t6 = d[1] - d[6];
    const int t2 = d[2] + d[5], t5 = d[2] - d[5], t3 = d[3] + d[4], t4 = d[3] - d[4];
    const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    constexpr int SH = FIRST ? kCB - kP1 : kCB + kP1;
    d[0] = FIRST ? (t10 + t11) << kP1 : descale(t10 + t11, kP1);
    d[4] = FIRST ? (t10 - t11) << kP1 : descale(t10 - t11, kP1);
    int z1 = (t12 + t13) * LF_FIX(0.541196100);
    d[2] = descale(z1 + t13 * LF_FIX(0.765366865), SH);
    d[6] = descale(z1 + t12 * (-LF_FIX(1.847759065)), SH);
    z1 = t4 + t7;
    int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
    const int z5 = (z3 + z4) * LF_FIX(1.175875602);
    const int u4 = t4 * LF_FIX(0.298631336), u5 = t5 * LF_FIX(2.053119869);
    const int u6 = t6 * LF_FIX(3.072711026), u7 = t7 * LF_FIX(1.501321110);
    z1 *= -LF_FIX(0.899976223);
    z2 *= -LF_FIX(2.562915447);
    z3 = z3 * (-LF_FIX(1.961570560)) + z5;
    z4 = z4 * (-LF_FIX(0.390180644)) + z5;
    d[7] = descale(u4 + z1 + z3, SH);
    d[5] = descale(u5 + z2 + z4, SH);
    d[3] = descale(u6 + z2 + z3, SH);
    d[1] = descale(u7 + z1 + z4, SH);
}

// RAGGED: sizes that are not whole MCUs, padded as libjpeg pads them — samples replicated to the right (at
// full resolution, before the chroma box filter) and downwards (the last luminance row; the last chroma row), and
// luminance blocks wholly outside the image turned into dummies (no AC, the DC of the previous block of their MCU;
// jccoefct.c compress_data).  Byte loads instead of 12-byte groups: rows are not 4-byte aligned.
// MULTI: images of different sizes in one launch (the balancer's rotated canvases): `items` (device) gives every
// image its place in the pixel buffer and in the coefficient buffer, its size and the number of the first of its
// (MCU row, group of kGroup MCUs) passes; a pass finds its image by bisection.
template <bool RAGGED, bool MULTI = false>
__global__ __launch_bounds__(kT) void jpeg_fdct_quant_kernel(const uint8_t* __restrict__ rgb0,
                                                             int16_t* __restrict__ coef0, int h0, int w0, int n_images,
                                                             JpegQuant q, const lf_jpeg_item* __restrict__ items = nullptr,
                                                             long total_groups = 0) {
    __shared__ __attribute__((aligned(16))) uint8_t sy[16][16 * kGroup], scb[16][16 * kGroup], scr[16][16 * kGroup];
    __shared__ uint8_t sc[2][8][8 * kGroup];
    __shared__ int mid[6 * kGroup][64];
    __shared__ __attribute__((aligned(16))) int16_t outb[6 * kGroup][64];
    __shared__ uint16_t sdiv[2][64];
    __shared__ uint8_t spos[64];
    const int tid = threadIdx.x;
    if (tid < 64) {
        sdiv[0][tid] = q.div[0][tid];
        sdiv[1][tid] = q.div[1][tid];
        spos[tid] = q.pos[tid];
    }
    int h = h0, w = w0;
    const uint8_t* rgb = rgb0;
    int16_t* coef = coef0;
    int mcu_w = (w + 15) / 16, mcu_h = (h + 15) / 16, gw = (mcu_w + kGroup - 1) / kGroup;
    const long groups = MULTI ? total_groups : (long)n_images * mcu_h * gw;
    for (long g0 = blockIdx.x; g0 < groups; g0 += gridDim.x) {
        long g = g0;
        if (MULTI) {
            int lo = 0, hi = n_images - 1;
            while (lo < hi) {
                const int m = (lo + hi + 1) >> 1;
                if (items[m].group_start <= g0)
                    lo = m;
                else
                    hi = m - 1;
            }
            const lf_jpeg_item it = items[lo];
            h = it.h;
            w = it.w;
            rgb = rgb0 + it.rgb_off;
            coef = coef0 + it.coef_off;
            mcu_w = (w + 15) / 16;
            mcu_h = (h + 15) / 16;
            gw = (mcu_w + kGroup - 1) / kGroup;
            g = g0 - it.group_start;
        }
        const int gx = (int)(g % gw);
        const long t1 = g / gw;
        const int my = (int)(t1 % mcu_h);
        const size_t n = MULTI ? 0 : (size_t)(t1 / mcu_h);
        const int mx0 = gx * kGroup, nm = min(kGroup, mcu_w - mx0);
        __syncthreads();   // the previous pass is done with the planes; the tables are in place
        {   // ---- four pixels per thread: RGB -> Y, Cb, Cr (jccolor.c, SCALEBITS 16)
            const int row = tid >> 4, x = (tid & 15) * 4;
            if (x < 16 * nm) {
                unsigned wd[3];
                if (RAGGED) {   // right / bottom edge replicated
                    const uint8_t* rp = rgb + (n * h + min(16 * my + row, h - 1)) * (size_t)w * 3;
                    wd[0] = wd[1] = wd[2] = 0u;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint8_t* px = rp + (size_t)min(16 * mx0 + x + k, w - 1) * 3;
#pragma unroll
                        for (int c = 0; c < 3; ++c) wd[(3 * k + c) >> 2] |= (unsigned)px[c] << (8 * ((3 * k + c) & 3));
                    }
                } else {
                    const uint8_t* p = rgb + ((n * h + 16 * my + row) * (size_t)w + 16 * mx0 + x) * 3;
                    const unsigned* p4 = reinterpret_cast<const unsigned*>(p);   // 12-byte groups of a 4-pixel-aligned column
                    wd[0] = p4[0];
                    wd[1] = p4[1];
                    wd[2] = p4[2];
                }
                unsigned yy = 0, cb = 0, cr = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = (wd[(3 * k) >> 2] >> (8 * ((3 * k) & 3))) & 255;
                    const int gg = (wd[(3 * k + 1) >> 2] >> (8 * ((3 * k + 1) & 3))) & 255;
                    const int b = (wd[(3 * k + 2) >> 2] >> (8 * ((3 * k + 2) & 3))) & 255;
                    yy |= (unsigned)((19595 * r + 38470 * gg + 7471 * b + 32768) >> 16) << (8 * k);
                    cb |= (unsigned)((-11059 * r - 21709 * gg + 32768 * b + (128 << 16) + 32767) >> 16) << (8 * k);
                    cr |= (unsigned)((32768 * r - 27439 * gg - 5329 * b + (128 << 16) + 32767) >> 16) << (8 * k);
                }
                *reinterpret_cast<unsigned*>(&sy[row][x]) = yy;
                *reinterpret_cast<unsigned*>(&scb[row][x]) = cb;
                *reinterpret_cast<unsigned*>(&scr[row][x]) = cr;
            }
        }
        __syncthreads();
        for (int i = tid; i < 2 * 8 * 8 * kGroup; i += kT) {   // ---- h2v2_downsample, bias 1, 2, 1, 2 ...
            const int pl = i / (8 * 8 * kGroup), r0 = (i / (8 * kGroup)) & 7, c = i % (8 * kGroup);
            // below the image the last CHROMA row repeats (its second source row is already the replicated one)
            const int r = RAGGED ? min(r0, (h + 1) / 2 - 1 - 8 * my) : r0;
            const uint8_t(*P)[16 * kGroup] = pl ? scr : scb;
            sc[pl][r0][c] = (uint8_t)((P[2 * r][2 * c] + P[2 * r][2 * c + 1] + P[2 * r + 1][2 * c] + P[2 * r + 1][2 * c + 1] +
                                      1 + (c & 1)) >> 2);
        }
        __syncthreads();
        const int blk = tid >> 3, k8 = tid & 7, mi = blk / 6, b = blk - 6 * mi;
        const bool work = blk < 6 * nm;
        if (work) {   // ---- row pass
            int d[8];
            if (b < 4) {
                const uint8_t* s = &sy[8 * (b >> 1) + k8][16 * mi + 8 * (b & 1)];
#pragma unroll
                for (int k = 0; k < 8; ++k) d[k] = (int)s[k] - 128;
            } else {
                const uint8_t* s = &sc[b - 4][k8][8 * mi];
#pragma unroll
                for (int k = 0; k < 8; ++k) d[k] = (int)s[k] - 128;
            }
            fdct8<true>(d);
#pragma unroll
            for (int k = 0; k < 8; ++k) mid[blk][8 * k8 + k] = d[k];
        }
        __syncthreads();
        if (work) {   // ---- column pass, quantisation, zigzag
            int d[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) d[k] = mid[blk][8 * k + k8];
            fdct8<false>(d);
            const uint16_t* dv = sdiv[b < 4 ? 0 : 1];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int idx = 8 * k + k8, dq = dv[idx];
                const int a = d[k] < 0 ? -d[k] : d[k];
                const int v = (a + (dq >> 1)) / dq;
                outb[blk][spos[idx]] = (int16_t)(d[k] < 0 ? -v : v);
            }
        }
        __syncthreads();
        if (RAGGED) {
            if (tid < nm) {   // dummy luminance blocks of MCU tid, in buffer order Y00 Y01 Y10 Y11
                const int mx = mx0 + tid;
                const bool col1 = 2 * mx + 1 < (w + 7) / 8, row1 = 2 * my + 1 < (h + 7) / 8;
                int16_t(*b)[64] = &outb[6 * tid];
                auto dummy = [&](int blk, int16_t dc) {
                    for (int k = 1; k < 64; ++k) b[blk][k] = 0;
                    b[blk][0] = dc;
                };
                if (!col1) dummy(1, b[0][0]);
                if (row1) {
                    if (!col1) dummy(3, b[2][0]);
                } else {
                    dummy(2, b[1][0]);
                    dummy(3, b[1][0]);
                }
            }
            __syncthreads();
        }
        if (work) {   // 16 bytes per lane, 768 contiguous bytes per MCU
            int16_t* dst = coef + ((n * mcu_h + my) * (size_t)mcu_w + mx0) * (6 * 64);
            reinterpret_cast<lf::u32x4*>(dst)[tid] = reinterpret_cast<const lf::u32x4*>(&outb[0][0])[tid];
        }
    }
}

// ---------------------------------------------------------------------------
// Decoding: the pixel half of Image.open(path).convert("RGB") — jidctint.c jpeg_idct_islow with the
// dequantisation folded in, jdsample.c h2v2_fancy_upsample (the default triangle filter), jdcolor.c
// ycc_rgb_convert.  The Huffman decoding that comes before it is the host's (lf_jpeg_read_file).
// ---------------------------------------------------------------------------
// one 1-D pass of jpeg_idct_islow over m[0..7]
template <int SHIFT>
__device__ __forceinline__ void idct8(int* m) {
    int z2 = m[2], z3 = m[6];
    int z1 = (z2 + z3) * LF_FIX(0.541196100);
    const int tmp2 = z1 + z3 * (-LF_FIX(1.847759065));
    const int tmp3 = z1 + z2 * LF_FIX(0.765366865);
    z2 = m[0];
    z3 = m[4];
    const int tmp0 = (z2 + z3) << kCB, tmp1 = (z2 - z3) << kCB;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    int t0 = m[7], t1 = m[5], t2 = m[3], t3 = m[1];
    z1 = t0 + t3;
    z2 = t1 + t2;
    z3 = t0 + t2;
    int z4 = t1 + t3;
    const int z5 = (z3 + z4) * LF_FIX(1.175875602);
    t0 *= LF_FIX(0.298631336);
    t1 *= LF_FIX(2.053119869);
    t2 *= LF_FIX(3.072711026);
    t3 *= LF_FIX(1.501321110);
    z1 *= -LF_FIX(0.899976223);
    z2 *= -LF_FIX(2.562915447);
    z3 = z3 * (-LF_FIX(1.961570560)) + z5;
    z4 = z4 * (-LF_FIX(0.390180644)) + z5;
    t0 += z1 + z3;
    t1 += z2 + z4;
    t2 += z2 + z3;
    t3 += z1 + z4;
    m[0] = descale(tmp10 + t3, SHIFT);
    m[7] = descale(tmp10 - t3, SHIFT);
    m[1] = descale(tmp11 + t2, SHIFT);
    m[6] = descale(tmp11 - t2, SHIFT);
    m[2] = descale(tmp12 + t1, SHIFT);
    m[5] = descale(tmp12 - t1, SHIFT);
    m[3] = descale(tmp13 + t0, SHIFT);
    m[4] = descale(tmp13 - t0, SHIFT);
}

struct ZigPos {
    uint8_t pos[64];   // row-major index -> zigzag position
};

// coefficients (zigzag, quantised) of four MCUs at a time -> Y plane and the two half-size chroma planes
__global__ __launch_bounds__(kT) void jpeg_idct_kernel(const uint8_t* __restrict__ coef, size_t coef_stride,
                                                       const uint8_t* __restrict__ qtab, size_t qtab_stride,
                                                       uint8_t* __restrict__ yp, uint8_t* __restrict__ cbp,
                                                       uint8_t* __restrict__ crp, int h, int w, int n_images, ZigPos zp) {
    __shared__ __attribute__((aligned(16))) int16_t cin[6 * kGroup][64];
    __shared__ int mid[6 * kGroup][64];
    __shared__ uint16_t sq[2][64];
    __shared__ uint8_t spos[64];
    const int tid = threadIdx.x;
    if (tid < 64) spos[tid] = zp.pos[tid];
    const int mcu_w = w / 16, mcu_h = h / 16, gw = (mcu_w + kGroup - 1) / kGroup;
    const long groups = (long)n_images * mcu_h * gw;
    long cur_img = -1;
    for (long g = blockIdx.x; g < groups; g += gridDim.x) {
        const int gx = (int)(g % gw);
        const long t1 = g / gw;
        const int my = (int)(t1 % mcu_h);
        const long n = t1 / mcu_h;
        const int mx0 = gx * kGroup, nm = min(kGroup, mcu_w - mx0);
        __syncthreads();
        if (n != cur_img) {   // this image's two tables
            if (tid < 128) sq[tid >> 6][tid & 63] = reinterpret_cast<const uint16_t*>(qtab + (size_t)n * qtab_stride)[tid];
            cur_img = n;
        }
        const int blk = tid >> 3, k8 = tid & 7, mi = blk / 6, b = blk - 6 * mi;
        const bool work = blk < 6 * nm;
        if (work) {
            const int16_t* src = reinterpret_cast<const int16_t*>(coef + (size_t)n * coef_stride) +
                                 ((size_t)my * mcu_w + mx0) * (6 * 64);
            reinterpret_cast<lf::u32x4*>(&cin[0][0])[tid] = reinterpret_cast<const lf::u32x4*>(src)[tid];
        }
        __syncthreads();
        if (work) {   // column pass: column k8 of block blk
            int d[8];
            const uint16_t* q = sq[b < 4 ? 0 : 1];
#pragma unroll
            for (int k = 0; k < 8; ++k) d[k] = (int)cin[blk][spos[8 * k + k8]] * (int)q[8 * k + k8];
            idct8<kCB - kP1>(d);
#pragma unroll
            for (int k = 0; k < 8; ++k) mid[blk][8 * k + k8] = d[k];
        }
        __syncthreads();
        if (work) {   // row pass: row k8 of block blk -> eight samples
            int d[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) d[k] = mid[blk][8 * k8 + k];
            idct8<kCB + kP1 + 3>(d);
            unsigned lo = 0, hi = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                lo |= (unsigned)min(max(d[k] + 128, 0), 255) << (8 * k);
                hi |= (unsigned)min(max(d[k + 4] + 128, 0), 255) << (8 * k);
            }
            uint8_t* dst;
            if (b < 4)
                dst = yp + ((size_t)n * h + 16 * my + 8 * (b >> 1) + k8) * w + 16 * (mx0 + mi) + 8 * (b & 1);
            else
                dst = (b == 4 ? cbp : crp) + ((size_t)n * (h / 2) + 8 * my + k8) * (w / 2) + 8 * (mx0 + mi);
            *reinterpret_cast<uint2*>(dst) = make_uint2(lo, hi);
        }
    }
}

// one thread: four chroma samples of a row -> 2 x 8 output pixels (h2v2_fancy_upsample + ycc_rgb_convert)
__global__ __launch_bounds__(kT) void jpeg_upsample_rgb_kernel(const uint8_t* __restrict__ yp,
                                                               const uint8_t* __restrict__ cbp,
                                                               const uint8_t* __restrict__ crp,
                                                               uint8_t* __restrict__ rgb, int h, int w, size_t total) {
    const int cw = w / 2, ch = h / 2, qw = cw / 4;
    for (size_t t = (size_t)blockIdx.x * kT + threadIdx.x; t < total; t += (size_t)gridDim.x * kT) {
        const int jq = (int)(t % qw);
        const size_t r = t / qw;
        const int i = (int)(r % ch);
        const size_t n = r / ch;
        const int j0 = 4 * jq;
        const uint8_t* planes[2] = {cbp + n * (size_t)ch * cw, crp + n * (size_t)ch * cw};
        int up[2][2][8];   // [plane][output row v][output column]
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const uint8_t* c0 = planes[pl] + (size_t)i * cw;
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int inb = v == 0 ? max(i - 1, 0) : min(i + 1, ch - 1);
                const uint8_t* c1 = planes[pl] + (size_t)inb * cw;
                int cs[6];   // column sums of columns j0-1 .. j0+4 (the image's own edge stands in outside)
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int j = min(max(j0 - 1 + k, 0), cw - 1);
                    cs[k] = 3 * (int)c0[j] + (int)c1[j];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    up[pl][v][2 * k] = (3 * cs[k + 1] + cs[k] + 8) >> 4;
                    up[pl][v][2 * k + 1] = (3 * cs[k + 1] + cs[k + 2] + 7) >> 4;
                }
            }
        }
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const size_t row = n * (size_t)h + 2 * i + v;
            const uint2 yy = *reinterpret_cast<const uint2*>(yp + row * w + 8 * jq);
            unsigned o[6] = {0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int y = (int)(((k < 4 ? yy.x : yy.y) >> (8 * (k & 3))) & 255u);
                const int xb = up[0][v][k] - 128, xr = up[1][v][k] - 128;
                const int rr = min(max(y + ((91881 * xr + 32768) >> 16), 0), 255);
                const int bb = min(max(y + ((116130 * xb + 32768) >> 16), 0), 255);
                const int gg = min(max(y + ((-22554 * xb + 32768 - 46802 * xr) >> 16), 0), 255);
                o[(3 * k) >> 2] |= (unsigned)rr << (8 * ((3 * k) & 3));
                o[(3 * k + 1) >> 2] |= (unsigned)gg << (8 * ((3 * k + 1) & 3));
                o[(3 * k + 2) >> 2] |= (unsigned)bb << (8 * ((3 * k + 2) & 3));
            }
            unsigned* dst = reinterpret_cast<unsigned*>(rgb + (row * w + 8 * jq) * 3);
#pragma unroll
            for (int k = 0; k < 6; ++k) dst[k] = o[k];
        }
    }
}

// ---------------------------------------------------------------------------
// Entropy coding on the GPU (jchuff.c with the Annex K tables): one workgroup per image.  Every thread takes
// a contiguous run of the scan's blocks, (1) adds up their code lengths, (2) a workgroup scan turns the sums
// into bit offsets, (3) the thread writes its run's bits MSB-first into a zeroed word array — whole words
// with plain stores, the first and last word of a run (shared with the neighbours) with atomicOr —, (4) the
// last byte is padded with ones, 0xFF bytes are counted, scanned, and the stream is copied out with the
// 0x00 stuffing.  Output row: int32 length (or -1: did not fit), then the scan's bytes.
// ---------------------------------------------------------------------------
struct HuffTab {
    uint32_t dc[2][16];    // (code << 8) | length
    uint32_t ac[2][256];
};

constexpr int kEB = 256;

template <bool EMIT>
struct BitSink {
    unsigned long long acc = 0;
    int fill = 0;
    size_t w = 0;
    bool first = true;
    unsigned total = 0;
    uint32_t* words = nullptr;
    __device__ __forceinline__ void put(unsigned bits, int len) {
        if (!EMIT) {
            total += (unsigned)len;
            return;
        }
        acc |= (unsigned long long)bits << (64 - fill - len);
        fill += len;
        if (fill >= 32) {
            const uint32_t word = (uint32_t)(acc >> 32);
            if (first) {
                atomicOr(&words[w], word);
                first = false;
            } else {
                words[w] = word;
            }
            ++w;
            acc <<= 32;
            fill -= 32;
        }
    }
};

template <bool EMIT>
__device__ __forceinline__ void code_block_gpu(BitSink<EMIT>& bs, const int16_t* __restrict__ blk, int pred,
                                               const uint32_t* __restrict__ dc, const uint32_t* __restrict__ ac) {
    lf::u32x4 q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = reinterpret_cast<const lf::u32x4*>(blk)[i];
    auto at = [&](int k) -> int {
        const unsigned wd = q[k >> 3][(k >> 1) & 3];
        return (int)(short)((k & 1) ? wd >> 16 : wd & 0xffffu);
    };
    const int diff = at(0) - pred;
    int nb = diff ? 32 - __clz(diff < 0 ? -diff : diff) : 0;
    unsigned e = dc[nb];
    bs.put(((e >> 8) << nb) | ((unsigned)(diff + (diff >> 31)) & ((1u << nb) - 1u)), (int)(e & 255u) + nb);
    int run = 0;
#pragma unroll
    for (int k = 1; k < 64; ++k) {
        const int v = at(k);
        if (v == 0) {
            ++run;
        } else {
            while (run > 15) {
                bs.put(ac[0xF0] >> 8, (int)(ac[0xF0] & 255u));
                run -= 16;
            }
            nb = 32 - __clz(v < 0 ? -v : v);
            e = ac[(run << 4) | nb];
            bs.put(((e >> 8) << nb) | ((unsigned)(v + (v >> 31)) & ((1u << nb) - 1u)), (int)(e & 255u) + nb);
            run = 0;
        }
    }
    if (run) bs.put(ac[0] >> 8, (int)(ac[0] & 255u));
}

// exclusive scan of one value per thread over the workgroup; returns the exclusive prefix, `total` the sum
__device__ __forceinline__ unsigned wg_exclusive_scan(unsigned v, unsigned* lds, unsigned& total) {
    const int tid = threadIdx.x;
    lds[tid] = v;
    __syncthreads();
    for (int off = 1; off < kEB; off <<= 1) {
        const unsigned add = tid >= off ? lds[tid - off] : 0u;
        __syncthreads();
        lds[tid] += add;
        __syncthreads();
    }
    total = lds[kEB - 1];
    const unsigned excl = lds[tid] - v;
    __syncthreads();
    return excl;
}

__global__ __launch_bounds__(kEB) void jpeg_entropy_kernel(const uint8_t* __restrict__ coef_base, size_t coef_stride,
                                                           uint32_t* __restrict__ tmp, size_t tmp_words,
                                                           uint8_t* __restrict__ out, size_t out_stride, int nblocks0,
                                                           HuffTab tab, const lf_jpeg_item* __restrict__ items = nullptr) {
    __shared__ uint32_t sdc[2][16], sac[2][256];
    __shared__ unsigned scan[kEB];
    const int tid = threadIdx.x;
    const size_t n = blockIdx.x;
    // items: every image its own size and places (out_stride is then the room each image's scan may take)
    const int16_t* coef = items ? reinterpret_cast<const int16_t*>(coef_base) + items[n].coef_off
                                : reinterpret_cast<const int16_t*>(coef_base + n * coef_stride);
    const int nblocks = items ? items[n].nblocks : nblocks0;
    uint32_t* tw = tmp + n * tmp_words;
    uint8_t* orow = items ? out + items[n].out_off : out + n * out_stride;
    if (tid < 32) sdc[tid >> 4][tid & 15] = tab.dc[tid >> 4][tid & 15];
    for (int i = tid; i < 512; i += kEB) sac[i >> 8][i & 255] = tab.ac[i >> 8][i & 255];
    for (size_t i = tid; i < tmp_words; i += kEB) tw[i] = 0u;
    __syncthreads();
    const int per = (nblocks + kEB - 1) / kEB;
    const int b0 = min(tid * per, nblocks), b1 = min(b0 + per, nblocks);
    auto pred_of = [&](int b) -> int {   // DC of the previous block of the same component in scan order
        const int k6 = b % 6;
        const int pb = k6 == 0 ? b - 3 : (k6 < 4 ? b - 1 : b - 6);
        return pb >= 0 ? (int)coef[(size_t)pb * 64] : 0;
    };
    BitSink<false> count;
    for (int b = b0; b < b1; ++b) {
        const int t = (b % 6) < 4 ? 0 : 1;
        code_block_gpu<false>(count, coef + (size_t)b * 64, pred_of(b), sdc[t], sac[t]);
    }
    unsigned total_bits = 0;
    const unsigned start = wg_exclusive_scan(count.total, scan, total_bits);
    const size_t nbytes = ((size_t)total_bits + 7) / 8;
    if (nbytes > tmp_words * 4 || nbytes + 4 > out_stride) {   // uniform: the whole workgroup leaves
        if (tid == 0) *reinterpret_cast<int*>(orow) = -1;
        return;
    }
    BitSink<true> bs;
    bs.words = tw;
    bs.w = start >> 5;
    bs.fill = (int)(start & 31u);
    for (int b = b0; b < b1; ++b) {
        const int t = (b % 6) < 4 ? 0 : 1;
        code_block_gpu<true>(bs, coef + (size_t)b * 64, pred_of(b), sdc[t], sac[t]);
    }
    if (bs.fill > 0 && b1 > b0) atomicOr(&tw[bs.w], (uint32_t)(bs.acc >> 32));
    if (tid == 0 && (total_bits & 7u)) {   // pad the last byte with ones
        const unsigned pad = 8u - (total_bits & 7u), at = total_bits & 31u;
        atomicOr(&tw[total_bits >> 5], ((1u << pad) - 1u) << (32u - at - pad));
    }
    __threadfence_block();
    __syncthreads();
    // ---- 0xFF -> 0xFF 0x00
    const size_t chunk = (nbytes + kEB - 1) / kEB;
    const size_t i0 = min((size_t)tid * chunk, nbytes), i1 = min(i0 + chunk, nbytes);
    auto byte_at = [&](size_t i) -> unsigned {
        return (__hip_atomic_load(&tw[i >> 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (24 - 8 * (i & 3))) & 255u;
    };
    unsigned ff = 0;
    for (size_t i = i0; i < i1; ++i) ff += byte_at(i) == 0xFFu;
    unsigned total_ff = 0;
    const unsigned before = wg_exclusive_scan(ff, scan, total_ff);
    if (4 + nbytes + total_ff > out_stride) {
        if (tid == 0) *reinterpret_cast<int*>(orow) = -1;
        return;
    }
    uint8_t* dst = orow + 4 + i0 + before;
    for (size_t i = i0; i < i1; ++i) {
        const unsigned bv = byte_at(i);
        *dst++ = (uint8_t)bv;
        if (bv == 0xFFu) *dst++ = 0;
    }
    if (tid == 0) *reinterpret_cast<int*>(orow) = (int)(nbytes + total_ff);
}

}  // namespace

extern "C" void lf_jpeg_std_huffman(uint32_t* dc32, uint32_t* ac512);   // lf_jpeg_host.cpp

extern "C" {

size_t lf_jpeg_entropy_workspace(int n, size_t out_stride) {
    return n > 0 ? (size_t)n * ((out_stride + 3) / 4) * 4 : 0;
}

int lf_jpeg_entropy_u8(const void* coef, size_t coef_stride, uint8_t* out, size_t out_stride, int n, int h, int w,
                       void* workspace, size_t ws_bytes, lf_stream_t stream) {
    LF_REQUIRE(coef && out && workspace, "lf_jpeg_entropy: null buffer");
    LF_REQUIRE(n > 0 && h > 0 && w > 0, "lf_jpeg_entropy: bad dims n=%d h=%d w=%d", n, h, w);
    const size_t mcus = (size_t)((h + 15) / 16) * ((w + 15) / 16);
    LF_REQUIRE(coef_stride >= mcus * 768 && coef_stride % 16 == 0 && (reinterpret_cast<size_t>(coef) & 15) == 0,
               "lf_jpeg_entropy: coefficients must be 16-byte aligned with a stride >= 768 bytes per MCU");
    LF_REQUIRE(out_stride >= 1024 && out_stride % 4 == 0 && (reinterpret_cast<size_t>(out) & 3) == 0,
               "lf_jpeg_entropy: output rows are 4-byte aligned and at least 1 KiB");
    LF_REQUIRE(ws_bytes >= lf_jpeg_entropy_workspace(n, out_stride) && (reinterpret_cast<size_t>(workspace) & 3) == 0,
               "lf_jpeg_entropy: workspace too small");
    LF_REQUIRE(n <= 1 << 20, "lf_jpeg_entropy: batch too large");
    static const HuffTab tab = []() {
        HuffTab t;
        lf_jpeg_std_huffman(&t.dc[0][0], &t.ac[0][0]);
        return t;
    }();
    const int nblocks = (int)mcus * 6;
    jpeg_entropy_kernel<<<n, kEB, 0, lf::as_stream(stream)>>>(static_cast<const uint8_t*>(coef), coef_stride,
                                                             static_cast<uint32_t*>(workspace), (out_stride + 3) / 4, out,
                                                             out_stride, nblocks, tab);
    return lf::check_launch("lf_jpeg_entropy");
}

long lf_jpeg_fdct_groups(int h, int w) {
    if (h <= 0 || w <= 0) return 0;
    return (long)((h + 15) / 16) * (((w + 15) / 16 + kGroup - 1) / kGroup);
}

int lf_jpeg_fdct_quant_items_u8(const uint8_t* rgb_base, int16_t* coef_base, const lf_jpeg_item* items, int n,
                                long total_groups, int quality, lf_stream_t stream) {
    LF_REQUIRE(rgb_base && coef_base && items, "lf_jpeg_fdct_quant_items: null buffer");
    LF_REQUIRE(n > 0 && total_groups > 0, "lf_jpeg_fdct_quant_items: nothing to do");
    LF_REQUIRE(quality >= 1 && quality <= 100, "lf_jpeg_fdct_quant_items: quality %d", quality);
    LF_REQUIRE((reinterpret_cast<size_t>(coef_base) & 15) == 0, "lf_jpeg_fdct_quant_items: coef must be 16-byte aligned");
    JpegQuant q;
    uint8_t tabs[2][64];
    lf_jpeg_quant_tables(quality, tabs[0], tabs[1]);
    static const uint8_t natural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                        41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                        30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    for (int i = 0; i < 64; ++i) {
        q.div[0][i] = (uint16_t)(tabs[0][i] << 3);
        q.div[1][i] = (uint16_t)(tabs[1][i] << 3);
        q.pos[natural[i]] = (uint8_t)i;
    }
    const unsigned grid = (unsigned)(total_groups < 256 * 16 ? total_groups : 256 * 16);
    jpeg_fdct_quant_kernel<true, true><<<grid, kT, 0, lf::as_stream(stream)>>>(rgb_base, coef_base, 0, 0, n, q, items,
                                                                               total_groups);
    return lf::check_launch("lf_jpeg_fdct_quant_items");
}

int lf_jpeg_entropy_items_u8(const void* coef_base, const lf_jpeg_item* items, uint8_t* out_base, size_t out_room, int n,
                             void* workspace, size_t ws_bytes, lf_stream_t stream) {
    LF_REQUIRE(coef_base && items && out_base && workspace, "lf_jpeg_entropy_items: null buffer");
    LF_REQUIRE(n > 0 && n <= 1 << 20, "lf_jpeg_entropy_items: bad batch size %d", n);
    LF_REQUIRE(out_room >= 1024 && out_room % 4 == 0 && (reinterpret_cast<size_t>(out_base) & 3) == 0,
               "lf_jpeg_entropy_items: output places are 4-byte aligned and at least 1 KiB");
    LF_REQUIRE(ws_bytes >= lf_jpeg_entropy_workspace(n, out_room) && (reinterpret_cast<size_t>(workspace) & 3) == 0,
               "lf_jpeg_entropy_items: workspace too small");
    static const HuffTab tab = []() {
        HuffTab t;
        lf_jpeg_std_huffman(&t.dc[0][0], &t.ac[0][0]);
        return t;
    }();
    jpeg_entropy_kernel<<<n, kEB, 0, lf::as_stream(stream)>>>(static_cast<const uint8_t*>(coef_base), 0,
                                                             static_cast<uint32_t*>(workspace), (out_room + 3) / 4,
                                                             out_base, out_room, 0, tab, items);
    return lf::check_launch("lf_jpeg_entropy_items");
}

size_t lf_jpeg_decode_workspace(int n, int h, int w) {
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    const size_t px = (size_t)n * h * w;
    return ((px + 255) & ~(size_t)255) + 2 * ((px / 4 + 255) & ~(size_t)255);
}

int lf_jpeg_idct_rgb_u8(const void* coef, size_t coef_stride, const void* qtab, size_t qtab_stride, uint8_t* rgb,
                        int n, int h, int w, void* workspace, size_t ws_bytes, lf_stream_t stream) {
    LF_REQUIRE(coef && qtab && rgb && workspace, "lf_jpeg_idct_rgb: null buffer");
    LF_REQUIRE(n > 0 && h > 0 && w > 0, "lf_jpeg_idct_rgb: bad dims n=%d h=%d w=%d", n, h, w);
    LF_REQUIRE(h % 16 == 0 && w % 16 == 0, "lf_jpeg_idct_rgb: whole 16x16 MCUs only (%d x %d)", h, w);
    LF_REQUIRE(coef_stride >= (size_t)h * w * 3 && coef_stride % 16 == 0 && (reinterpret_cast<size_t>(coef) & 15) == 0,
               "lf_jpeg_idct_rgb: coefficients must be 16-byte aligned, stride a multiple of 16 and >= 3*h*w bytes");
    LF_REQUIRE(qtab_stride >= 256 && qtab_stride % 2 == 0 && (reinterpret_cast<size_t>(qtab) & 1) == 0,
               "lf_jpeg_idct_rgb: tables are 128 uint16 per image");
    LF_REQUIRE((reinterpret_cast<size_t>(rgb) & 3) == 0, "lf_jpeg_idct_rgb: rgb must be 4-byte aligned");
    LF_REQUIRE(ws_bytes >= lf_jpeg_decode_workspace(n, h, w) && (reinterpret_cast<size_t>(workspace) & 15) == 0,
               "lf_jpeg_idct_rgb: workspace too small or misaligned");
    static const uint8_t natural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                        41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                        30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    ZigPos zp;
    for (int i = 0; i < 64; ++i) zp.pos[natural[i]] = (uint8_t)i;
    const size_t px = (size_t)n * h * w;
    uint8_t* yp = static_cast<uint8_t*>(workspace);
    uint8_t* cbp = yp + ((px + 255) & ~(size_t)255);
    uint8_t* crp = cbp + ((px / 4 + 255) & ~(size_t)255);
    hipStream_t s = lf::as_stream(stream);
    const long groups = (long)n * (h / 16) * ((w / 16 + kGroup - 1) / kGroup);
    const unsigned grid = (unsigned)(groups < 256 * 16 ? groups : 256 * 16);
    jpeg_idct_kernel<<<grid, kT, 0, s>>>(static_cast<const uint8_t*>(coef), coef_stride, static_cast<const uint8_t*>(qtab),
                                         qtab_stride, yp, cbp, crp, h, w, n, zp);
    const size_t total = (size_t)n * (h / 2) * (w / 8);
    jpeg_upsample_rgb_kernel<<<lf::stream_grid(total, kT, 256 * 32), kT, 0, s>>>(yp, cbp, crp, rgb, h, w, total);
    return lf::check_launch("lf_jpeg_idct_rgb");
}

int lf_jpeg_fdct_quant_u8(const uint8_t* rgb, int16_t* coef, int n, int h, int w, int quality,
                          lf_stream_t stream) {
    LF_REQUIRE(rgb && coef, "lf_jpeg_fdct_quant: null buffer");
    LF_REQUIRE(n > 0 && h > 0 && w > 0, "lf_jpeg_fdct_quant: bad dims n=%d h=%d w=%d", n, h, w);
    LF_REQUIRE(h <= 65535 && w <= 65535, "lf_jpeg_fdct_quant: JPEG dimensions are 16-bit");
    LF_REQUIRE(quality >= 1 && quality <= 100, "lf_jpeg_fdct_quant: quality %d", quality);
    const bool ragged = h % 16 != 0 || w % 16 != 0;
    LF_REQUIRE((ragged || (reinterpret_cast<size_t>(rgb) & 3) == 0) && (reinterpret_cast<size_t>(coef) & 15) == 0,
               "lf_jpeg_fdct_quant: rgb must be 4-byte and coef 16-byte aligned");
    JpegQuant q;
    uint8_t tabs[2][64];
    lf_jpeg_quant_tables(quality, tabs[0], tabs[1]);
    static const uint8_t natural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                        41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                        30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    for (int i = 0; i < 64; ++i) {
        q.div[0][i] = (uint16_t)(tabs[0][i] << 3);
        q.div[1][i] = (uint16_t)(tabs[1][i] << 3);
        q.pos[natural[i]] = (uint8_t)i;
    }
    const long groups = (long)n * ((h + 15) / 16) * (((w + 15) / 16 + kGroup - 1) / kGroup);
    const unsigned grid = (unsigned)(groups < 256 * 16 ? groups : 256 * 16);
    if (ragged)
        jpeg_fdct_quant_kernel<true><<<grid, kT, 0, lf::as_stream(stream)>>>(rgb, coef, h, w, n, q);
    else
        jpeg_fdct_quant_kernel<false><<<grid, kT, 0, lf::as_stream(stream)>>>(rgb, coef, h, w, n, q);
    return lf::check_launch("lf_jpeg_fdct_quant");
}

}  // extern "C"
