// libleafhip — geometric augmentation kernels with Pillow (libImaging) semantics.
//
// Coordinates and the bicubic polynomial are evaluated in IEEE double in the same
// operation order as libImaging/Geometry.c; this file is compiled with
// -ffp-contract=off so that no multiply-add is fused (a fused FMA rounds once where C
// rounds twice and would break bit-exactness).  The resampler is pure integer.
#include <algorithm>

#include "lf_common.h"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ int pil_floor(double v) { return v < 0.0 ? (int)floor(v) : (int)v; }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Geometry.c BICUBIC(v, v1, v2, v3, v4, d)
__device__ __forceinline__ double bicubic(double v1, double v2, double v3, double v4, double d) {
    const double p1 = v2;
    const double p2 = -v1 + v3;
    const double p3 = 2 * (v1 - v2) + v3 - v4;
    const double p4 = -v1 + v2 - v3 + v4;
    return p1 + d * (p2 + d * (p3 + d * p4));
}

// One thread per output pixel (3 channels).  grid = (ceil(h*w/256), n).
// The 4x4 footprint is fetched with ONE unaligned 12-byte load per footprint row (4 pixels x
// RGB; gfx950 runs in unaligned-access mode) instead of 48 byte gathers — the byte-gather
// version was bound by the texture-address unit, not by HBM or the f64 pipe.  Two exact
// shortcuts trim the double-precision work without changing a bit of the result:
//   * a6 == a7 == 0  ->  the perspective denominator is exactly 1.0 and x/1.0 == x;
//   * d == 0.0       ->  BICUBIC(v1..v4, 0) = v2 + 0*(...) = v2 exactly, so an axis that is
//                        not resampled (shear along the other axis) costs no polynomial.
struct __attribute__((packed, aligned(1))) Row12 {
    uint32_t a, b, c;
};

__device__ __forceinline__ double bicubic_z(double v1, double v2, double v3, double v4, double d) {
    return d == 0.0 ? v2 : bicubic(v1, v2, v3, v4, d);
}

// the three channels of one footprint row, horizontally interpolated
__device__ __forceinline__ void hrow(const uint8_t* __restrict__ row, int x, int w, double dx,
                                     double* v) {
    if (x >= 0 && x + 3 < w) {  // four consecutive pixels: one 12-byte load
        const Row12 q = *reinterpret_cast<const Row12*>(row + x * 3);
        const unsigned wd[3] = {q.a, q.b, q.c};
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            // pixel k channel b = byte 3k+b
            const double p0 = (double)((wd[(b) >> 2] >> (8 * ((b) & 3))) & 0xffu);
            const double p1 = (double)((wd[(3 + b) >> 2] >> (8 * ((3 + b) & 3))) & 0xffu);
            const double p2 = (double)((wd[(6 + b) >> 2] >> (8 * ((6 + b) & 3))) & 0xffu);
            const double p3 = (double)((wd[(9 + b) >> 2] >> (8 * ((9 + b) & 3))) & 0xffu);
            v[b] = bicubic_z(p0, p1, p2, p3, dx);
        }
    } else {  // clamped columns at the image border
        const int x0 = clampi(x, 0, w - 1) * 3, x1 = clampi(x + 1, 0, w - 1) * 3,
                  x2 = clampi(x + 2, 0, w - 1) * 3, x3 = clampi(x + 3, 0, w - 1) * 3;
#pragma unroll
        for (int b = 0; b < 3; ++b)
            v[b] = bicubic_z(row[x0 + b], row[x1 + b], row[x2 + b], row[x3 + b], dx);
    }
}

// One thread per output pixel (general maps: shear, true perspective).
// one output pixel of a general map (shear, true perspective, anything)
__device__ __forceinline__ void warp_pixel(const uint8_t* __restrict__ src, uint8_t* __restrict__ o,
                                           const double* __restrict__ a, bool divide, int h, int w,
                                           int ox, int oy) {
    const double xi = ox + 0.5, yi = oy + 0.5;
    double xin = a[0] * xi + a[1] * yi + a[2];
    double yin = a[3] * xi + a[4] * yi + a[5];
    if (divide) {
        xin = xin / (a[6] * xi + a[7] * yi + 1);
        yin = yin / (a[6] * xi + a[7] * yi + 1);
    }
    if (xin < 0.0 || xin >= w || yin < 0.0 || yin >= h) {
        o[0] = 0;
        o[1] = 0;
        o[2] = 0;
        return;
    }
    xin -= 0.5;
    yin -= 0.5;
    int x = pil_floor(xin), y = pil_floor(yin);
    const double dx = xin - x, dy = yin - y;
    x--;
    y--;
    const bool ok1 = (y + 1 >= 0 && y + 1 < h), ok2 = (y + 2 >= 0 && y + 2 < h),
               ok3 = (y + 3 >= 0 && y + 3 < h);
    const int yr0 = clampi(y, 0, h - 1);
    double v[3];
    if (dy == 0.0) {
        // BICUBIC(v1, v2, v3, v4, 0) = v2: only footprint row y+1 (or its fallback) matters
        hrow(src + (size_t)(ok1 ? y + 1 : yr0) * w * 3, x, w, dx, v);
    } else if (dx == 0.0) {
        // no horizontal resampling (a shear along y): every footprint row contributes exactly its
        // pixel at column x+1 (clamped like hrow's border path), so four pixels are fetched
        // instead of sixteen; the polynomial's p2..p4 are sums of small integers, formed
        // exactly in int before the one conversion to double each.
        const int col = clampi(x + 1, 0, w - 1) * 3;
        const uint8_t* r1 = src + (size_t)yr0 * w * 3 + col;
        const uint8_t* r2 = ok1 ? src + (size_t)(y + 1) * w * 3 + col : r1;
        const uint8_t* r3 = ok2 ? src + (size_t)(y + 2) * w * 3 + col : r2;
        const uint8_t* r4 = ok3 ? src + (size_t)(y + 3) * w * 3 + col : r3;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int q1 = r1[b], q2 = r2[b], q3 = r3[b], q4 = r4[b];
            const double p1 = (double)q2, p2 = (double)(q3 - q1), p3 = (double)(2 * (q1 - q2) + q3 - q4),
                         p4 = (double)(-q1 + q2 - q3 + q4);
            v[b] = p1 + dy * (p2 + dy * (p3 + dy * p4));
        }
    } else {
        double v1[3], v2[3], v3[3], v4[3];
        hrow(src + (size_t)yr0 * w * 3, x, w, dx, v1);
        if (ok1) hrow(src + (size_t)(y + 1) * w * 3, x, w, dx, v2);
        if (ok2) hrow(src + (size_t)(y + 2) * w * 3, x, w, dx, v3);
        if (ok3) hrow(src + (size_t)(y + 3) * w * 3, x, w, dx, v4);
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const double u2 = ok1 ? v2[b] : v1[b];
            const double u3 = ok2 ? v3[b] : u2;
            const double u4 = ok3 ? v4[b] : u3;
            v[b] = bicubic(v1[b], u2, u3, u4, dy);
        }
    }
#pragma unroll
    for (int b = 0; b < 3; ++b) o[b] = v[b] <= 0.0 ? 0 : (v[b] >= 255.0 ? 255 : (uint8_t)v[b]);
}

__global__ __launch_bounds__(kBlock) void warp_bicubic_kernel(const uint8_t* __restrict__ in,
                                                              uint8_t* __restrict__ out,
                                                              const double* __restrict__ coeffs,
                                                              int perspective, int h, int w) {
    // plain block order: the XCD-aware order that helps the nearest-neighbour rotate measured 2 %
    // slower here (this kernel is bound by its double-precision arithmetic)
    const unsigned n = blockIdx.y, bx = blockIdx.x;
    const double* a = coeffs + (size_t)n * 8;
    const bool divide = perspective && !(a[6] == 0.0 && a[7] == 0.0);
    const uint8_t* src = in + (size_t)n * h * w * 3;
    uint8_t* dst = out + (size_t)n * h * w * 3;
    const int total = h * w;
    for (int t = bx * kBlock + threadIdx.x; t < total; t += gridDim.x * kBlock) {
        const int oy = t / w, ox = t - oy * w;
        warp_pixel(src, dst + (size_t)t * 3, a, divide, h, w, ox, oy);
    }
}

// Axis-aligned maps (a1 == a3 == 0 and no perspective divide: the reference's "skew" is such a
// scale about a corner) are separable: the source column / dx depend on the output column only,
// the source row / dy on the output row only.  A workgroup makes a kTX x kTY output tile in two
// phases through LDS: every needed (source row, output column) is interpolated horizontally
// ONCE (~1.3 row interpolations per output pixel instead of 4), then each output pixel combines
// four of those values vertically.  Same hrow / BICUBIC arithmetic in double, same bits.
constexpr int kTX = 32, kTY = 32, kTileRows = 48;

__global__ __launch_bounds__(kBlock) void warp_bicubic_tile_kernel(const uint8_t* __restrict__ in,
                                                                   uint8_t* __restrict__ out,
                                                                   const double* __restrict__ coeffs,
                                                                   int perspective, int h, int w, int n_images) {
    __shared__ double Hs[kTileRows][kTX][3];
    __shared__ double sdx[kTX], sdy[kTY];
    __shared__ int sx[kTX], sy[kTY];
    __shared__ int svx[kTX], svy[kTY];
    __shared__ int srange[2];
    const lf::TileId tile = lf::xcd_tile((w + kTX - 1) / kTX, (h + kTY - 1) / kTY, n_images);
    if (!tile.ok) return;
    const unsigned n = (unsigned)tile.n;
    const double* a = coeffs + (size_t)n * 8;
    const bool divide = perspective && !(a[6] == 0.0 && a[7] == 0.0);
    const bool axis = !divide && a[1] == 0.0 && a[3] == 0.0;
    const uint8_t* src = in + (size_t)n * h * w * 3;
    uint8_t* dst = out + (size_t)n * h * w * 3;
    const int x0 = tile.tx * kTX, y0 = tile.ty * kTY;
    const int tid = threadIdx.x;
    if (axis) {
        if (tid < kTX) {  // per output column: source column, dx, inside?
            const double xi = (x0 + tid) + 0.5;
            double xin = a[0] * xi + a[1] * 0.5 + a[2];  // a1 == 0: the y term adds an exact zero
            const bool ok = !(xin < 0.0 || xin >= w);
            xin -= 0.5;
            const int x = pil_floor(xin);
            sdx[tid] = xin - x;
            sx[tid] = x - 1;
            svx[tid] = ok && x0 + tid < w;
        } else if (tid < kTX + kTY) {  // per output row
            const int k = tid - kTX;
            const double yi = (y0 + k) + 0.5;
            double yin = a[3] * 0.5 + a[4] * yi + a[5];
            const bool ok = !(yin < 0.0 || yin >= h);
            yin -= 0.5;
            const int y = pil_floor(yin);
            sdy[k] = yin - y;
            sy[k] = y - 1;
            svy[k] = ok && y0 + k < h;
        }
        __syncthreads();
        if (tid == 0) {  // source rows any valid output row of the tile touches
            int lo = 0x7fffffff, hi = -1;
            for (int k = 0; k < kTY; ++k)
                if (svy[k]) {
                    lo = min(lo, clampi(sy[k], 0, h - 1));
                    hi = max(hi, clampi(sy[k] + 3, 0, h - 1));
                }
            srange[0] = lo;
            srange[1] = hi;
        }
        __syncthreads();
    }
    const int rlo = axis ? srange[0] : 0, rhi = axis ? srange[1] : -1;
    const int nrows = rhi - rlo + 1;
    if (!axis || nrows > kTileRows) {
        // general map inside a hinted batch, or a strong down-scale: per-pixel path
        for (int i = tid; i < kTX * kTY; i += kBlock) {
            const int ox = x0 + i % kTX, oy = y0 + i / kTX;
            if (ox < w && oy < h) warp_pixel(src, dst + ((size_t)oy * w + ox) * 3, a, divide, h, w, ox, oy);
        }
        return;
    }
    // phase 1: horizontal interpolation, once per (source row, output column)
    for (int i = tid; i < nrows * kTX; i += kBlock) {
        const int r = i / kTX, c = i - r * kTX;
        if (svx[c]) hrow(src + (size_t)(rlo + r) * w * 3, sx[c], w, sdx[c], Hs[r][c]);
    }
    __syncthreads();
    // phase 2: vertical interpolation
    for (int i = tid; i < kTX * kTY; i += kBlock) {
        const int ty = i / kTX, tx = i - ty * kTX;
        const int ox = x0 + tx, oy = y0 + ty;
        if (ox >= w || oy >= h) continue;
        uint8_t* o = dst + ((size_t)oy * w + ox) * 3;
        if (!svx[tx] || !svy[ty]) {
            o[0] = 0;
            o[1] = 0;
            o[2] = 0;
            continue;
        }
        const int y = sy[ty];
        const double dy = sdy[ty];
        const bool ok1 = (y + 1 >= 0 && y + 1 < h), ok2 = (y + 2 >= 0 && y + 2 < h),
                   ok3 = (y + 3 >= 0 && y + 3 < h);
        const int i1 = clampi(y, 0, h - 1) - rlo;
        const int i2 = ok1 ? y + 1 - rlo : i1;
        const int i3 = ok2 ? y + 2 - rlo : i2;
        const int i4 = ok3 ? y + 3 - rlo : i3;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            // dy == 0: BICUBIC(.., 0) = v2 + 0*(..) = v2 exactly, the value the per-pixel path takes
            const double v = dy == 0.0 ? Hs[i2][tx][b]
                                       : bicubic(Hs[i1][tx][b], Hs[i2][tx][b], Hs[i3][tx][b], Hs[i4][tx][b], dy);
            o[b] = v <= 0.0 ? 0 : (v >= 255.0 ? 255 : (uint8_t)v);
        }
    }
}

struct __attribute__((packed, aligned(1))) Pix4 {
    uint32_t v;
};

// Geometry.c affine_fixed: xx = a2 + a1*y + a0*x in wrapping int32 16.16.
// Ragged batch: image n writes ohw[n] = (oh, ow) pixels at byte offset out_off[n].
__global__ __launch_bounds__(kBlock) void affine_nearest_kernel(const uint8_t* __restrict__ in,
                                                                uint8_t* __restrict__ out,
                                                                const int32_t* __restrict__ fix6,
                                                                const int32_t* __restrict__ ohw,
                                                                const int64_t* __restrict__ out_off,
                                                                int h, int w, unsigned fill) {
    // One thread = four consecutive output pixels = three whole dwords of the packed RGB
    // output: every pixel is computed once, the row/column split costs one integer division
    // per group (the following pixels step along the row), stores are whole dwords.
    // out_off[n] is 16-byte aligned and each image's region is padded to 16 bytes.
    const lf::Block2 blk = lf::xcd_block2();  // consecutive blocks of an image under one L2
    const unsigned n = blk.y;
    const int32_t* a = fix6 + (size_t)n * 6;
    const unsigned a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4], a5 = a[5];
    const int oh = ohw[2 * n], ow = ohw[2 * n + 1];
    const uint8_t* src = in + (size_t)n * h * w * 3;
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + out_off[n]);
    const int total = oh * ow;            // pixels
    const int nd = (total * 3 + 3) / 4;   // dwords that hold at least one pixel byte
    const int ng = (total + 3) / 4;       // groups of four pixels
    const int last = h * w - 1;
    // row / column of the group's first pixel: float reciprocal + one correction step while the
    // pixel index is exact in float32 (an integer division is a long multiply-and-correct sequence)
    const bool small = total < (1 << 22);  // quotient error <= 2^22 * 1.2e-7 < 1: at most one step off
    const float inv_ow = 1.0f / (float)ow;
    for (int g = blk.x * kBlock + threadIdx.x; g < ng; g += gridDim.x * kBlock) {
        const int p0 = 4 * g;
        unsigned oy, ox;
        if (small) {
            oy = (unsigned)((float)p0 * inv_ow);
            int rem = p0 - __mul24((int)oy, ow);
            if (rem < 0) {
                --oy;
                rem += ow;
            } else if (rem >= ow) {
                ++oy;
                rem -= ow;
            }
            ox = (unsigned)rem;
        } else {
            oy = (unsigned)p0 / (unsigned)ow;
            ox = (unsigned)p0 - oy * (unsigned)ow;
        }
        // unsigned arithmetic == two's-complement wrap of the C int accumulation
        unsigned xx = a2 + a1 * oy + a0 * ox, yy = a5 + a4 * oy + a3 * ox;
        unsigned px[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned v = fill * 0x010101u;
            const int xin = (int)xx >> 16, yin = (int)yy >> 16;
            if (p0 + k < total && xin >= 0 && xin < w && yin >= 0 && yin < h) {
                const int sp = __mul24(yin, w) + xin;  // h, w < 32768
                const uint8_t* s = src + (unsigned)sp * 3u;
                if (sp < last) {  // 4-byte unaligned load stays inside the image
                    v = reinterpret_cast<const Pix4*>(s)->v & 0xffffffu;
                } else {
                    v = s[0] | s[1] << 8 | s[2] << 16;
                }
            }
            px[k] = v;
            // next pixel of the row, or the first of the next row
            if (++ox == (unsigned)ow) {
                ox = 0;
                ++oy;
                xx = a2 + a1 * oy;
                yy = a5 + a4 * oy;
            } else {
                xx += a0;
                yy += a3;
            }
        }
        const int d0 = 3 * g;
        if (d0 < nd) dst[d0] = px[0] | px[1] << 24;
        if (d0 + 1 < nd) dst[d0 + 1] = px[1] >> 8 | px[2] << 16;
        if (d0 + 2 < nd) dst[d0 + 2] = px[2] >> 16 | px[3] << 8;
    }
}

// The same transform for source images that fit a CU's LDS (224 x 224 x 3 = 147 KB does): one workgroup per image
// copies the whole source into LDS with aligned 16-byte loads — every source byte crosses the memory pipe ONCE, as
// part of a full line — and the per-pixel gathers go to LDS.  The global-memory version above issues one unaligned
// 4-byte gather per output pixel and is bound by the texture addresser (TA_BUSY 80 %, 2.25x the algorithmic bytes
// before the XCD-aware block order); here the vector memory pipe only sees the streaming copy in and the coalesced
// dword stores out.  Same arithmetic, same bytes (tests compare both with the oracle).
constexpr int kRotT = 1024;

__global__ __launch_bounds__(kRotT) void affine_nearest_lds_kernel(const uint8_t* __restrict__ in,
                                                                   uint8_t* __restrict__ out,
                                                                   const int32_t* __restrict__ fix6,
                                                                   const int32_t* __restrict__ ohw,
                                                                   const int64_t* __restrict__ out_off, int n_images,
                                                                   int h, int w, unsigned fill) {
    extern __shared__ __attribute__((aligned(16))) uint8_t simg[];
    const int nbytes = h * w * 3;   // a multiple of 16 (checked on the host)
    for (int n = blockIdx.x; n < n_images; n += gridDim.x) {
        const lf::u32x4* s16 = reinterpret_cast<const lf::u32x4*>(in + (size_t)n * nbytes);
        __syncthreads();   // the previous image's gathers are done
        for (int i = threadIdx.x; i < nbytes / 16; i += kRotT) reinterpret_cast<lf::u32x4*>(simg)[i] = s16[i];
        const int32_t* a = fix6 + (size_t)n * 6;
        const unsigned a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4], a5 = a[5];
        const int oh = ohw[2 * n], ow = ohw[2 * n + 1];
        uint32_t* dst = reinterpret_cast<uint32_t*>(out + out_off[n]);
        const int total = oh * ow;            // pixels
        const int nd = (total * 3 + 3) / 4;   // dwords that hold at least one pixel byte
        const int ng = (total + 3) / 4;       // groups of four pixels
        const bool small = total < (1 << 22);
        const float inv_ow = 1.0f / (float)ow;
        __syncthreads();
        for (int g = threadIdx.x; g < ng; g += kRotT) {
            const int p0 = 4 * g;
            unsigned oy, ox;
            if (small) {
                oy = (unsigned)((float)p0 * inv_ow);
                int rem = p0 - __mul24((int)oy, ow);
                if (rem < 0) {
                    --oy;
                    rem += ow;
                } else if (rem >= ow) {
                    ++oy;
                    rem -= ow;
                }
                ox = (unsigned)rem;
            } else {
                oy = (unsigned)p0 / (unsigned)ow;
                ox = (unsigned)p0 - oy * (unsigned)ow;
            }
            unsigned xx = a2 + a1 * oy + a0 * ox, yy = a5 + a4 * oy + a3 * ox;
            unsigned px[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                unsigned v = fill * 0x010101u;
                const int xin = (int)xx >> 16, yin = (int)yy >> 16;
                if (p0 + k < total && xin >= 0 && xin < w && yin >= 0 && yin < h) {
                    const uint8_t* sp = simg + (unsigned)(__mul24(yin, w) + xin) * 3u;
                    v = sp[0] | sp[1] << 8 | sp[2] << 16;
                }
                px[k] = v;
                if (++ox == (unsigned)ow) {
                    ox = 0;
                    ++oy;
                    xx = a2 + a1 * oy;
                    yy = a5 + a4 * oy;
                } else {
                    xx += a0;
                    yy += a3;
                }
            }
            const int d0 = 3 * g;
            if (d0 < nd) dst[d0] = px[0] | px[1] << 24;
            if (d0 + 1 < nd) dst[d0 + 1] = px[1] >> 8 | px[2] << 16;
            if (d0 + 2 < nd) dst[d0 + 2] = px[2] >> 16 | px[3] << 8;
        }
    }
}

// Resample.c ImagingResampleHorizontal_8bpc / Vertical_8bpc: ss = 1<<21 + sum px*k; clip8(ss>>22).
constexpr int kPrec = 22;

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= kPrec;
    // Keep the shift and the clamp apart: hipcc 7.2 (clang 22) folds "two clamped shifts packed
    // into 16 bits" into gfx950's v_ashr_pk_u8_i32 and then ORs the other bytes onto its result as
    // if the upper half of the destination were zero — the instruction leaves it unchanged, so
    // bytes 2 and 3 of a packed dword come out as whatever the register held before.
    asm volatile("" : "+v"(v));
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// pixel (0..255) x coefficient (22-bit fixed point, |k| < 2^23: checked on the host) as a
// v_mad_i32_i24 / SDWA v_mul_i32_i24 (the byte extraction folds into the operand select).
__device__ __forceinline__ int mac24(int acc, unsigned px, int k) { return acc + __mul24((int)px, k); }

// horizontal: in [n][h][w][3] -> tmp [n][h][ow][3]; one thread per (y, ox).  The taps are
// fetched four pixels (12 bytes, one unaligned load) at a time.
__global__ __launch_bounds__(kBlock) void resample_h_kernel(const uint8_t* __restrict__ in,
                                                            uint8_t* __restrict__ tmp, int h, int w,
                                                            int ow, const int32_t* __restrict__ bounds,
                                                            const int32_t* __restrict__ kk, int ks,
                                                            int per_image) {
    const unsigned n = blockIdx.y;
    const int32_t* bnd = bounds + (per_image ? (size_t)n * ow * 2 : 0);
    const int32_t* kx = kk + (per_image ? (size_t)n * ow * ks : 0);
    const uint8_t* src = in + (size_t)n * h * w * 3;
    uint8_t* dst = tmp + (size_t)n * h * ow * 3;
    const int total = h * ow;
    for (int t = blockIdx.x * kBlock + threadIdx.x; t < total; t += gridDim.x * kBlock) {
        const int y = t / ow, ox = t - y * ow;
        // clamp the host-provided window so a bad table can never read out of bounds
        const int xmin = clampi(bnd[2 * ox], 0, w);
        const int cnt = min(bnd[2 * ox + 1], min(ks, w - xmin));
        const int32_t* k = kx + (size_t)ox * ks;
        const uint8_t* p = src + ((size_t)y * w + xmin) * 3;
        int s0 = 1 << (kPrec - 1), s1 = s0, s2 = s0;
        int i = 0;
        for (; i + 4 <= cnt; i += 4) {
            const Row12 q = *reinterpret_cast<const Row12*>(p + 3 * i);
            const int c0 = k[i], c1 = k[i + 1], c2 = k[i + 2], c3 = k[i + 3];
            s0 = mac24(mac24(mac24(mac24(s0, q.a & 0xff, c0), q.a >> 24, c1), (q.b >> 16) & 0xff, c2),
                       (q.c >> 8) & 0xff, c3);
            s1 = mac24(mac24(mac24(mac24(s1, (q.a >> 8) & 0xff, c0), q.b & 0xff, c1), q.b >> 24, c2),
                       (q.c >> 16) & 0xff, c3);
            s2 = mac24(mac24(mac24(mac24(s2, (q.a >> 16) & 0xff, c0), (q.b >> 8) & 0xff, c1), q.c & 0xff, c2),
                       q.c >> 24, c3);
        }
        for (; i < cnt; ++i) {
            const int c = k[i];
            s0 = mac24(s0, p[3 * i], c);
            s1 = mac24(s1, p[3 * i + 1], c);
            s2 = mac24(s2, p[3 * i + 2], c);
        }
        uint8_t* o = dst + (size_t)t * 3;
        o[0] = clip8(s0);
        o[1] = clip8(s1);
        o[2] = clip8(s2);
    }
}

// horizontal, coefficient-resident form: one thread per (output column, strip of kHStrip rows).
// An output column's window and taps are the same on every row, so they are read once into
// registers (<= kHMaxTaps of them) and reused down the strip instead of being re-fetched per
// pixel — the per-pixel version spends most of its loads on coefficients (36 B per 3 B written).
constexpr int kHStrip = 8, kHMaxTaps = 12;
__global__ __launch_bounds__(kBlock) void resample_h_strip_kernel(const uint8_t* __restrict__ in,
                                                                  uint8_t* __restrict__ tmp, int h,
                                                                  int w, int ow,
                                                                  const int32_t* __restrict__ bounds,
                                                                  const int32_t* __restrict__ kk, int ks,
                                                                  int per_image) {
    const unsigned n = blockIdx.y;
    const int32_t* bnd = bounds + (per_image ? (size_t)n * ow * 2 : 0);
    const int32_t* kx = kk + (per_image ? (size_t)n * ow * ks : 0);
    const uint8_t* src = in + (size_t)n * h * w * 3;
    uint8_t* dst = tmp + (size_t)n * h * ow * 3;
    const int strips = (h + kHStrip - 1) / kHStrip;
    const int total = strips * ow;
    for (int t = blockIdx.x * kBlock + threadIdx.x; t < total; t += gridDim.x * kBlock) {
        const int strip = t / ow, ox = t - strip * ow;
        const int xmin = clampi(bnd[2 * ox], 0, w);
        const int cnt = min(bnd[2 * ox + 1], min(ks, w - xmin));
        int c[kHMaxTaps];
#pragma unroll
        for (int i = 0; i < kHMaxTaps; ++i) c[i] = i < cnt ? kx[(size_t)ox * ks + i] : 0;
        // whole groups of four taps may read up to 3 pixels past the window: keep them in the row
        const int groups = (cnt + 3) / 4;
        const bool vec = xmin + 4 * groups <= w;
        const int y1 = min((strip + 1) * kHStrip, h);
        for (int y = strip * kHStrip; y < y1; ++y) {
            const uint8_t* p = src + ((size_t)y * w + xmin) * 3;
            int s0 = 1 << (kPrec - 1), s1 = s0, s2 = s0;
            if (vec) {
#pragma unroll
                for (int g = 0; g < kHMaxTaps / 4; ++g) {
                    if (g < groups) {
                        const Row12 q = *reinterpret_cast<const Row12*>(p + 12 * g);
                        const int c0 = c[4 * g], c1 = c[4 * g + 1], c2 = c[4 * g + 2], c3 = c[4 * g + 3];
                        s0 = mac24(mac24(mac24(mac24(s0, q.a & 0xff, c0), q.a >> 24, c1), (q.b >> 16) & 0xff, c2),
                                   (q.c >> 8) & 0xff, c3);
                        s1 = mac24(mac24(mac24(mac24(s1, (q.a >> 8) & 0xff, c0), q.b & 0xff, c1), q.b >> 24, c2),
                                   (q.c >> 16) & 0xff, c3);
                        s2 = mac24(mac24(mac24(mac24(s2, (q.a >> 16) & 0xff, c0), (q.b >> 8) & 0xff, c1),
                                         q.c & 0xff, c2), q.c >> 24, c3);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < kHMaxTaps; ++i) {
                    if (i < cnt) {
                        s0 = mac24(s0, p[3 * i], c[i]);
                        s1 = mac24(s1, p[3 * i + 1], c[i]);
                        s2 = mac24(s2, p[3 * i + 2], c[i]);
                    }
                }
            }
            uint8_t* o = dst + ((size_t)y * ow + ox) * 3;
            o[0] = clip8(s0);
            o[1] = clip8(s1);
            o[2] = clip8(s2);
        }
    }
}

// vertical: tmp [n][h][ow][3] -> out [n][oh][ow][3]; one thread per (oy, 4 consecutive bytes)
// when the row is a whole number of dwords, else one thread per byte.
template <int VEC>
__global__ __launch_bounds__(kBlock) void resample_v_kernel(const uint8_t* __restrict__ tmp,
                                                            uint8_t* __restrict__ out, int h, int oh,
                                                            int ow, const int32_t* __restrict__ bounds,
                                                            const int32_t* __restrict__ kk, int ks,
                                                            int per_image) {
    const unsigned n = blockIdx.y;
    const int32_t* bnd = bounds + (per_image ? (size_t)n * oh * 2 : 0);
    const int32_t* ky = kk + (per_image ? (size_t)n * oh * ks : 0);
    const int rowb = ow * 3, rowv = rowb / VEC;
    const uint8_t* src = tmp + (size_t)n * h * rowb;
    uint8_t* dst = out + (size_t)n * oh * rowb;
    const int total = oh * rowv;
    for (int t = blockIdx.x * kBlock + threadIdx.x; t < total; t += gridDim.x * kBlock) {
        const int oy = t / rowv, xv = t - oy * rowv;
        const int ymin = clampi(bnd[2 * oy], 0, h);
        const int cnt = min(bnd[2 * oy + 1], min(ks, h - ymin));
        const int32_t* k = ky + (size_t)oy * ks;
        const uint8_t* p = src + (size_t)ymin * rowb + xv * VEC;
        if (VEC == 4) {
            int a0 = 1 << (kPrec - 1), a1 = a0, a2 = a0, a3 = a0;
            for (int i = 0; i < cnt; ++i) {
                const unsigned v = *reinterpret_cast<const uint32_t*>(p + (size_t)i * rowb);
                const int c = k[i];
                a0 = mac24(a0, v & 0xff, c);
                a1 = mac24(a1, (v >> 8) & 0xff, c);
                a2 = mac24(a2, (v >> 16) & 0xff, c);
                a3 = mac24(a3, v >> 24, c);
            }
            *reinterpret_cast<uint32_t*>(dst + (size_t)oy * rowb + xv * 4) =
                (unsigned)clip8(a0) | (unsigned)clip8(a1) << 8 | (unsigned)clip8(a2) << 16 |
                (unsigned)clip8(a3) << 24;
        } else {
            int sacc = 1 << (kPrec - 1);
            for (int i = 0; i < cnt; ++i) sacc = mac24(sacc, p[(size_t)i * rowb], k[i]);
            dst[(size_t)oy * rowb + xv] = clip8(sacc);
        }
    }
}

// Both passes in one kernel for resamples whose windows are short (crop -> LANCZOS back to the
// original size, mild up/down scales): one workgroup = a 32x32 output tile.  The input window of
// the tile (<= 48 rows x 48 pixels) is copied into LDS with aligned dword loads, the horizontal
// pass runs LDS -> LDS (8-bit intermediate, as Pillow keeps it), the vertical pass LDS -> global.
// Against the two-kernel form this drops the HBM round trip of the intermediate image and the
// unaligned 12-byte global loads that kept the texture addresser 84 % busy.
constexpr int kRT = 32, kRWin = 48, kRTapsMax = 10;  // kRTaps: 8 (crop, up-scales) or 10 (256 -> 224)
constexpr int kRPitch = kRWin * 3 + 8;  // bytes per window row in LDS (a row starts up to 3 bytes in)

template <int kRTaps>
__global__ __launch_bounds__(kBlock) void resample_tile_kernel(const uint8_t* __restrict__ in,
                                                               uint8_t* __restrict__ out, int h, int w,
                                                               int oh, int ow,
                                                               const int32_t* __restrict__ xbounds,
                                                               const int32_t* __restrict__ xkk, int kx,
                                                               const int32_t* __restrict__ ybounds,
                                                               const int32_t* __restrict__ ykk, int ky,
                                                               int per_image, int n_images) {
    __shared__ uint32_t win[(kRWin * kRPitch + 40) / 4];
    __shared__ uint32_t tmpw[(kRWin + kRTaps) * kRT * 3 / 4];  // + rows that only zero taps reach
    __shared__ int kxs[kRT][kRTaps], kys[kRT][kRTaps];
    __shared__ int xmn[kRT], xct[kRT], ymn[kRT], yct[kRT], rsh[kRWin];
    const lf::TileId tile = lf::xcd_tile((ow + kRT - 1) / kRT, (oh + kRT - 1) / kRT, n_images);
    if (!tile.ok) return;
    const unsigned n = (unsigned)tile.n;
    const int ox0 = tile.tx * kRT, oy0 = tile.ty * kRT;
    const int cols = min(kRT, ow - ox0), rows = min(kRT, oh - oy0);
    const int tid = threadIdx.x;
    if (tid < 2 * kRT) {  // tables of this tile's columns (threads 0..31) and rows (32..63)
        const bool isx = tid < kRT;
        const int l = tid & (kRT - 1);
        const int cnt_axis = isx ? cols : rows, o = (isx ? ox0 : oy0) + l;
        const int len = isx ? w : h, ks = isx ? kx : ky, on = isx ? ow : oh;
        const int32_t* bnd = (isx ? xbounds : ybounds) + (per_image ? (size_t)n * on * 2 : 0);
        const int32_t* kk = (isx ? xkk : ykk) + (per_image ? (size_t)n * on * ks : 0);
        int mn = 0, ct = 0;
        if (l < cnt_axis) {
            mn = clampi(bnd[2 * o], 0, len);
            ct = max(0, min(min(bnd[2 * o + 1], ks), min(len - mn, kRTaps)));
        }
#pragma unroll
        for (int i = 0; i < kRTaps; ++i) {
            const int v = i < ct ? kk[(size_t)o * ks + i] : 0;
            if (isx) kxs[l][i] = v; else kys[l][i] = v;
        }
        if (isx) { xmn[l] = mn; xct[l] = ct; } else { ymn[l] = mn; yct[l] = ct; }
    }
    __syncthreads();
    // windows start at the first column / row's start (the starts grow with the output index)
    const int xlo = xmn[0], ylo = ymn[0];
    const int wy = min(kRWin, min(h, ymn[rows - 1] + yct[rows - 1]) - ylo);
    const size_t img_bytes = (size_t)h * w * 3;
    const uint8_t* img = in + (size_t)n * img_bytes;
    const unsigned mis = (unsigned)(reinterpret_cast<size_t>(img) & 3);
    // resource over this image, base aligned down: dwords that stick out read as zero
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(img - mis), 0, (int)((img_bytes + mis + 3) & ~(size_t)3), 0x00020000);
    // a window row (<= 3 + 144 + 3 bytes from its aligned start) as 19 eight-byte pieces:
    // three rows per wave
    constexpr int kPieces = kRPitch / 8;
    for (int it = tid; it < wy * kPieces; it += kBlock) {
        const int r = it / kPieces, d = it - r * kPieces;
        const unsigned off = ((unsigned)(ylo + r) * (unsigned)w + (unsigned)xlo) * 3u + mis;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, (off & ~3u) + 8u * d, 0, 0));
        win[(r * kRPitch) / 4 + 2 * d] = v.x;
        win[(r * kRPitch) / 4 + 2 * d + 1] = v.y;
        if (d == 0) rsh[r] = (int)(off & 3u);
    }
    __syncthreads();
    const uint8_t* wb = reinterpret_cast<const uint8_t*>(win);
    uint8_t* tb = reinterpret_cast<uint8_t*>(tmpw);
    {   // horizontal pass: thread = one output column, every 8th window row
        const int c = tid & (kRT - 1);
        if (c < cols) {
            int k[kRTaps];
#pragma unroll
            for (int i = 0; i < kRTaps; ++i) k[i] = kxs[c][i];
            const int rel = min(xmn[c] - xlo, kRWin - 1) * 3;
            for (int r = tid / kRT; r < wy; r += kBlock / kRT) {
                const uint8_t* q = wb + r * kRPitch + rsh[r] + rel;
                int s0 = 1 << (kPrec - 1), s1 = s0, s2 = s0;
#pragma unroll
                for (int i = 0; i < kRTaps; ++i) {  // taps past the count have k = 0
                    s0 = mac24(s0, q[3 * i], k[i]);
                    s1 = mac24(s1, q[3 * i + 1], k[i]);
                    s2 = mac24(s2, q[3 * i + 2], k[i]);
                }
                uint8_t* o = tb + (r * kRT + c) * 3;
                o[0] = clip8(s0);
                o[1] = clip8(s1);
                o[2] = clip8(s2);
            }
        }
    }
    __syncthreads();
    // vertical pass: thread = four consecutive bytes of one output row of the tile
    uint8_t* dst = out + (((size_t)n * oh + oy0) * ow + ox0) * 3;
    const int rowd = cols * 3 / 4;  // cols % 4 == 0 (ow % 4 == 0)
    for (int it = tid; it < rows * (kRT * 3 / 4); it += kBlock) {
        const int oyl = it / (kRT * 3 / 4), dc = it - oyl * (kRT * 3 / 4);
        if (dc >= rowd) continue;
        const int rely = min(ymn[oyl] - ylo, kRWin);  // rows past the window meet zero taps only
        const uint32_t* col = tmpw + __mul24(rely, kRT * 3 / 4) + dc;
        int a0 = 1 << (kPrec - 1), a1 = a0, a2 = a0, a3 = a0;
#pragma unroll
        for (int i = 0; i < kRTaps; ++i) {
            const int kv = kys[oyl][i];
            const unsigned v = col[i * (kRT * 3 / 4)];
            a0 = mac24(a0, v & 0xff, kv);
            a1 = mac24(a1, (v >> 8) & 0xff, kv);
            a2 = mac24(a2, (v >> 16) & 0xff, kv);
            a3 = mac24(a3, v >> 24, kv);
        }
        *reinterpret_cast<uint32_t*>(dst + (size_t)oyl * ow * 3 + 4 * dc) =
            (unsigned)clip8(a0) | (unsigned)clip8(a1) << 8 | (unsigned)clip8(a2) << 16 | (unsigned)clip8(a3) << 24;
    }
}

}  // namespace

extern "C" {

int lf_warp_bicubic_u8(const uint8_t* in, uint8_t* out, const double* coeffs, int perspective,
                       int n, int h, int w, lf_stream_t stream) {
    LF_REQUIRE(in && out && coeffs, "lf_warp_bicubic: null buffer");
    LF_REQUIRE(n > 0 && h > 0 && w > 0, "lf_warp_bicubic: bad dims n=%d h=%d w=%d", n, h, w);
    LF_REQUIRE((size_t)h * w < (1u << 30), "lf_warp_bicubic: image too large");
    LF_REQUIRE(in != out, "lf_warp_bicubic: in-place warp is not supported");
    const int persp = perspective & 1;
    if (perspective & 2) {  // caller's hint: the maps are axis-aligned scales (verified per image)
        const unsigned grid = lf::xcd_grid((size_t)((w + kTX - 1) / kTX) * ((h + kTY - 1) / kTY) * n);
        warp_bicubic_tile_kernel<<<grid, kBlock, 0, lf::as_stream(stream)>>>(in, out, coeffs, persp, h, w, n);
    } else {
        dim3 grid(lf::stream_grid((size_t)h * w, kBlock, 1024), n);
        warp_bicubic_kernel<<<grid, kBlock, 0, lf::as_stream(stream)>>>(in, out, coeffs, persp, h, w);
    }
    return lf::check_launch("lf_warp_bicubic");
}

int lf_affine_nearest_fixed_u8(const uint8_t* in, uint8_t* out, const int32_t* fix6,
                               const int32_t* ohw, const int64_t* out_off, int n, int h, int w,
                               int max_out_pixels, int fill, lf_stream_t stream) {
    LF_REQUIRE(in && out && fix6 && ohw && out_off, "lf_affine_nearest_fixed: null buffer");
    LF_REQUIRE(n > 0 && h > 0 && w > 0 && max_out_pixels > 0,
               "lf_affine_nearest_fixed: bad dims n=%d h=%d w=%d max_out_pixels=%d", n, h, w,
               max_out_pixels);
    LF_REQUIRE(h < 32768 && w < 32768, "lf_affine_nearest_fixed: 16.16 fixed point needs sizes < 32768");
    LF_REQUIRE(fill >= 0 && fill <= 255, "lf_affine_nearest_fixed: fill must be 0..255");
    const size_t nbytes = (size_t)h * w * 3;
    if (nbytes <= (size_t)152 * 1024 && nbytes % 16 == 0 && (reinterpret_cast<size_t>(in) & 15) == 0) {
        // the source image fits a CU's LDS: one workgroup per image, source staged once (see the kernel)
        static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(affine_nearest_lds_kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) == hipSuccess;
        if (ok) {
            const unsigned wgs = (unsigned)std::min<size_t>((size_t)n, (size_t)256 * 4);
            affine_nearest_lds_kernel<<<wgs, kRotT, nbytes, lf::as_stream(stream)>>>(in, out, fix6, ohw, out_off, n, h, w,
                                                                                   (unsigned)fill);
            return lf::check_launch("lf_affine_nearest_fixed");
        }
    }
    dim3 grid(lf::stream_grid(((size_t)max_out_pixels + 3) / 4, kBlock, 1024), n);
    affine_nearest_kernel<<<grid, kBlock, 0, lf::as_stream(stream)>>>(in, out, fix6, ohw, out_off, h,
                                                                      w, (unsigned)fill);
    return lf::check_launch("lf_affine_nearest_fixed");
}

int lf_resample_tile_u8(const uint8_t* in, uint8_t* out, int n, int h, int w, int oh, int ow,
                        const int32_t* xbounds, const int32_t* xk, int kx, const int32_t* ybounds,
                        const int32_t* yk, int ky, int per_image_coeffs, lf_stream_t stream) {
    LF_REQUIRE(in && out && xbounds && xk && ybounds && yk, "lf_resample_tile: null buffer");
    LF_REQUIRE(n > 0 && h > 0 && w > 0 && oh > 0 && ow > 0, "lf_resample_tile: bad dims n=%d h=%d w=%d oh=%d ow=%d",
               n, h, w, oh, ow);
    LF_REQUIRE(kx > 0 && kx <= kRTapsMax && ky > 0 && ky <= kRTapsMax, "lf_resample_tile: kx=%d ky=%d (1..%d)",
               kx, ky, kRTapsMax);
    LF_REQUIRE(ow % 4 == 0 && (reinterpret_cast<size_t>(out) & 3) == 0,
               "lf_resample_tile: ow must be a multiple of 4 and out 4-byte aligned");
    LF_REQUIRE((size_t)h * w * 3 + 3 < ((size_t)1 << 31), "lf_resample_tile: image too large");
    LF_REQUIRE(n <= 65535, "lf_resample_tile: batch too large for grid.z");
    LF_REQUIRE(in != out, "lf_resample_tile: in-place resample is not supported");
    const unsigned grid = lf::xcd_grid((size_t)((ow + kRT - 1) / kRT) * ((oh + kRT - 1) / kRT) * n);
    if (kx <= 8 && ky <= 8)
        resample_tile_kernel<8><<<grid, kBlock, 0, lf::as_stream(stream)>>>(
            in, out, h, w, oh, ow, xbounds, xk, kx, ybounds, yk, ky, per_image_coeffs, n);
    else
        resample_tile_kernel<10><<<grid, kBlock, 0, lf::as_stream(stream)>>>(
            in, out, h, w, oh, ow, xbounds, xk, kx, ybounds, yk, ky, per_image_coeffs, n);
    return lf::check_launch("lf_resample_tile");
}

int lf_resample_u8(const uint8_t* in, uint8_t* tmp, uint8_t* out, int n, int h, int w, int oh,
                   int ow, const int32_t* xbounds, const int32_t* xk, int kx,
                   const int32_t* ybounds, const int32_t* yk, int ky, int per_image_coeffs,
                   lf_stream_t stream) {
    LF_REQUIRE(in && tmp && out && xbounds && xk && ybounds && yk, "lf_resample: null buffer");
    LF_REQUIRE(n > 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && kx > 0 && ky > 0,
               "lf_resample: bad dims n=%d h=%d w=%d oh=%d ow=%d kx=%d ky=%d", n, h, w, oh, ow, kx,
               ky);
    hipStream_t s = lf::as_stream(stream);
    if (kx <= kHMaxTaps)
        resample_h_strip_kernel<<<dim3(lf::stream_grid((size_t)((h + kHStrip - 1) / kHStrip) * ow, kBlock, 1024), n),
                                  kBlock, 0, s>>>(in, tmp, h, w, ow, xbounds, xk, kx, per_image_coeffs);
    else
        resample_h_kernel<<<dim3(lf::stream_grid((size_t)h * ow, kBlock, 1024), n), kBlock, 0, s>>>(
            in, tmp, h, w, ow, xbounds, xk, kx, per_image_coeffs);
    const bool vec = (ow * 3) % 4 == 0 && ((reinterpret_cast<size_t>(tmp) | reinterpret_cast<size_t>(out)) & 3) == 0;
    if (vec)
        resample_v_kernel<4><<<dim3(lf::stream_grid((size_t)oh * ow * 3 / 4, kBlock, 1024), n), kBlock,
                               0, s>>>(tmp, out, h, oh, ow, ybounds, yk, ky, per_image_coeffs);
    else
        resample_v_kernel<1><<<dim3(lf::stream_grid((size_t)oh * ow * 3, kBlock, 1024), n), kBlock, 0,
                               s>>>(tmp, out, h, oh, ow, ybounds, yk, ky, per_image_coeffs);
    return lf::check_launch("lf_resample");
}

}  // extern "C"
