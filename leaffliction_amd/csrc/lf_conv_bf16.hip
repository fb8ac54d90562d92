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
// 32x8 output pixels x 32*NCO output channels; per 16-channel chunk the input patch sits in LDS as
// eight planes of channel PAIRS (one dword = bf16(ci), bf16(ci+1) of one pixel), so the B operand of
// a lane (one pixel, eight consecutive channels) is four conflict-free ds_read_b32 and staging is
// one ds_write_b128 per four pixels of a pair plane; weights are pre-packed once per model to
// [chunk][tap][cout][16] bf16 so the A operand (one output channel, eight channels) is one 16-byte
// LDS read.
#include "lf_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kTW = 32, kTH = 8;  // output tile
constexpr int kPW = 40;           // patch row pitch in dwords: columns x0-4 .. x0+35

struct Bf16ConvArgs {
    const float* x;
    const uint16_t* wprep;
    float* y;
    int n, cin, h, w, cout, chunks;
    const float* in_scale;
    const float* in_shift;
    int in_relu;
};

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v;
    v.x = (__bf16)lo;  // round to nearest even
    v.y = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}

template <int TAPS, int NCO>
__global__ __launch_bounds__(kThreads) void conv_bf16_kernel(Bf16ConvArgs p) {
    constexpr int R = TAPS == 9 ? 1 : 0, PH = kTH + 2 * R, KS = TAPS == 9 ? 3 : 1;
    __shared__ uint32_t patch[8][PH][kPW];
    __shared__ lf::u32x4 wl[TAPS * NCO * 32 * 2];
    const int tiles_x = (p.w + kTW - 1) / kTW;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int x0 = tx * kTW, y0 = ty * kTH;
    const int co0 = blockIdx.y * (NCO * 32);
    const int n = blockIdx.z;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, px = lane & 31, half = lane >> 5;
    const size_t hw = (size_t)p.h * p.w;
    const float* xn = p.x + (size_t)n * p.cin * hw;

    f32x16 acc[2][NCO];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NCO; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    for (int c = 0; c < p.chunks; ++c) {
        __syncthreads();  // the previous chunk's reads are done
        // input patch: item = (pair plane, patch row, group of four columns)
        for (int it = tid; it < 8 * PH * (kPW / 4); it += kThreads) {
            const int pl = it / (PH * (kPW / 4)), rem = it - pl * (PH * (kPW / 4));
            const int row = rem / (kPW / 4), q = rem - row * (kPW / 4);
            const int gy = y0 - R + row, gx = x0 - 4 + 4 * q;
            const int ci0 = c * 16 + 2 * pl;
            f32x4v a = {0.0f, 0.0f, 0.0f, 0.0f}, b = {0.0f, 0.0f, 0.0f, 0.0f};
            // w % 4 == 0: a group of four columns is inside the image or outside it as a whole
            if (gy >= 0 && gy < p.h && gx >= 0 && gx < p.w) {
                const size_t off = (size_t)gy * p.w + gx;
                if (ci0 < p.cin) {
                    a = *reinterpret_cast<const f32x4v*>(xn + (size_t)ci0 * hw + off);
                    if (p.in_scale) {
                        const float s = p.in_scale[ci0], t = p.in_shift[ci0];
#pragma unroll
                        for (int e = 0; e < 4; ++e) a[e] = fmaf(a[e], s, t);
                    }
                    if (p.in_relu)
#pragma unroll
                        for (int e = 0; e < 4; ++e) a[e] = fmaxf(a[e], 0.0f);
                }
                if (ci0 + 1 < p.cin) {
                    b = *reinterpret_cast<const f32x4v*>(xn + (size_t)(ci0 + 1) * hw + off);
                    if (p.in_scale) {
                        const float s = p.in_scale[ci0 + 1], t = p.in_shift[ci0 + 1];
#pragma unroll
                        for (int e = 0; e < 4; ++e) b[e] = fmaf(b[e], s, t);
                    }
                    if (p.in_relu)
#pragma unroll
                        for (int e = 0; e < 4; ++e) b[e] = fmaxf(b[e], 0.0f);
                }
            }
            lf::u32x4 o;
            o.x = pack_bf16(a[0], b[0]);
            o.y = pack_bf16(a[1], b[1]);
            o.z = pack_bf16(a[2], b[2]);
            o.w = pack_bf16(a[3], b[3]);
            *reinterpret_cast<lf::u32x4*>(&patch[pl][row][4 * q]) = o;
        }
        // weights of this chunk for the workgroup's output channels: 16-byte pieces
        for (int it = tid; it < TAPS * NCO * 64; it += kThreads) {
            const int tap = it / (NCO * 64), rem = it - tap * (NCO * 64);
            const int col = rem >> 1, hf = rem & 1;
            const size_t src = (((size_t)c * TAPS + tap) * p.cout + co0 + col) * 16 + 8 * hf;
            wl[it] = *reinterpret_cast<const lf::u32x4*>(p.wprep + src);
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int dy = tap / KS, dx = tap - dy * KS;
            bf16x8 A[NCO];
#pragma unroll
            for (int cb = 0; cb < NCO; ++cb)
                A[cb] = __builtin_bit_cast(bf16x8, wl[((tap * NCO + cb) * 32 + px) * 2 + half]);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int row = 2 * wv + nb + dy;
                const int col = 4 + px + dx - R;
                lf::u32x4 bv;
                bv.x = patch[4 * half + 0][row][col];
                bv.y = patch[4 * half + 1][row][col];
                bv.z = patch[4 * half + 2][row][col];
                bv.w = patch[4 * half + 3][row][col];
                const bf16x8 B = __builtin_bit_cast(bf16x8, bv);
#pragma unroll
                for (int cb = 0; cb < NCO; ++cb)
                    acc[nb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cb], B, acc[nb][cb], 0, 0, 0);
            }
        }
    }
    // D[row = output channel][col = pixel]: register r of a lane is channel 8*(r/4) + 4*half + r%4
    float* yn = p.y + (size_t)n * p.cout * hw;
    const int gx = x0 + px;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int gy = y0 + 2 * wv + nb;
        if (gy >= p.h || gx >= p.w) continue;
#pragma unroll
        for (int cb = 0; cb < NCO; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + cb * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
                yn[(size_t)co * hw + (size_t)gy * p.w + gx] = acc[nb][cb][r];
            }
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

int lf_conv2d_bf16_f32(const float* x, const uint16_t* wprep, float* y, int n, int cin, int h, int w,
                       int cout, int ksize, const float* in_scale, const float* in_shift, int in_relu,
                       lf_stream_t stream) {
    LF_REQUIRE(x && wprep && y, "lf_conv2d_bf16: null buffer");
    LF_REQUIRE(n > 0 && cin > 0 && h > 0 && w > 0 && cout > 0, "lf_conv2d_bf16: bad dims");
    LF_REQUIRE(ksize == 1 || ksize == 3, "lf_conv2d_bf16: ksize must be 1 or 3");
    LF_REQUIRE(w % 4 == 0, "lf_conv2d_bf16: width must be a multiple of 4 (got %d)", w);
    LF_REQUIRE(cout % 32 == 0, "lf_conv2d_bf16: cout must be a multiple of 32 (got %d)", cout);
    LF_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "lf_conv2d_bf16: scale/shift must both be set");
    LF_REQUIRE(n <= 65535, "lf_conv2d_bf16: batch too large for grid.z");
    LF_REQUIRE(((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(wprep)) & 15) == 0,
               "lf_conv2d_bf16: x and wprep must be 16-byte aligned");
    Bf16ConvArgs a{x, wprep, y, n, cin, h, w, cout, (cin + 15) / 16, in_scale, in_shift, in_relu};
    const int tiles = ((w + kTW - 1) / kTW) * ((h + kTH - 1) / kTH);
    // two 32-channel blocks per workgroup share one staged patch; four measured the same
    const int nco = cout % 64 == 0 ? 2 : 1;
    dim3 grid(tiles, cout / (32 * nco), n);
    hipStream_t s = lf::as_stream(stream);
    if (ksize == 3) {
        if (nco == 2)
            conv_bf16_kernel<9, 2><<<grid, kThreads, 0, s>>>(a);
        else
            conv_bf16_kernel<9, 1><<<grid, kThreads, 0, s>>>(a);
    } else {
        if (nco == 2)
            conv_bf16_kernel<1, 2><<<grid, kThreads, 0, s>>>(a);
        else
            conv_bf16_kernel<1, 1><<<grid, kThreads, 0, s>>>(a);
    }
    return lf::check_launch("lf_conv2d_bf16");
}

}  // extern "C"
