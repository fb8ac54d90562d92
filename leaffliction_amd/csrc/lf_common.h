// Internal helpers shared by the libleafhip translation units (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "leafhip.h"

namespace lf {

void set_error(const char* fmt, ...);
void clear_error();

inline hipStream_t as_stream(lf_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// After a kernel launch: translate a HIP launch error into the ABI's error code.
inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return LF_ERR_LAUNCH;
    }
    return LF_OK;
}

#define LF_REQUIRE(cond, ...)        \
    do {                             \
        if (!(cond)) {               \
            lf::set_error(__VA_ARGS__); \
            return LF_ERR_INVALID;   \
        }                            \
    } while (0)

// Grid for a grid-stride streaming kernel: enough workgroups to fill 256 CUs a few
// times over, capped (guide §6 G11).
inline unsigned stream_grid(size_t work_items, unsigned block, unsigned cap = 256u * 8u) {
    size_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

// Byte-moving kernels run one short pass per thread on the largest grid that fits
// (measured on MI355X, 4096 x 224x224x3: a 2048-workgroup grid-stride copy moves
// 5.1 TB/s, one 16-byte item per thread 6.1 TB/s, the same with nontemporal accesses
// 6.6 TB/s; scripts/microbench/copy_bw.hip).
constexpr unsigned kFullGrid = 1u << 30;

// Nontemporal accesses pay once the launch is larger than what the Infinity Cache would
// have kept for the next kernel anyway (256 MiB); below that plain accesses win
// (ping-pong copy of 512 images: 7.0 TB/s plain, 6.3 TB/s nontemporal).
inline bool streaming(size_t bytes_touched) { return bytes_touched >= ((size_t)256 << 20); }

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Q8.8 Gaussian taps as a kernel argument (they live in scalar registers), and the i8-MFMA blur
// (lf_blur_mfma.hip) that lf_gauss_blur_u8 tries first.
struct BlurTaps {
    uint16_t k[32];
};
bool blur_mfma_launch(const uint8_t* in, uint8_t* out, int n, int h, int w, int channels,
                      const BlurTaps& taps, int ksize, hipStream_t s);
typedef float f32x4 __attribute__((ext_vector_type(4)));

// lf_conv2d_bf16_train is served by two kernels: the streaming one (lf_conv_bf16s.hip: Cin, Cout <= 64,
// filter bank resident in LDS, one statistics partial per workgroup) and the K-chunked one
// (lf_conv_bf16.hip: everything else, one partial per tile).
struct ConvBf16TrainArgs {
    const void* x;            // fp32 or bf16 NCHW
    const uint16_t* wprep;    // bf16 [ceil(cin/16)][taps][cout][16]
    uint16_t* y;              // bf16 NCHW
    int n, cin, h, w, cout;
    const float* in_scale;    // optional prologue relu?(x*scale+shift)
    const float* in_shift;
    int in_relu;
    int accumulate;
    float* stat_part;         // [cout][stat_tiles][2] or null
    const float* stat_pivot;
    long long stat_tiles;
    const uint16_t* stat_mask_y;
    const float* mask_scale;
    const float* mask_shift;
    int mask_relu;
    int tiles_x, tiles_y;     // filled by the streaming launcher
    int seg_tiles, segs;      // a column strip is walked in `segs` segments of `seg_tiles` tiles (see conv_bf16s_kernel)
    int interleave;           // 1: an image's strips on one XCD (see conv_bf16s_kernel)
    // inference: v*out_scale[co]+out_shift[co] (+ReLU) on the fp32 accumulators before rounding — the
    // layer's folded BatchNorm(+ReLU), so that what is stored is the activation itself
    const float* out_scale;
    const float* out_shift;
    int out_relu;
    // inference, optional: per (image, segment) channel sums of the STORED values (what a global-average pool of the
    // stored activation adds up): unit_sums[(n * segments_per_image + segment) * cout + co]
    float* unit_sums;
};
// arguments of the bf16 weight-gradient kernel (lf_wgrad_bf16.hip)
struct WgradBf16Args {
    const uint16_t* x;   // [N][Cin][H][W] bf16 (STEM: const float*, fp32 [N][Cin][H][W])
    const uint16_t* g;   // [N][Cout][H][W] bf16: dY itself, or the upstream gradient when bn_y is set
    float* part;         // [splits][Cin][TAPS][Cout]
    const float* in_scale;  // optional prologue on A: relu?(x*scale[ci]+shift[ci])
    const float* in_shift;
    int in_relu;
    int n, cin, cout, h, w;
    int tiles_x, tiles_y;
    int seg_tiles, segs;   // a column strip is walked in `segs` segments of `seg_tiles` tiles (wgrad_bf16_kernel)
    int interleave;        // 1: an image's strips on one XCD
    // optional: dY = BatchNorm backward of g (BN input bn_y), formed while staging:
    //   dz = (g*alpha[n][co] + add[n][co]) * [bn_y*coef0[co] + coef1[co] > 0 or !bn_relu]
    //   dY = bf16(coef2[co]*dz + coef3[co]*bn_y + coef4[co])
    // and written to dy_out (may be null) by the ci-block-0 workgroups, each element once
    const uint16_t* bn_y;
    const float* bn_alpha;
    const float* bn_add;
    const float* bn_coef;  // [5][Cout]
    uint16_t* dy_out;
    int bn_relu;
};

long long conv_bf16s_parts(int n, int cin, int h, int w, int cout, int ksize, int x_bf16);  // 0 = shape not covered
int conv_bf16s_units_per_image(int n, int cin, int h, int w, int cout, int ksize, int x_bf16);   // segments per image
int conv_bf16s_launch(ConvBf16TrainArgs a, int ksize, int x_bf16, hipStream_t s);

// Tile kernels whose neighbouring tiles share input halos: workgroups are dealt to the eight XCDs
// round-robin in dispatch order and every XCD has its own L2, so a plain (tile_x, tile_y, image)
// grid puts horizontally adjacent tiles under different L2s and every halo line is fetched from
// HBM once per XCD that touches it.  These kernels launch a 1-D grid of 8*ceil(total/8) workgroups
// and give XCD k the k-th contiguous eighth of the tiles (x fastest, then y, then image).
inline unsigned xcd_grid(size_t total_tiles) { return (unsigned)(8 * ((total_tiles + 7) / 8)); }

#if defined(__HIPCC__)
struct TileId {
    int tx, ty, n;
    bool ok;
};
__device__ __forceinline__ TileId xcd_tile(int tiles_x, int tiles_y, int n_images) {
    const unsigned per_image = (unsigned)(tiles_x * tiles_y), total = per_image * (unsigned)n_images;
    const unsigned per_xcd = (total + 7) / 8, b = blockIdx.x;
    const unsigned id = (b & 7u) * per_xcd + (b >> 3);
    TileId t;
    t.ok = id < total;
    t.n = (int)(id / per_image);
    const unsigned rem = id - (unsigned)t.n * per_image;
    t.ty = (int)(rem / (unsigned)tiles_x);
    t.tx = (int)(rem - (unsigned)t.ty * (unsigned)tiles_x);
    return t;
}

// The same for kernels on a (blocks, images) grid whose neighbouring blocks read neighbouring
// source lines (gathers along rotated / sheared rows): the block a workgroup should act as.
struct Block2 {
    unsigned x, y;
};
__device__ __forceinline__ Block2 xcd_block2() {
    const unsigned gx = gridDim.x, total = gx * gridDim.y;
    const unsigned b = blockIdx.x + gx * blockIdx.y;
    const unsigned k = b & 7u, fl = total >> 3, rm = total & 7u;
    const unsigned id = k * fl + (k < rm ? k : rm) + (b >> 3);
    Block2 r;
    r.y = id / gx;
    r.x = id - r.y * gx;
    return r;
}

template <bool NT, typename T>
__device__ __forceinline__ T ldg(const T* p) {
    return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT, typename T>
__device__ __forceinline__ void stg(T* p, T v) {
    if (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}
#endif

}  // namespace lf
