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
typedef float f32x4 __attribute__((ext_vector_type(4)));

#if defined(__HIPCC__)
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
