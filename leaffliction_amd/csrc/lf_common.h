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

}  // namespace lf
