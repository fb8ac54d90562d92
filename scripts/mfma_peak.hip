// Sustained fp32 MFMA rate calibration: register-operand v_mfma_f32_32x32x2_f32 loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = seed + threadIdx.x * 1e-3f, b = seed - threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a += 1e-7f;
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 256 * 8192 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs_per_cu = 1; wgs_per_cu <= 3; ++wgs_per_cu) {
        const int grid = 256 * wgs_per_cu, iters = 20000;
        k<4><<<grid, 256>>>(out, 100, 0.5f); hipDeviceSynchronize();
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            k<4><<<grid, 256>>>(out, iters, 0.5f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flop = (double)grid * 4 /*waves*/ * iters * 4 /*NACC*/ * 4096.0;
            printf("wg/cu %d rep %d: %.2f ms  %.1f TFLOP/s\n", wgs_per_cu, rep, ms, flop / ms / 1e9);
        }
    }
    return 0;
}
