// Probe: where does global_load_lds_dwordx4 put each lane's 16 bytes?  (gfx950)
// Expectation used by the kernels: LDS address = (wave-uniform pointer argument) + lane * 16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const u32x4* in, u32x4* out, int n) {
    __shared__ __attribute__((aligned(16))) u32x4 lds[512];
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < 512; i += blockDim.x) lds[i] = u32x4{0xdeadu, 0, 0, 0};
    __syncthreads();
    const int src = (tid * 37 + 5) % n;  // scattered per-lane source
    // two DMA loads per thread into two 1 KiB pieces per wave
    __builtin_amdgcn_global_load_lds(in + src, (__attribute__((address_space(3))) void*)(lds + wave * 64), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(in + (src + 1) % n, (__attribute__((address_space(3))) void*)(lds + 256 + wave * 64), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0070 | 0x0F00);  // vmcnt(0), expcnt/lgkmcnt untouched
    __syncthreads();
    out[tid] = lds[tid];
    out[256 + tid] = lds[256 + tid];
}
int main() {
    const int n = 1000;
    std::vector<unsigned> h(4 * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < 4; ++j) h[4 * i + j] = i * 10 + j;
    u32x4 *in, *out;
    hipMalloc(&in, 16 * n); hipMalloc(&out, 16 * 512);
    hipMemcpy(in, h.data(), 16 * n, hipMemcpyHostToDevice);
    k<<<1, 256>>>(in, out, n);
    std::vector<unsigned> o(4 * 512);
    hipMemcpy(o.data(), out, 16 * 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) {
        const int src = (t * 37 + 5) % n;
        for (int j = 0; j < 4; ++j) {
            if (o[4 * t + j] != (unsigned)(src * 10 + j)) ++bad;
            if (o[4 * (256 + t) + j] != (unsigned)(((src + 1) % n) * 10 + j)) ++bad;
        }
    }
    printf("lds dma probe: %d mismatches (lane t of a wave -> base + 16*t)\n", bad);
    if (bad) for (int t = 0; t < 8; ++t) printf("t=%d got %u expect %u\n", t, o[4 * t], ((t * 37 + 5) % n) * 10);
    return bad != 0;
}
