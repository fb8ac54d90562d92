// Development microbenchmark (not part of libleafhip): what a byte-moving kernel can reach on
// this box, by access width / cache policy / grid shape, on a 4096 x 224x224x3 uint8 batch.
// Build: hipcc -O3 --offload-arch=gfx950 copy_bw.hip -o copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <typename T, bool NTL, bool NTS, int UNROLL>
__global__ __launch_bounds__(256) void copy_kernel(const T* __restrict__ in, T* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        T v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NTL ? __builtin_nontemporal_load(&in[i + u * stride]) : in[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (NTS) __builtin_nontemporal_store(v[u], &out[i + u * stride]);
            else out[i + u * stride] = v[u];
        }
    }
    for (; i < n; i += stride) out[i] = in[i];
}

// the shape of flip/pack: grid (splits, images), 12 bytes per lane
template <bool NT>
__global__ __launch_bounds__(256) void copy3_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, unsigned per_img) {
    const uint32_t* s = in + (size_t)blockIdx.y * per_img * 3;
    uint32_t* d = out + (size_t)blockIdx.y * per_img * 3;
    for (unsigned t = blockIdx.x * 256 + threadIdx.x; t < per_img; t += gridDim.x * 256) {
        unsigned a, b, c;
        if (NT) { a = __builtin_nontemporal_load(s + 3 * t); b = __builtin_nontemporal_load(s + 3 * t + 1); c = __builtin_nontemporal_load(s + 3 * t + 2); }
        else { a = s[3 * t]; b = s[3 * t + 1]; c = s[3 * t + 2]; }
        if (NT) { __builtin_nontemporal_store(a, d + 3 * t); __builtin_nontemporal_store(b, d + 3 * t + 1); __builtin_nontemporal_store(c, d + 3 * t + 2); }
        else { d[3 * t] = a; d[3 * t + 1] = b; d[3 * t + 2] = c; }
    }
}

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <typename F>
static double time_it(F launch, int iters) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main() {
    const size_t nimg = 4096, per = 224 * 224 * 3, bytes = nimg * per;
    uint8_t *in, *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    const size_t n16 = bytes / 16;
    auto report = [&](const char* name, double ms) { printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, 2.0 * bytes / ms * 1e-6); fflush(stdout); };
    const int grids[] = {1024, 2048, 4096, 8192, 16384, (int)((n16 + 255) / 256)};
    for (int g : grids) {
        char nm[96];
        snprintf(nm, 96, "x4 plain u1 grid %d", g);
        report(nm, time_it([&] { copy_kernel<u4, false, false, 1><<<g, 256>>>((const u4*)in, (u4*)out, n16); }, 20));
        snprintf(nm, 96, "x4 plain u4 grid %d", g);
        report(nm, time_it([&] { copy_kernel<u4, false, false, 4><<<g, 256>>>((const u4*)in, (u4*)out, n16); }, 20));
        snprintf(nm, 96, "x4 nt-store u4 grid %d", g);
        report(nm, time_it([&] { copy_kernel<u4, false, true, 4><<<g, 256>>>((const u4*)in, (u4*)out, n16); }, 20));
        snprintf(nm, 96, "x4 nt-both u4 grid %d", g);
        report(nm, time_it([&] { copy_kernel<u4, true, true, 4><<<g, 256>>>((const u4*)in, (u4*)out, n16); }, 20));
        snprintf(nm, 96, "x4 nt-both u1 grid %d", g);
        report(nm, time_it([&] { copy_kernel<u4, true, true, 1><<<g, 256>>>((const u4*)in, (u4*)out, n16); }, 20));
    }
    const unsigned per_img = 224 * 224 / 4;
    for (int sp : {1, 2, 4, 7, 13, 49}) {
        char nm[96];
        snprintf(nm, 96, "x3 per-image plain splits %d", sp);
        report(nm, time_it([&] { copy3_kernel<false><<<dim3(sp, nimg), 256>>>((const uint32_t*)in, (uint32_t*)out, per_img); }, 20));
        snprintf(nm, 96, "x3 per-image nt splits %d", sp);
        report(nm, time_it([&] { copy3_kernel<true><<<dim3(sp, nimg), 256>>>((const uint32_t*)in, (uint32_t*)out, per_img); }, 20));
    }
    // ping-pong a->b, b->a at smaller batches: does nt give up Infinity Cache residency?
    for (size_t ni : {64, 128, 256, 512, 1024, 2048, 4096}) {
        const size_t nb = ni * per, m16 = nb / 16;
        const int g = (int)((m16 + 255) / 256);
        char nm[96];
        double ms;
        ms = time_it([&] { copy_kernel<u4, false, false, 1><<<g, 256>>>((const u4*)in, (u4*)out, m16);
                           copy_kernel<u4, false, false, 1><<<g, 256>>>((const u4*)out, (u4*)in, m16); }, 20);
        snprintf(nm, 96, "pingpong plain  %zu images", ni);
        printf("%-44s %8.4f ms  %7.1f GB/s\n", nm, ms, 4.0 * nb / ms * 1e-6);
        ms = time_it([&] { copy_kernel<u4, true, true, 1><<<g, 256>>>((const u4*)in, (u4*)out, m16);
                           copy_kernel<u4, true, true, 1><<<g, 256>>>((const u4*)out, (u4*)in, m16); }, 20);
        snprintf(nm, 96, "pingpong nt     %zu images", ni);
        printf("%-44s %8.4f ms  %7.1f GB/s\n", nm, ms, 4.0 * nb / ms * 1e-6);
        ms = time_it([&] { copy_kernel<u4, true, false, 1><<<g, 256>>>((const u4*)in, (u4*)out, m16);
                           copy_kernel<u4, true, false, 1><<<g, 256>>>((const u4*)out, (u4*)in, m16); }, 20);
        snprintf(nm, 96, "pingpong nt-load %zu images", ni);
        printf("%-44s %8.4f ms  %7.1f GB/s\n", nm, ms, 4.0 * nb / ms * 1e-6);
        ms = time_it([&] { copy_kernel<u4, false, true, 1><<<g, 256>>>((const u4*)in, (u4*)out, m16);
                           copy_kernel<u4, false, true, 1><<<g, 256>>>((const u4*)out, (u4*)in, m16); }, 20);
        snprintf(nm, 96, "pingpong nt-store %zu images", ni);
        printf("%-44s %8.4f ms  %7.1f GB/s\n", nm, ms, 4.0 * nb / ms * 1e-6);
    }
    report("hipMemcpyDtoD", time_it([&] { CK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0)); }, 20));
    return 0;
}
