// What a LOW-OCCUPANCY streaming reader reaches (2 workgroups of 4 waves per CU: the residency of the bf16
// convolution kernels), by how its loads are kept in flight:
//   reg   : global_load_dwordx4 into registers, D loads per lane in flight (D * 4 VGPRs of staging)
//   ring  : global_load_lds_dwordx4 (LDS-DMA) into a per-wave ring of R tiles x T KiB, consumed with ds_read_b128
//           written in inline asm behind a counted s_waitcnt vmcnt((R-1)*T): no staging VGPRs at all
//   ringc : the same ring consumed with ordinary (compiler-visible) LDS reads: hipcc puts vmcnt(0) in front of
//           every LDS read that may alias an in-flight DMA, which empties the ring at every tile
// hipcc -O3 --offload-arch=gfx950 lds_ring_bw.hip -o lds_ring_bw && ./lds_ring_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int kT = 256, kWaves = 4;

__device__ __forceinline__ u32x4 lds_read_asm(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

// every wave streams its own contiguous run of `per_wave` 1-KiB pieces
template <int D>
__global__ __launch_bounds__(kT) void reg_kernel(const u32x4* __restrict__ in, unsigned* __restrict__ out,
                                                 size_t per_wave) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t w = (size_t)blockIdx.x * kWaves + wave;
    const u32x4* src = in + w * per_wave * 64 + lane;
    u32x4 r[D];
    unsigned acc = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) r[d] = src[(size_t)d * 64];
    for (size_t i = 0; i + D <= per_wave; i += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const u32x4 v = r[d];
            const size_t nxt = i + D + d < per_wave ? i + D + d : per_wave - 1;   // clamped: loads stay unconditional
            r[d] = src[nxt * 64];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (lane == 0) out[w] = 0;
    atomicXor(&out[w], acc);
}

template <int R, int T, bool ASM>
__global__ __launch_bounds__(kT) void ring_kernel(const u32x4* __restrict__ in, unsigned* __restrict__ out,
                                                  size_t per_wave) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t w = (size_t)blockIdx.x * kWaves + wave;
    const u32x4* src = in + w * per_wave * 64 + lane;
    u32x4* ring = reinterpret_cast<u32x4*>(lds) + (size_t)wave * R * T * 64;   // [R][T][64 lanes]
    const size_t tiles = per_wave / T;
    unsigned acc = 0;
    auto issue = [&](size_t t) {
        const size_t tt = t < tiles ? t : tiles - 1;   // clamped: the DMAs stay unconditional
        u32x4* slot = ring + (t % R) * T * 64;
#pragma unroll
        for (int k = 0; k < T; ++k)
            __builtin_amdgcn_global_load_lds(src + (tt * T + k) * 64, LDSP(slot + k * 64), 16, 0, 0);
    };
#pragma unroll
    for (int t = 0; t < R - 1; ++t) issue(t);
    for (size_t t = 0; t < tiles; ++t) {
        issue(t + R - 1);   // into the slot whose reads finished an iteration ago
        u32x4* slot = ring + (t % R) * T * 64;
        if (ASM) {
            __builtin_amdgcn_s_waitcnt(0x0F70 | (((R - 1) * T) & 15) | ((((R - 1) * T) >> 4) << 14));  // vmcnt((R-1)*T)
            const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) u32x4*)(slot + lane);
#pragma unroll
            for (int k = 0; k < T; ++k) {
                u32x4 v = lds_read_asm(a + k * 1024);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v));   // the value is not there before this: tie it in
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < T; ++k) {
                const u32x4 v = slot[k * 64 + lane];
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            }
        }
    }
    if (lane == 0) out[w] = 0;
    atomicXor(&out[w], acc);
}

__global__ void fill_kernel(unsigned* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 13);
}

template <typename F>
static double time_ms(F launch) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

int main() {
    const int wgs = 512;                       // two resident workgroups per CU
    const size_t per_wave = 1024;              // 1 MiB per wave, 2 GiB in all
    const size_t bytes = (size_t)wgs * kWaves * per_wave * 1024;
    u32x4* in; unsigned* out;
    const int nw = wgs * kWaves;
    if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&out, 4 * nw) != hipSuccess) return 1;
    fill_kernel<<<2048, 256>>>(reinterpret_cast<unsigned*>(in), bytes / 4);
    std::vector<unsigned> ref(nw), got(nw);
    bool have_ref = false;
    int bad_total = 0;
    auto report = [&](const char* name, double ms, int vgpr_note) {
        hipMemcpy(got.data(), out, 4 * nw, hipMemcpyDeviceToHost);
        int bad = 0;
        if (!have_ref) { ref = got; have_ref = true; }
        else for (int i = 0; i < nw; ++i) bad += got[i] != ref[i];
        bad_total += bad;
        printf("%-50s %7.1f us  %5.2f TB/s  staging VGPRs/lane %2d  per-wave checksums differing from the first line: %d\n",
               name, ms * 1e3, bytes / ms / 1e9, vgpr_note, bad);
    };
    report("reg, 4 loads in flight", time_ms([&] { reg_kernel<4><<<wgs, kT>>>(in, out, per_wave); }), 16);
    report("reg, 8 loads in flight", time_ms([&] { reg_kernel<8><<<wgs, kT>>>(in, out, per_wave); }), 32);
    report("reg, 16 loads in flight", time_ms([&] { reg_kernel<16><<<wgs, kT>>>(in, out, per_wave); }), 64);
#define RING(R, T, ASM, label)                                                                                   \
    {                                                                                                            \
        const int ldsb = kWaves * R * T * 1024;                                                                  \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&ring_kernel<R, T, ASM>),                              \
                            hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);                                   \
        report(label, time_ms([&] { ring_kernel<R, T, ASM><<<wgs, kT, ldsb>>>(in, out, per_wave); }), 0);        \
    }
    RING(3, 4, true, "ring 3 x 4 KiB per wave, asm reads")
    RING(4, 4, true, "ring 4 x 4 KiB per wave, asm reads")
    RING(5, 4, true, "ring 5 x 4 KiB per wave (80 KiB/WG: 1 WG/CU), asm")
    RING(4, 2, true, "ring 4 x 2 KiB per wave, asm reads")
    RING(4, 4, false, "ring 4 x 4 KiB per wave, compiler-visible reads")
    hipFree(in); hipFree(out);
    return bad_total != 0;
}
