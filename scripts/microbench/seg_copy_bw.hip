// The bf16 tile kernels' memory shape as a plain copy: read SHORT contiguous segments scattered over one buffer and
// write segments of the same length scattered over another (an NCHW tile row of one channel plane = tile width x 2
// bytes; the next piece lies a row or a plane further on) — how much of the HBM rate survives the MIX of reads and
// writes at that granularity?  (seg_bw.hip measures reads alone.)
//   hipcc -O3 --offload-arch=gfx950 seg_copy_bw.hip -o seg_copy_bw && ./seg_copy_bw
// For each segment length L: 1 GiB read + 1 GiB written per launch as 1 GiB / L segments each; segment k of the source
// sits at scramble(k) * STRIDE (STRIDE = 4 KiB, or L when longer), the destination likewise with another scramble;
// every lane moves 16 bytes, L/16 consecutive lanes share a segment; 512 workgroups x 256 threads (two per CU, the
// residency of the convolution kernels) and 2,048; 4 loads in flight per lane, then their 4 stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct alignas(16) V16 { unsigned a[4]; };

template <bool WRITE>
__global__ __launch_bounds__(256) void seg_copy(const char* __restrict__ src, char* __restrict__ dst,
                                                unsigned long long nseg, unsigned seg_bytes, unsigned long long stride,
                                                unsigned long long* __restrict__ sink) {
    const unsigned lps = seg_bytes / 16;
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long nthreads = (unsigned long long)gridDim.x * blockDim.x;
    const unsigned long long items = nseg * lps;
    unsigned long long acc = 0;
    for (unsigned long long i0 = tid; i0 < items; i0 += nthreads * 4) {
        V16 v[4];
        unsigned long long to[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned long long i = i0 + (unsigned long long)u * nthreads;
            const unsigned long long seg = i / lps, within = i - seg * lps;
            const unsigned long long from = (seg * 0x9E3779B1ull) % nseg;
            to[u] = ((seg * 0x85EBCA6Bull) % nseg) * stride + within * 16;
            v[u] = i < items ? *reinterpret_cast<const V16*>(src + from * stride + within * 16) : V16{};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned long long i = i0 + (unsigned long long)u * nthreads;
            if (WRITE) {
                if (i < items) *reinterpret_cast<V16*>(dst + to[u]) = v[u];
            } else {
                acc += v[u].a[0] + v[u].a[3];
            }
        }
    }
    if (!WRITE && acc == 0x1234567ull) sink[0] = acc;
}

template <bool WRITE>
double run(const char* src, char* dst, unsigned long long* sink, unsigned seg, unsigned long long total, int grid) {
    const unsigned long long nseg = total / seg, stride = seg > 4096 ? seg : 4096;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    seg_copy<WRITE><<<grid, 256>>>(src, dst, nseg, seg, stride, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int it = 0; it < 3; ++it) seg_copy<WRITE><<<grid, 256>>>(src, dst, nseg, seg, stride, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return 3.0 * total * (WRITE ? 2 : 1) / (ms * 1e-3) / 1e12;
}

int main() {
    const unsigned long long cap = 8ull << 30;
    char *src, *dst;
    CK(hipMalloc(&src, cap));
    CK(hipMalloc(&dst, cap));
    CK(hipMemset(src, 1, cap));
    CK(hipMemset(dst, 2, cap));
    unsigned long long* sink;
    CK(hipMalloc(&sink, 8));
    printf("segment_bytes  read-only TB/s (512 wg / 2048 wg)   read+write TB/s, both directions counted (512 wg / 2048 wg)\n");
    const unsigned segs[] = {128, 256, 512, 1024, 4096, 65536};
    for (unsigned s : segs) {
        unsigned long long total = 1ull << 30;
        const unsigned long long stride = s > 4096 ? s : 4096;
        while ((total / s) * stride > cap) total >>= 1;   // power-of-two segment counts: the scrambles stay bijective
        const double r1 = run<false>(src, dst, sink, s, total, 512), r2 = run<false>(src, dst, sink, s, total, 2048);
        const double c1 = run<true>(src, dst, sink, s, total, 512), c2 = run<true>(src, dst, sink, s, total, 2048);
        printf("%8u       %6.2f / %6.2f                      %6.2f / %6.2f\n", s, r1, r2, c1, c2);
        fflush(stdout);
    }
    return 0;
}
