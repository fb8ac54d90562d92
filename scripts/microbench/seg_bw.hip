// How much HBM bandwidth survives when a kernel reads SHORT contiguous segments scattered over a
// large buffer — the access shape of an NCHW tile kernel (one tile row of one channel plane =
// tile_width * 2 bytes contiguous, the next piece a row or a plane further on).
//   hipcc -O3 --offload-arch=gfx950 seg_bw.hip -o seg_bw && ./seg_bw
// For each segment length L the kernel reads 2 GiB as N = 2 GiB / L segments; segment k sits at
// offset scramble(k) * STRIDE (STRIDE = 4 KiB, or L when L > 4 KiB) of an 8+ GiB buffer; every lane
// loads 16 bytes, L/16 consecutive lanes share a segment; 2048 workgroups x 256 threads, 8 loads
// in flight per lane.  Also: the same with 8-byte and 4-byte loads per lane (what bf16 tile
// kernels issue) for L = 64 .. 512.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <typename T>
__global__ __launch_bounds__(256) void seg_read(const char* __restrict__ buf, unsigned long long* __restrict__ sink,
                                                unsigned long long nseg, unsigned seg_bytes, unsigned long long stride,
                                                unsigned long long mult) {
    const unsigned lanes_per_seg = seg_bytes / sizeof(T);
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long nthreads = (unsigned long long)gridDim.x * blockDim.x;
    const unsigned long long items = nseg * lanes_per_seg;
    unsigned long long acc = 0;
    for (unsigned long long i0 = tid; i0 < items; i0 += nthreads * 8) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned long long i = i0 + (unsigned long long)u * nthreads;
            const unsigned long long seg = i / lanes_per_seg, within = i - seg * lanes_per_seg;
            const unsigned long long where = (seg * mult) % nseg;  // bijective scramble (mult odd, nseg power of two)
            v[u] = i < items ? *reinterpret_cast<const T*>(buf + where * stride + within * sizeof(T)) : T{};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned* w = reinterpret_cast<const unsigned*>(&v[u]);
            for (unsigned j = 0; j < sizeof(T) / 4; ++j) acc += w[j];
        }
    }
    if (acc == 0x1234567ull) sink[0] = acc;
}

template <typename T>
double run(const char* buf, unsigned long long* sink, unsigned seg, unsigned long long total) {
    const unsigned long long nseg = total / seg;
    const unsigned long long stride = seg > 4096 ? seg : 4096;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    seg_read<T><<<2048, 256>>>(buf, sink, nseg, seg, stride, 0x9E3779B1ull);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int it = 0; it < 3; ++it) seg_read<T><<<2048, 256>>>(buf, sink, nseg, seg, stride, 0x9E3779B1ull);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return 3.0 * total / (ms * 1e-3) / 1e12;
}

struct alignas(16) V16 { unsigned a[4]; };
struct alignas(8) V8 { unsigned a[2]; };
struct alignas(4) V4 { unsigned a[1]; };

int main() {
    const unsigned long long total = 2ull << 30;           // bytes read per launch
    const unsigned long long bufsz = (total / 64) * 4096;  // worst case: 64-byte segments at 4 KiB stride = 128 GiB?  cap below
    (void)bufsz;
    // cap the footprint at 16 GiB: segments wrap (nseg * stride is reduced modulo the buffer)
    const unsigned long long cap = 16ull << 30;
    char* buf;
    CK(hipMalloc(&buf, cap));
    CK(hipMemset(buf, 1, cap));
    unsigned long long* sink;
    CK(hipMalloc(&sink, 8));
    printf("segment_bytes  TB/s(16B/lane)  TB/s(8B/lane)  TB/s(4B/lane)\n");
    const unsigned segs[] = {64, 128, 256, 512, 1024, 2048, 4096, 16384, 65536};
    for (unsigned s : segs) {
        // keep nseg * stride within the buffer: read fewer bytes for the short segments
        unsigned long long stride = s > 4096 ? s : 4096;
        unsigned long long tot = total;
        while ((tot / s) * stride > cap) tot >>= 1;
        const double a = run<V16>(buf, sink, s, tot);
        const double b = s <= 1024 ? run<V8>(buf, sink, s, tot) : 0.0;
        const double c = s <= 1024 ? run<V4>(buf, sink, s, tot) : 0.0;
        printf("%8u  %10.2f  %10.2f  %10.2f   (%.2f GiB read per launch)\n", s, a, b, c, tot / 1073741824.0);
    }
    return 0;
}
