// The distortion op's noise plane on the GPU: np.random.RandomState(seed).normal(loc, scale, n).astype(np.uint8)
// (srcs/preprocessing/image_augmenter.py:121-123), the stream lf_legacy_normal_u8 (lf_jpeg_host.cpp) restates on
// the host — MT19937 seeded by init_genrand, 53-bit doubles from two draws, the polar method with rejection, an
// accepted pair giving f*x2 then f*x1 — here one workgroup per image.
//
// What is serial in it and what is not:
//   the generator: word n depends on words n-624, n-623 and n-227, so 227 words can be made at once and a chain of
//     ~1,700 such steps is behind a 224 x 224 x 3 plane (383 k words).  The words live in a ring in LDS; one barrier
//     per step (new words go to fresh places, nothing is updated in place);
//   the rejection: which output an accepted attempt lands on depends on how many attempts before it were accepted —
//     a prefix count (ballots per wave, a sum over the waves), 1024 attempts per round;
//   everything else (tempering, the doubles, log / divide / sqrt, the cast) is per attempt.
//
// Exactness: every operation up to r2 = x1*x1 + x2*x2 is exact or correctly rounded IEEE arithmetic on both sides
// (this file is compiled without FMA contraction, as the host's x86-64 code is), so the same attempts are accepted.
// f = sqrt(-2 log(r2) / r2): divide and sqrt are correctly rounded, log is not (glibc's and the device library's
// both stay within an ulp of the true value but need not agree in the last bit), which can move loc + scale * f * x
// by a few ulps.  The cast truncates, so that only matters within ~1e-14 of an integer: a value closer than 1e-9
// to one raises the image's flag and the caller has that plane made on the host (about one image in 3,000).
#include "lf_common.h"

namespace {

constexpr int kNT = 1024;        // threads = attempts per round
constexpr int kRingW = 8192;     // words of generator output kept in LDS
constexpr double kGuard = 1e-9;

__device__ __forceinline__ uint32_t temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

__device__ __forceinline__ double to_double(uint32_t w0, uint32_t w1) {
    const int32_t a = (int32_t)(w0 >> 5), b = (int32_t)(w1 >> 6);
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

__global__ __launch_bounds__(kNT) void legacy_normal_kernel(const uint32_t* __restrict__ seeds, double loc, double scale,
                                                           size_t count, uint8_t* __restrict__ out, size_t out_stride,
                                                           int* __restrict__ flags) {
    __shared__ uint32_t X[kRingW];
    __shared__ uint32_t wave_cnt[kNT / 64];
    __shared__ int flag_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t img = blockIdx.x;
    if (tid == 0) {   // init_genrand
        uint32_t v = seeds[img];
        X[0] = v;
        for (uint32_t i = 1; i < 624; ++i) {
            v = 1812433253u * (v ^ (v >> 30)) + i;
            X[i] = v;
        }
        flag_s = 0;
    }
    __syncthreads();
    uint8_t* o = out + img * out_stride;
    const size_t need_pairs = (count + 1) / 2;
    // words [0, G) of the generator's sequence exist (the first 624 are the seeded state, outputs start behind them);
    // word C is the first one no attempt has used.  A round needs 4 * kNT words; the ring holds the 624 words the
    // next step reaches back to, the unused ones and the step being written: G - 624 - 227 .. G + 227 at the most
    // with G <= C + 4 * kNT + 226, well inside kRingW.
    size_t G = 624, C = 624, base = 0;
    // with an acceptance rate of pi/4 the attempts needed are need_pairs / 0.785 +- a few hundred: 1.3 x is out of reach
    const size_t max_rounds = (need_pairs + need_pairs / 3) / kNT + 2;
    for (size_t round = 0; base < need_pairs; ++round) {
        if (round >= max_rounds) {
            if (tid == 0) flag_s = 2;
            break;
        }
        while (G < C + 4 * kNT) {
            if (tid < 227) {
                const size_t n = G + (size_t)tid;
                const uint32_t a = X[(n - 624) % kRingW], b = X[(n - 623) % kRingW], c = X[(n - 227) % kRingW];
                const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
                X[n % kRingW] = c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            G += 227;
            __syncthreads();
        }
        const size_t w = C + 4 * (size_t)tid;
        const uint32_t w0 = temper(X[w % kRingW]), w1 = temper(X[(w + 1) % kRingW]);
        const uint32_t w2 = temper(X[(w + 2) % kRingW]), w3 = temper(X[(w + 3) % kRingW]);
        const double x1 = 2.0 * to_double(w0, w1) - 1.0, x2 = 2.0 * to_double(w2, w3) - 1.0;
        const double r2 = x1 * x1 + x2 * x2;
        const bool acc = r2 < 1.0 && r2 != 0.0;
        const unsigned long long m = __ballot(acc);
        if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (int i = 0; i < kNT / 64; ++i) {
            const uint32_t c = wave_cnt[i];
            before += i < wave ? c : 0u;
            total += c;
        }
        const size_t q = base + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (acc && q < need_pairs) {
            const double f = sqrt(-2.0 * log(r2) / r2);
            const double v0 = loc + scale * (f * x2), v1 = loc + scale * (f * x1);
            bool near = fabs(v0 - rint(v0)) < kGuard;
            o[2 * q] = (uint8_t)(int32_t)v0;
            if (2 * q + 1 < count) {
                near = near || fabs(v1 - rint(v1)) < kGuard;
                o[2 * q + 1] = (uint8_t)(int32_t)v1;
            }
            if (near) flag_s = 1;
        }
        base += total;
        C += 4 * kNT;
        __syncthreads();   // wave_cnt and the ring places behind C are free again
    }
    __syncthreads();
    if (tid == 0) flags[img] = flag_s;
}

}  // namespace

extern "C" int lf_legacy_normal_batch_u8(const uint32_t* seeds, double loc, double scale, size_t count, uint8_t* out,
                                         size_t out_stride, int n, int* flags, lf_stream_t stream) {
    LF_REQUIRE(seeds && out && flags, "lf_legacy_normal_batch: null buffer");
    LF_REQUIRE(n > 0 && count > 0 && out_stride >= count, "lf_legacy_normal_batch: bad sizes n=%d count=%zu stride=%zu", n,
               count, out_stride);
    LF_REQUIRE(scale > 0.0 && fabs(loc) + 8.0 * scale < 1e9, "lf_legacy_normal_batch: loc / scale out of range");
    legacy_normal_kernel<<<(unsigned)n, kNT, 0, lf::as_stream(stream)>>>(seeds, loc, scale, count, out, out_stride, flags);
    return lf::check_launch("lf_legacy_normal_batch");
}
