// Huffman decoding of baseline JPEG scans on the GPU: the step between a file's bytes and the coefficients
// lf_jpeg_idct_rgb_u8 takes (Image.open(path).convert("RGB"), srcs/utils/image_utils.py:19-33 — the balancer's
// input step; libjpeg's jdhuff.c decode_mcu is the algorithm, lf_jpeg_read_file in lf_jpeg_host.cpp the host
// restatement these kernels are tested against, coefficient for coefficient).
//
// The host worker (lf_jpeg_scan_prepare, lf_jpeg_host.cpp) has parsed the markers, undone the 0xFF00 stuffing and
// taken the RSTn markers out, so a reader here has no byte-wise special cases.
//
// A scan is one serial bit stream: where a code starts depends on every code before it.  Two kernels:
//
// jpeg_huffman_par_kernel — one workgroup per image, 256 threads, the stream staged in LDS and cut into 256
//   subsequences of equal bit length (a scan over 96 KB stays in global memory and is read through the L2).  Huffman streams re-synchronise by themselves: a decoder started at a wrong
//   bit, in a wrong place of a block, falls into step with the true sequence of codes after a few dozen symbols.
//   So every thread decodes its own subsequence from a GUESSED entry state (bit position, index in the block,
//   block of the MCU) and hands the state it leaves with to its right-hand neighbour as that one's entry state;
//   whoever's entry state changed decodes again; thread 0's entry state is the true one, so after r rounds the
//   first r subsequences are certainly right, and in practice everything is after a few (Weissenberger & Schmidt's
//   scheme, ICPP 2018, restated as a fixed point per workgroup).  Then the blocks each subsequence completed are
//   prefix-summed into block numbers, a last decode writes the coefficients where they belong (DC terms still as
//   differences), and three waves turn the DC differences into values with a scan per component.  A file with restart
//   markers needs none of the guessing: its intervals start at known states, one thread decodes each.
//
// jpeg_huffman_seq_kernel — one LANE per image, every trip of the loop one Huffman symbol whatever the lane is in
//   the middle of.  Correct for everything the host prepares (restart intervals, scans of any length),
//   but a wave executes the union of its lanes' paths at one instruction every ~8 cycles: 28 ms for 64..256 images
//   of 224 x 224 whether the bytes come straight from global memory or through an LDS ring (both were measured:
//   9 M instructions per wave, 0.6 M cycles of waiting in 68 M — issue-bound, not latency-bound).  It takes the
//   images the first kernel marks with status 4.
//
// LDS tables: per Huffman table a 12-bit lookahead (code length << 8 | symbol), and for the codes of 13..16 bits
// one 16-entry second level per 12-bit prefix (bit 15 of the first-level entry set, index in the low bits).
#include "lf_common.h"

namespace {

constexpr int kLook = 12;
constexpr int kSub = 64;                    // second-level tables per Huffman table (a table with more: status 3)
constexpr uint32_t kMagic = 0x4353464Cu;
constexpr int kPT = 256;                    // threads of the parallel kernel = subsequences per image
constexpr uint32_t kStreamCap = 96u << 10;  // bytes of scan the parallel kernel stages in LDS
constexpr int kHT = 64;                     // sequential kernel: one wave, one image per lane

struct HuffTab {
    uint16_t fast[4][1 << kLook];
    uint16_t sub[4][kSub * 16];
};

// All `nthr` threads of the workgroup call this with the same arguments.  dht: four times 16 counts + 256 symbols.
// A workgroup of four waves builds the four tables side by side (one wave each), a single wave one after the other.
__device__ bool build_tables(HuffTab& T, const uint8_t* dht, int tid, int nthr) {
    for (int i = tid; i < 4 * (1 << kLook); i += nthr) (&T.fast[0][0])[i] = 0;
    for (int i = tid; i < 4 * kSub * 16; i += nthr) (&T.sub[0][0])[i] = 0;
    __syncthreads();
    const bool split = nthr >= 256;
    const int lanes = split ? 64 : nthr, lid = split ? (tid & 63) : tid, tstep = split ? nthr / 64 : 1;
    bool ok = true;
    for (int t = split ? (tid >> 6) : 0; t < 4 && ok; t += tstep) {
        const uint8_t* bits = dht + t * 272;
        int code = 0, k = 0, nsub = 0, last_prefix = -1;
        for (int l = 1; l <= 16 && ok; ++l) {
            const int cnt = bits[l - 1];
            for (int i = 0; i < cnt; ++i, ++k, ++code) {
                if (code >= (1 << l) || k >= 256) {
                    ok = false;
                    break;
                }
                const uint16_t e = (uint16_t)((l << 8) | bits[16 + k]);
                if (l <= kLook) {
                    const int first = code << (kLook - l), span = 1 << (kLook - l);
                    for (int f = lid; f < span; f += lanes) T.fast[t][first + f] = e;
                } else {
                    const int rem = l - kLook, prefix = code >> rem;
                    if (prefix != last_prefix) {   // codes come in increasing order: a new prefix is a new second level
                        if (nsub >= kSub) {
                            ok = false;
                            break;
                        }
                        last_prefix = prefix;
                        if (lid == 0) T.fast[t][prefix] = (uint16_t)(0x8000 | nsub);
                        ++nsub;
                    }
                    const int first = (code & ((1 << rem) - 1)) << (4 - rem), span = 1 << (4 - rem);
                    if (lid < span) T.sub[t][(nsub - 1) * 16 + first + lid] = e;
                }
            }
            code <<= 1;
        }
    }
    return __syncthreads_and(ok) != 0;
}

__device__ __forceinline__ unsigned lookup(const HuffTab& T, int t, uint64_t buf) {
    unsigned e = T.fast[t][(unsigned)(buf >> (64 - kLook))];
    if (e & 0x8000u) e = T.sub[t][((e & 0x7FFFu) << 4) | ((unsigned)(buf >> (64 - kLook - 4)) & 15u)];
    return e;   // 0: no such code
}

struct ScanHeader {
    uint32_t restart, nint, data_off, data_len;
    uint64_t hash;
    bool ok;
};

__device__ __forceinline__ ScanHeader read_header(const uint8_t* a, int h, int w, size_t aux, size_t stride) {
    ScanHeader H;
    uint32_t magic;
    uint16_t hw[2];
    __builtin_memcpy(&magic, a, 4);
    __builtin_memcpy(hw, a + 4, 4);
    __builtin_memcpy(&H.restart, a + 8, 4);
    __builtin_memcpy(&H.nint, a + 12, 4);
    __builtin_memcpy(&H.hash, a + 16, 8);
    __builtin_memcpy(&H.data_off, a + 24, 4);
    __builtin_memcpy(&H.data_len, a + 28, 4);
    const uint32_t mcus = (uint32_t)(h / 16) * (uint32_t)(w / 16);
    const uint32_t want_int = H.restart ? (mcus + H.restart - 1) / H.restart : 1u;
    H.ok = magic == kMagic && hw[0] == h && hw[1] == w && H.nint == want_int && H.data_off % 16 == 0 &&
           H.data_off >= 1120 + 4 * (H.nint + 1) && aux + H.data_off + (size_t)H.data_len + 16 <= stride;
    return H;
}

__global__ __launch_bounds__(256) void jpeg_huff_zero_kernel(uint8_t* __restrict__ slots, size_t stride, int n,
                                                            size_t bytes) {
    const size_t per = bytes / 16;
    const size_t total = per * (size_t)n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t img = i / per, o = i - img * per;
        *reinterpret_cast<uint4*>(slots + img * stride + 256 + o * 16) = make_uint4(0, 0, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The parallel kernel
// ---------------------------------------------------------------------------------------------------------------
struct ParLds {
    HuffTab T;
    uint64_t stream[kStreamCap / 8 + 4];
    uint32_t state[kPT + 1];   // a subsequence's entry state: bit position << 9 | index in the block << 3 | block of the MCU
    uint32_t cnt[kPT];         // blocks the subsequence's chain completed
    uint32_t scan[2][kPT];
    int fail;
};

// the eight bytes at byte position `pos` of the staged stream, first byte on top
__device__ __forceinline__ uint64_t stream8_be(const uint64_t* stream, uint32_t pos) {
    const uint32_t wi = pos >> 3, sh = (pos & 7u) * 8u;
    const uint64_t lo = stream[wi], hi = stream[wi + 1];
    const uint64_t v = sh ? (lo >> sh) | (hi << (64u - sh)) : lo;
    return __builtin_bswap64(v);
}

// A speculative chain from `entry` while symbols START before bit `stop`: only where it ends up and how many blocks it
// completed matter, so a trip is a table lookup and a handful of selects — the code's length plus its value bits to step
// over, and how far the index in the block moves (DC: to 1; run/size: run + 1; ZRL: 16; end of block: to 64).  What cannot
// be (no such code) is stepped over by a bit.  On a well-formed stream it walks through the states decode_span does.
__device__ __forceinline__ uint32_t spec_span(const HuffTab& T, const uint64_t* stream, uint32_t entry, uint32_t stop,
                                              uint32_t& blocks) {
    uint32_t p = entry >> 9;
    int k = (int)((entry >> 3) & 63u), b6 = (int)(entry & 7u);
    uint32_t bytepos = p >> 3;
    uint64_t buf = stream8_be(stream, bytepos) << (p & 7u);
    int nb = 64 - (int)(p & 7u);
    bytepos += 8;
    uint32_t done = 0;
    while (p < stop) {
        if (nb < 32) {
            const int take = (64 - nb) >> 3;
            const uint64_t v = stream8_be(stream, bytepos);
            buf |= (take == 8 ? v : v & ~(~0ull >> (8 * take))) >> nb;
            bytepos += (uint32_t)take;
            nb += 8 * take;
        }
        const unsigned e = lookup(T, (b6 < 4 ? 0 : 2) + (k > 0 ? 1 : 0), buf);
        const int size = (int)(e & 15u), run = (int)((e >> 4) & 15u);
        const int adv = e ? (int)(e >> 8) + size : 1;
        const int dk = !e ? 0 : (k == 0 ? 1 : (size ? run + 1 : (run == 15 ? 16 : 64)));
        buf <<= adv;
        nb -= adv;
        k += dk;
        p = bytepos * 8u - (uint32_t)nb;
        if (k >= 64) {
            k = 0;
            b6 = b6 == 5 ? 0 : b6 + 1;
            ++done;
        }
    }
    blocks = done;
    return (p << 9) | ((uint32_t)k << 3) | (uint32_t)b6;
}

// The true chain from `entry` while symbols START before bit `stop`: coefficients go to coef[block][index] (DC terms as
// differences), anything that cannot be (no such code, a DC size over 11, a run past the end of the block, a last
// block that ends behind the file's bits) sets *fail, and the chain ends with block `total`.
__device__ __forceinline__ uint32_t decode_span(const HuffTab& T, const uint64_t* stream, uint32_t entry, uint32_t stop,
                                                uint32_t& blocks, int16_t* coef, uint32_t blk, uint32_t total,
                                                uint32_t total_bits, int* fail) {
    uint32_t p = entry >> 9;
    int k = (int)((entry >> 3) & 63u), b6 = (int)(entry & 7u);
    uint32_t bytepos = p >> 3;
    uint64_t buf = stream8_be(stream, bytepos) << (p & 7u);
    int nb = 64 - (int)(p & 7u);
    bytepos += 8;
    uint32_t done = 0;
    bool bad = false;
    while (p < stop && blk < total && !bad) {
        if (nb < 32) {   // a code (<= 16 bits) and its value bits (<= 15) per refill
            const int take = (64 - nb) >> 3;   // 4..8 whole bytes
            const uint64_t v = stream8_be(stream, bytepos);
            buf |= (take == 8 ? v : v & ~(~0ull >> (8 * take))) >> nb;
            bytepos += (uint32_t)take;
            nb += 8 * take;
        }
        const int t = (b6 < 4 ? 0 : 2) + (k > 0 ? 1 : 0);
        const unsigned e = lookup(T, t, buf);
        if (e == 0) {   // no such code: the chain ends here
            bad = true;
            buf <<= 1;
            nb -= 1;
        } else {
            const int l = (int)(e >> 8), sym = (int)(e & 255u);
            buf <<= l;
            nb -= l;
            int size = k == 0 ? sym : (sym & 15);
            const int run = k == 0 ? 0 : sym >> 4;
            if (k == 0 && size > 11) {
                bad = true;
                size &= 7;
            }
            int val = 0;
            if (size) {   // receive + extend
                const int v = (int)((buf >> 1) >> (63 - size));
                val = v < (1 << (size - 1)) ? v - (1 << size) + 1 : v;
                buf <<= size;
                nb -= size;
            }
            if (k == 0) {
                if (val && !bad) coef[(size_t)blk * 64] = (int16_t)val;   // a DC difference for now
                k = 1;
            } else if (size == 0) {
                k = run == 15 ? k + 16 : 64;   // ZRL | end of block
            } else {
                k += run;
                if (k > 63) {
                    bad = true;
                    k = 64;
                } else {
                    if (!bad) coef[(size_t)blk * 64 + k] = (int16_t)val;
                    ++k;
                }
            }
        }
        p = bytepos * 8u - (uint32_t)nb;
        if (k >= 64) {
            k = 0;
            b6 = b6 == 5 ? 0 : b6 + 1;
            ++done;
            ++blk;
            // the last block must have ended inside the file's bits (lf_jpeg_read_file's Reader::starved)
            if (blk == total && p > total_bits) bad = true;
        }
    }
    blocks = done;
    if (bad) *fail = 1;
    return (p << 9) | ((uint32_t)k << 3) | (uint32_t)b6;
}

// One restart interval, whole, by one thread: its start is a byte boundary with fresh predictions, so its state is known
// and it is decoded for real at once (DC terms as values).  The interval must be used up to its padding bits (fewer
// than eight left, none taken from behind it): anything else goes back to the host decoder for its verdict, as in
// the sequential kernel.
__device__ __forceinline__ void decode_interval(const HuffTab& T, const uint64_t* stream, uint32_t byte_begin,
                                                uint32_t byte_end, int16_t* coef, uint32_t blk, uint32_t nblocks, int* fail) {
    const uint32_t end_bits = byte_end * 8u, last = blk + nblocks;
    uint32_t bytepos = byte_begin;
    uint64_t buf = stream8_be(stream, bytepos);
    int nb = 64;
    bytepos += 8;
    int k = 0, b6 = 0, pred0 = 0, pred1 = 0, pred2 = 0;
    bool bad = false;
    while (blk < last && !bad) {
        if (nb < 32) {
            const int take = (64 - nb) >> 3;
            const uint64_t v = stream8_be(stream, bytepos);
            buf |= (take == 8 ? v : v & ~(~0ull >> (8 * take))) >> nb;
            bytepos += (uint32_t)take;
            nb += 8 * take;
        }
        const unsigned e = lookup(T, (b6 < 4 ? 0 : 2) + (k > 0 ? 1 : 0), buf);
        if (e == 0) {
            bad = true;
            break;
        }
        const int l = (int)(e >> 8), sym = (int)(e & 255u);
        buf <<= l;
        nb -= l;
        const int size = k == 0 ? sym : (sym & 15), run = k == 0 ? 0 : sym >> 4;
        if (k == 0 && size > 11) {
            bad = true;
            break;
        }
        int val = 0;
        if (size) {
            const int v = (int)((buf >> 1) >> (63 - size));
            val = v < (1 << (size - 1)) ? v - (1 << size) + 1 : v;
            buf <<= size;
            nb -= size;
        }
        if (k == 0) {
            const int comp = b6 < 4 ? 0 : b6 - 3;
            int pr = comp == 0 ? pred0 : (comp == 1 ? pred1 : pred2);
            pr += val;
            pred0 = comp == 0 ? pr : pred0;
            pred1 = comp == 1 ? pr : pred1;
            pred2 = comp == 2 ? pr : pred2;
            if (pr) coef[(size_t)blk * 64] = (int16_t)pr;
            k = 1;
        } else if (size == 0) {
            k = run == 15 ? k + 16 : 64;
        } else {
            k += run;
            if (k > 63) {
                bad = true;
                break;
            }
            coef[(size_t)blk * 64 + k] = (int16_t)val;
            ++k;
        }
        if (k >= 64) {
            k = 0;
            b6 = b6 == 5 ? 0 : b6 + 1;
            ++blk;
        }
    }
    const uint32_t p = bytepos * 8u - (uint32_t)nb;
    if (bad || p > end_bits || end_bits - p >= 8u) *fail = 1;
}

__global__ __launch_bounds__(kPT) void jpeg_huffman_par_kernel(uint8_t* __restrict__ slots, size_t stride, int h, int w,
                                                              size_t aux, int* __restrict__ status) {
    __shared__ ParLds S;
    const int tid = threadIdx.x, img = blockIdx.x;
#ifdef LF_HUFF_STATS   // development build: cycle stamps of the phases into the first 64 bytes of the slot (over the tables)
    uint64_t stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    stamp[0] = __builtin_readcyclecounter();
#define LF_STAMP(i) stamp[i] = __builtin_readcyclecounter()
#else
#define LF_STAMP(i)
#endif
    uint8_t* slot = slots + (size_t)img * stride;
    const uint8_t* a = slot + aux;
    const ScanHeader H = read_header(a, h, w, aux, stride);   // the same for every thread
    if (!H.ok) {
        if (tid == 0) status[img] = 3;
        return;
    }
    // A scan that does not fit the LDS stage is read where it lies (through the L2: slower trips, the same walk); scans
    // too long for the 23 bits a state has for its bit position are the sequential kernel's.
    const bool in_lds = H.data_len <= kStreamCap;
    if (!in_lds && (H.data_len >= (1u << 20) - 64u || aux + H.data_off + (size_t)H.data_len + 48 > stride)) {
        if (tid == 0) status[img] = 4;
        return;
    }
    const uint64_t* gstream = reinterpret_cast<const uint64_t*>(a + H.data_off);
    if (tid == 0) S.fail = 0;
    if (!build_tables(S.T, a + 32, tid, kPT)) {
        if (tid == 0) status[img] = 3;
        return;
    }
    LF_STAMP(1);
    // the scan into LDS; the host left 16 zero bytes behind it, and what lies behind those is never looked at as data
    const uint4* src = reinterpret_cast<const uint4*>(a + H.data_off);
    const uint32_t pieces = H.data_len / 16 + 1, room = kStreamCap / 16 + 2;
    if (in_lds)
        for (uint32_t i = tid; i < room; i += kPT)
            reinterpret_cast<uint4*>(S.stream)[i] = i < pieces ? src[i] : make_uint4(0, 0, 0, 0);
    const uint32_t total_bits = H.data_len * 8u;
    const uint32_t mcus = (uint32_t)(h / 16) * (uint32_t)(w / 16), total = mcus * 6u;
    int16_t* coef = reinterpret_cast<int16_t*>(slot + 256);
    if (H.restart != 0) {
        // restart intervals are where the file itself says a decoder may start: one thread per interval, no guessing
        const uint32_t* offs = reinterpret_cast<const uint32_t*>(a + 1120);
        __syncthreads();   // the staged scan
        for (uint32_t i = tid; i < H.nint; i += kPT) {
            const uint32_t b = offs[i], e = offs[i + 1];
            const uint32_t first_mcu = i * H.restart;
            const uint32_t nm = mcus - first_mcu < H.restart ? mcus - first_mcu : H.restart;
            if (b > e || e > H.data_len || (i == 0 && b != 0)) {
                S.fail = 1;
                continue;
            }
            if (in_lds)
                decode_interval(S.T, S.stream, b, e, coef, first_mcu * 6u, nm * 6u, &S.fail);
            else
                decode_interval(S.T, gstream, b, e, coef, first_mcu * 6u, nm * 6u, &S.fail);
        }
        __syncthreads();
        if (tid == 0) status[img] = S.fail ? 1 : 0;
        return;
    }
    uint32_t L = ((total_bits + kPT - 1) / kPT + 31u) & ~31u;
    if (L < 128u) L = 128u;
    const uint32_t nsub = (total_bits + L - 1) / L;   // <= kPT
    S.state[tid] = ((uint32_t)tid * L) << 9;          // thread 0: bit 0, start of a block, first block of an MCU
    if (tid == 0) S.state[kPT] = 0xFFFFFFFFu;
    S.cnt[tid] = 0;
    __syncthreads();
    const bool mine = (uint32_t)tid < nsub;
    const uint32_t stop = (uint32_t)(tid + 1) * L < total_bits ? (uint32_t)(tid + 1) * L : total_bits;
    uint32_t entry = S.state[tid];
    bool dirty = mine;
    LF_STAMP(2);
    int round = 0;
    for (; round <= kPT; ++round) {
        uint32_t left = 0;
        bool changed = false;
        if (dirty) {
            uint32_t blocks;
            left = in_lds ? spec_span(S.T, S.stream, entry, stop, blocks) : spec_span(S.T, gstream, entry, stop, blocks);
            S.cnt[tid] = blocks;
            dirty = false;
            changed = (uint32_t)(tid + 1) < nsub && left != S.state[tid + 1];
        }
        if (!__syncthreads_or(changed)) break;
        if (changed) S.state[tid + 1] = left;
        __syncthreads();
        if (mine && S.state[tid] != entry) {
            entry = S.state[tid];
            dirty = true;
        }
    }
    LF_STAMP(3);
    // block numbers: exclusive prefix sum of the blocks each chain completed
    const uint32_t own = mine ? S.cnt[tid] : 0u;
    S.scan[0][tid] = own;
    __syncthreads();
    int cur = 0;
    for (int d = 1; d < kPT; d <<= 1) {
        const uint32_t v = S.scan[cur][tid] + (tid >= d ? S.scan[cur][tid - d] : 0u);
        S.scan[cur ^ 1][tid] = v;
        cur ^= 1;
        __syncthreads();
    }
    const uint32_t incl = S.scan[cur][tid], first_blk = incl - own;
    LF_STAMP(4);
    if (mine && first_blk < total) {
        uint32_t blocks;
        if (in_lds)
            decode_span(S.T, S.stream, entry, stop, blocks, coef, first_blk, total, total_bits, &S.fail);
        else
            decode_span(S.T, gstream, entry, stop, blocks, coef, first_blk, total, total_bits, &S.fail);
    }
    if (tid == kPT - 1 && incl < total) S.fail = 1;   // the scan ended before the image did
    __syncthreads();
    LF_STAMP(5);
    if (S.fail) {
        if (tid == 0) status[img] = 1;
        return;
    }
    // DC differences -> DC values: a running sum per component over its blocks in scan order (jdhuff.c's last_dc_val)
    const int wave = tid >> 6, lane = tid & 63;
    if (wave < 3) {
        const uint32_t nblk = wave == 0 ? mcus * 4u : mcus;
        int carry = 0;
        for (uint32_t base = 0; base < nblk; base += 64) {
            const uint32_t i = base + (uint32_t)lane;
            const uint32_t blk = wave == 0 ? (i >> 2) * 6u + (i & 3u) : i * 6u + 3u + (uint32_t)wave;
            const int diff = i < nblk ? (int)coef[(size_t)blk * 64] : 0;
            int v = diff;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(v, d, 64);
                if (lane >= d) v += up;
            }
            v += carry;
            if (i < nblk && (int16_t)v != (int16_t)diff) coef[(size_t)blk * 64] = (int16_t)v;
            carry = __shfl(v, 63, 64);
        }
    }
    if (tid == 0) status[img] = 0;
#ifdef LF_HUFF_STATS
    LF_STAMP(6);
    if (tid == 0) {
        uint64_t* o = reinterpret_cast<uint64_t*>(slot);
        for (int i = 1; i <= 6; ++i) o[i - 1] = stamp[i] - stamp[i - 1];
        o[6] = (uint64_t)round;
        o[7] = nsub;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// The sequential kernel
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t load8_be(const uint8_t* p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return __builtin_bswap64(v);
}

// redo != 0: only the images whose status is 4 (left by the parallel kernel).  The tables are those of the first image
// of the group of 64 that has work; an image with other tables reports status 2.
__global__ __launch_bounds__(kHT) void jpeg_huffman_seq_kernel(uint8_t* __restrict__ slots, size_t stride, int n, int h,
                                                              int w, size_t aux, int* __restrict__ status, int redo) {
    __shared__ HuffTab T;
    __shared__ int leader;
    const int lane = threadIdx.x;
    const int img = blockIdx.x * kHT + lane;
    const bool want = img < n && (!redo || status[img] == 4);
    if (lane == 0) leader = -1;
    __syncthreads();
    if (want) atomicMax(&leader, kHT - 1 - lane);   // the lowest lane that has work
    __syncthreads();
    if (leader < 0) return;
    const int lead = kHT - 1 - leader;
    const uint8_t* la = slots + ((size_t)blockIdx.x * kHT + lead) * stride + aux;
    const ScanHeader LH = read_header(la, h, w, aux, stride);
    if (!LH.ok) {   // uniform: nothing to build the tables from
        if (want) status[img] = 3;
        return;
    }
    const bool tables_ok = build_tables(T, la + 32, lane, kHT);
    if (!want) return;
    if (!tables_ok) {
        status[img] = 3;
        return;
    }
    uint8_t* slot = slots + (size_t)img * stride;
    const uint8_t* a = slot + aux;
    const ScanHeader H = read_header(a, h, w, aux, stride);
    if (!H.ok) {
        status[img] = 3;
        return;
    }
    if (H.hash != LH.hash) {
        status[img] = 2;
        return;
    }
    const uint32_t restart = H.restart, data_len = H.data_len;
    const uint32_t* offs = reinterpret_cast<const uint32_t*>(a + 1120);
    const uint8_t* sp = a + H.data_off;
    int16_t* coef = reinterpret_cast<int16_t*>(slot + 256);
    // the reader: `buf` holds nb bits at its top, the last `pad` of them zeros fed after the interval's bytes ran out
    uint32_t pos = offs[0], end = offs[1];
    if (end > data_len || pos != 0) {
        status[img] = 3;
        return;
    }
    uint64_t buf = 0, nxt = load8_be(sp + pos);
    int nb = 0, pad = 0;
    uint32_t iv = 0, until = restart;
    const uint32_t total = (uint32_t)(h / 16) * (uint32_t)(w / 16) * 6u;
    uint32_t blk = 0;
    int b6 = 0, k = 0;
    int pred0 = 0, pred1 = 0, pred2 = 0;
    int fail = 0;
    while (blk < total && !fail) {
        if (nb < 32) {
            const int take = (64 - nb) >> 3;
            const uint32_t avail = end - pos;
            const int real = avail < (uint32_t)take ? (int)avail : take;
            const uint64_t kept = real >= 8 ? nxt : (real ? nxt & ~(~0ull >> (8 * real)) : 0ull);
            buf |= kept >> nb;
            pos += (uint32_t)real;
            nb += 8 * take;
            pad += 8 * (take - real);
            nxt = load8_be(sp + pos);   // for the next refill (the slot has 16 bytes of zeros behind the data)
        }
        const int t = (b6 < 4 ? 0 : 2) + (k > 0 ? 1 : 0);
        const unsigned e = lookup(T, t, buf);
        if (e == 0) {
            fail = 1;
            break;
        }
        const int l = (int)(e >> 8), sym = (int)(e & 255u);
        buf <<= l;
        nb -= l;
        const int size = k == 0 ? sym : (sym & 15), run = k == 0 ? 0 : sym >> 4;
        if (k == 0 && size > 11) {
            fail = 1;
            break;
        }
        int val = 0;
        if (size) {   // receive + extend
            const int v = (int)((buf >> 1) >> (63 - size));
            buf <<= size;
            nb -= size;
            val = v < (1 << (size - 1)) ? v - (1 << size) + 1 : v;
        }
        if (k == 0) {
            const int comp = b6 < 4 ? 0 : b6 - 3;
            int p = comp == 0 ? pred0 : (comp == 1 ? pred1 : pred2);
            p += val;
            pred0 = comp == 0 ? p : pred0;
            pred1 = comp == 1 ? p : pred1;
            pred2 = comp == 2 ? p : pred2;
            if (p) coef[(size_t)blk * 64] = (int16_t)p;
            k = 1;
        } else if (size == 0) {
            k = run == 15 ? k + 16 : 64;   // ZRL | end of block
        } else {
            k += run;
            if (k > 63) {
                fail = 1;
                break;
            }
            coef[(size_t)blk * 64 + k] = (int16_t)val;
            ++k;
        }
        if (k >= 64) {
            k = 0;
            ++blk;
            if (++b6 == 6) {
                b6 = 0;
                if (restart && --until == 0 && blk < total) {
                    // a restart boundary: the interval must have been used up to its padding bits (anything else goes
                    // back to the host decoder for its verdict), then byte alignment, fresh predictions, next interval
                    if (nb < pad || pos != end || nb - pad >= 8) {
                        fail = 1;
                        break;
                    }
                    ++iv;   // the intervals lie back to back in the prepared data: `pos` is where the next one starts
                    end = offs[iv + 1];
                    if (offs[iv] != pos || end > data_len || end < pos) {
                        fail = 1;
                        break;
                    }
                    buf = 0;
                    nb = 0;
                    pad = 0;
                    pred0 = pred1 = pred2 = 0;
                    until = restart;
                    nxt = load8_be(sp + pos);
                }
            }
        }
    }
    // bits consumed that were never in the file: a truncated scan (lf_jpeg_read_file's Reader::starved)
    if (nb < pad) fail = 1;
    status[img] = fail;
}

}  // namespace

extern "C" {

size_t lf_jpeg_scan_aux_offset(int h, int w);   // lf_jpeg_host.cpp

int lf_jpeg_huffman_u8(void* slots, size_t stride, int n, int h, int w, int* status, int mode, lf_stream_t stream) {
    LF_REQUIRE(slots && status, "lf_jpeg_huffman: null buffer");
    LF_REQUIRE(n > 0 && h > 0 && w > 0, "lf_jpeg_huffman: bad dims n=%d h=%d w=%d", n, h, w);
    LF_REQUIRE(h % 16 == 0 && w % 16 == 0 && h < 65536 && w < 65536, "lf_jpeg_huffman: whole 16x16 MCUs only (%d x %d)",
               h, w);
    LF_REQUIRE(mode == 0 || mode == 1, "lf_jpeg_huffman: mode is 0 (parallel kernel first) or 1 (sequential kernel only)");
    const size_t aux = lf_jpeg_scan_aux_offset(h, w);
    LF_REQUIRE(stride % 16 == 0 && stride >= aux + 1120 + 32 && (reinterpret_cast<size_t>(slots) & 15) == 0,
               "lf_jpeg_huffman: slots must be 16-byte aligned, stride a multiple of 16 and hold a prepared scan");
    hipStream_t s = lf::as_stream(stream);
    const size_t bytes = (size_t)3 * h * w;
    jpeg_huff_zero_kernel<<<lf::stream_grid(bytes / 16 * (size_t)n, 256, 256u * 16u), 256, 0, s>>>(
        static_cast<uint8_t*>(slots), stride, n, bytes);
    if (mode == 0)
        jpeg_huffman_par_kernel<<<(unsigned)n, kPT, 0, s>>>(static_cast<uint8_t*>(slots), stride, h, w, aux, status);
    jpeg_huffman_seq_kernel<<<(unsigned)((n + kHT - 1) / kHT), kHT, 0, s>>>(static_cast<uint8_t*>(slots), stride, n, h, w,
                                                                           aux, status, mode == 0 ? 1 : 0);
    return lf::check_launch("lf_jpeg_huffman");
}

}  // extern "C"
