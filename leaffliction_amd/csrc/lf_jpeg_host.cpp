// Host half of the JPEG encoder: quantisation tables, markers and the Huffman coding of the coefficients
// lf_jpeg_fdct_quant_u8 left in scan order (jcmarker.c, jchuff.c with the Annex K tables; baseline,
// 4:2:0, no restart markers, JFIF 1.01 with density 1:1 — what Pillow's Image.save(path, quality=q)
// writes through libjpeg-turbo, srcs/utils/image_utils.py:49-56).  Plain C++ with no HIP in it: it is
// linked into libleafhip.so and, on its own, into libleafcodec.so, which is all a codec worker process loads.
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

extern "C" {
void lf_jpeg_quant_tables(int quality, uint8_t* lum64, uint8_t* chroma64);
size_t lf_jpeg_file_bound(int h, int w);
long lf_jpeg_write_file(const int16_t* coef, int h, int w, int quality, uint8_t* out, size_t cap);
}

namespace {

const uint8_t kStdLum[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                             14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                             18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                             49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t kStdChr[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                             99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                             99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
const uint8_t kNatural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                              41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                              30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
const uint8_t kDcLumBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcChrBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kAcLumVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kAcChrBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kAcChrVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
    0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1,
    0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct Huff {
    uint16_t code[256];
    uint8_t len[256];
};

void build(Huff& t, const uint8_t* bits, const uint8_t* vals) {
    memset(&t, 0, sizeof(t));
    unsigned code = 0;
    int k = 0;
    for (int l = 1; l <= 16; ++l) {
        for (int i = 0; i < bits[l - 1]; ++i, ++k, ++code) {
            t.code[vals[k]] = (uint16_t)code;
            t.len[vals[k]] = (uint8_t)l;
        }
        code <<= 1;
    }
}

struct Tables {
    Huff dc[2], ac[2];
    Tables() {
        build(dc[0], kDcLumBits, kDcVals);
        build(dc[1], kDcChrBits, kDcVals);
        build(ac[0], kAcLumBits, kAcLumVals);
        build(ac[1], kAcChrBits, kAcChrVals);
    }
};

// Bit writer in the manner of libjpeg-turbo's jchuff.c: a 64-bit accumulator, written out eight bytes at a
// time, with the 0xFF -> 0xFF 0x00 stuffing taken off the fast path by one test on the whole word.
struct Writer {
    uint8_t* p;
    uint8_t* end;
    uint64_t acc = 0;
    int free_bits = 64;
    bool overflow = false;
    void emit8(uint64_t v) {
        if (p + 16 > end) {
            overflow = true;
            return;
        }
        if (v & 0x8080808080808080ull & ~(v + 0x0101010101010101ull)) {   // some byte may be 0xFF
            for (int s = 56; s >= 0; s -= 8) {
                const uint8_t byte = (uint8_t)(v >> s);
                *p++ = byte;
                if (byte == 0xFF) *p++ = 0;
            }
        } else {
            const uint64_t be = __builtin_bswap64(v);
            memcpy(p, &be, 8);
            p += 8;
        }
    }
    void put(uint64_t code, int len) {   // len <= 32; bits above len in `code` must be zero
        free_bits -= len;
        if (free_bits < 0) {
            acc = (acc << (len + free_bits)) | (code >> -free_bits);
            emit8(acc);
            free_bits += 64;
            acc = code;   // the bits already written sit above the valid ones and are shifted out later
        } else {
            acc = (acc << len) | code;
        }
    }
    void flush() {   // whole bytes out, the last one padded with ones
        int bits = 64 - free_bits;
        if (p + 20 > end) {
            overflow = true;
            return;
        }
        uint64_t v = bits ? acc << free_bits : 0;   // valid bits at the top
        while (bits > 0) {
            uint8_t byte = (uint8_t)(v >> 56);
            if (bits < 8) byte |= (uint8_t)(0xFFu >> bits);
            *p++ = byte;
            if (byte == 0xFF) *p++ = 0;
            v <<= 8;
            bits -= 8;
        }
        acc = 0;
        free_bits = 64;
    }
};

inline int nbits_of(int a) { return a ? 32 - __builtin_clz((unsigned)a) : 0; }

// bit k set <=> zz[k] != 0
inline uint64_t nonzero_mask(const int16_t* zz) {
#if defined(__SSE2__)
    const __m128i zero = _mm_setzero_si128();
    uint64_t m = 0;
    for (int i = 0; i < 4; ++i) {
        const __m128i a = _mm_cmpeq_epi16(_mm_loadu_si128(reinterpret_cast<const __m128i*>(zz + 16 * i)), zero);
        const __m128i b = _mm_cmpeq_epi16(_mm_loadu_si128(reinterpret_cast<const __m128i*>(zz + 16 * i + 8)), zero);
        m |= (uint64_t)(uint16_t)_mm_movemask_epi8(_mm_packs_epi16(a, b)) << (16 * i);
    }
    return ~m;
#else
    uint64_t m = 0;
    for (int k = 0; k < 64; ++k) m |= (uint64_t)(zz[k] != 0) << k;
    return m;
#endif
}

inline int code_block(Writer& wr, const int16_t* zz, int last_dc, const Huff& dc, const Huff& ac) {
    const int diff = zz[0] - last_dc;
    int nb = nbits_of(diff < 0 ? -diff : diff);
    // a negative value is sent as value - 1 in nb bits
    wr.put(((uint64_t)dc.code[nb] << nb) | ((unsigned)(diff + (diff >> 31)) & ((1u << nb) - 1u)), dc.len[nb] + nb);
    uint64_t m = nonzero_mask(zz) & ~1ull;
    int prev = 0;
    while (m) {
        const int k = __builtin_ctzll(m);
        m &= m - 1;
        int run = k - prev - 1;
        prev = k;
        while (run > 15) {
            wr.put(ac.code[0xF0], ac.len[0xF0]);
            run -= 16;
        }
        const int v = zz[k];
        nb = nbits_of(v < 0 ? -v : v);
        const int sym = (run << 4) | nb;
        wr.put(((uint64_t)ac.code[sym] << nb) | ((unsigned)(v + (v >> 31)) & ((1u << nb) - 1u)), ac.len[sym] + nb);
    }
    if (prev != 63) wr.put(ac.code[0], ac.len[0]);
    return zz[0];
}

inline uint8_t* put_bytes(uint8_t* p, const void* src, size_t n) {
    memcpy(p, src, n);
    return p + n;
}

}  // namespace

extern "C" {

void lf_jpeg_quant_tables(int quality, uint8_t* lum64, uint8_t* chroma64) {
    const int q = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
    const int scale = q < 50 ? 5000 / q : 200 - 2 * q;   // jpeg_quality_scaling
    for (int i = 0; i < 64; ++i) {
        long a = ((long)kStdLum[i] * scale + 50) / 100, b = ((long)kStdChr[i] * scale + 50) / 100;
        lum64[i] = (uint8_t)(a < 1 ? 1 : (a > 255 ? 255 : a));      // force_baseline
        chroma64[i] = (uint8_t)(b < 1 ? 1 : (b > 255 ? 255 : b));
    }
}

size_t lf_jpeg_file_bound(int h, int w) {
    // markers (623 bytes) + at most 26 bits per coefficient (16-bit code + 10 value bits... the DC takes
    // 9 + 11), every byte possibly stuffed: 8 bytes per coefficient is far above anything the coder emits
    return (size_t)1024 + (size_t)h * w * 3 / 2 * 8;
}

long lf_jpeg_write_file(const int16_t* coef, int h, int w, int quality, uint8_t* out, size_t cap) {
    static const Tables tb;
    if (!coef || !out || h <= 0 || w <= 0 || h % 16 || w % 16 || h > 65535 || w > 65535 || cap < 1024) return -1;
    uint8_t lum[64], chr[64];
    lf_jpeg_quant_tables(quality, lum, chr);
    uint8_t* p = out;
    static const uint8_t soi_app0[] = {0xFF, 0xD8, 0xFF, 0xE0, 0x00, 0x10, 'J', 'F', 'I', 'F', 0x00,
                                       0x01, 0x01, 0x00, 0x00, 0x01, 0x00, 0x01, 0x00, 0x00};
    p = put_bytes(p, soi_app0, sizeof(soi_app0));
    for (int t = 0; t < 2; ++t) {
        const uint8_t hd[5] = {0xFF, 0xDB, 0x00, 0x43, (uint8_t)t};
        p = put_bytes(p, hd, 5);
        const uint8_t* q = t ? chr : lum;
        for (int i = 0; i < 64; ++i) *p++ = q[kNatural[i]];
    }
    const uint8_t sof[19] = {0xFF, 0xC0, 0x00, 0x11, 0x08, (uint8_t)(h >> 8), (uint8_t)h, (uint8_t)(w >> 8), (uint8_t)w,
                             0x03, 0x01, 0x22, 0x00, 0x02, 0x11, 0x01, 0x03, 0x11, 0x01};
    p = put_bytes(p, sof, 19);
    const struct {
        uint8_t id;
        const uint8_t* bits;
        const uint8_t* vals;
        int nvals;
    } dht[4] = {{0x00, kDcLumBits, kDcVals, 12}, {0x10, kAcLumBits, kAcLumVals, 162},
                {0x01, kDcChrBits, kDcVals, 12}, {0x11, kAcChrBits, kAcChrVals, 162}};
    for (const auto& d : dht) {
        const int len = 2 + 1 + 16 + d.nvals;
        const uint8_t hd[5] = {0xFF, 0xC4, (uint8_t)(len >> 8), (uint8_t)len, d.id};
        p = put_bytes(p, hd, 5);
        p = put_bytes(p, d.bits, 16);
        p = put_bytes(p, d.vals, d.nvals);
    }
    static const uint8_t sos[14] = {0xFF, 0xDA, 0x00, 0x0C, 0x03, 0x01, 0x00, 0x02, 0x11, 0x03, 0x11, 0x00, 0x3F, 0x00};
    p = put_bytes(p, sos, 14);
    Writer wr{p, out + cap - 2};   // the two bytes of EOI stay free
    int last[3] = {0, 0, 0};
    const long mcus = (long)(h / 16) * (w / 16);
    for (long m = 0; m < mcus; ++m) {
        const int16_t* b = coef + m * 6 * 64;
        for (int k = 0; k < 4; ++k) last[0] = code_block(wr, b + 64 * k, last[0], tb.dc[0], tb.ac[0]);
        last[1] = code_block(wr, b + 64 * 4, last[1], tb.dc[1], tb.ac[1]);
        last[2] = code_block(wr, b + 64 * 5, last[2], tb.dc[1], tb.ac[1]);
    }
    wr.flush();
    if (wr.overflow) return -1;
    p = wr.p;
    *p++ = 0xFF;
    *p++ = 0xD9;
    return (long)(p - out);
}

}  // extern "C"
