// Host half of the JPEG encoder: quantisation tables, markers and the Huffman coding of the coefficients
// lf_jpeg_fdct_quant_u8 left in scan order (jcmarker.c, jchuff.c with the Annex K tables; baseline,
// 4:2:0, no restart markers, JFIF 1.01 with density 1:1 — what Pillow's Image.save(path, quality=q)
// writes through libjpeg-turbo, srcs/utils/image_utils.py:49-56).  Plain C++ with no HIP in it: it is
// linked into libleafhip.so and, on its own, into libleafcodec.so, which is all a codec worker process loads.
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <math.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

extern "C" {
void lf_jpeg_quant_tables(int quality, uint8_t* lum64, uint8_t* chroma64);
size_t lf_jpeg_file_bound(int h, int w);
long lf_jpeg_write_file(const int16_t* coef, int h, int w, int quality, uint8_t* out, size_t cap);
int lf_jpeg_read_file(const uint8_t* data, size_t len, int16_t* coef, size_t coef_cap, uint16_t* qtab128, int* h, int* w);
size_t lf_jpeg_scan_aux_offset(int h, int w);
int lf_jpeg_scan_prepare(const uint8_t* data, size_t len, uint8_t* slot, size_t cap, int* h, int* w, uint64_t* hash);
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

// (code << 8 | length) of the Annex K tables, for the GPU entropy coder: dc32[2][16], ac512[2][256]
void lf_jpeg_std_huffman(uint32_t* dc32, uint32_t* ac512) {
    static const Tables tb;
    for (int t = 0; t < 2; ++t) {
        for (int i = 0; i < 16; ++i) dc32[16 * t + i] = ((uint32_t)tb.dc[t].code[i] << 8) | tb.dc[t].len[i];
        for (int i = 0; i < 256; ++i) ac512[256 * t + i] = ((uint32_t)tb.ac[t].code[i] << 8) | tb.ac[t].len[i];
    }
}

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
    return (size_t)1024 + (size_t)((h + 15) / 16) * ((w + 15) / 16) * 384 * 8;
}

static uint8_t* write_headers(uint8_t* p, int h, int w, int quality);

long lf_jpeg_wrap_scan(const uint8_t* scan, size_t scan_len, int h, int w, int quality, uint8_t* out, size_t cap) {
    if (!scan || !out || h <= 0 || w <= 0 || h > 65535 || w > 65535 || cap < 1024 + scan_len) return -1;
    uint8_t* p = write_headers(out, h, w, quality);
    memcpy(p, scan, scan_len);
    p += scan_len;
    *p++ = 0xFF;
    *p++ = 0xD9;
    return (long)(p - out);
}

long lf_jpeg_write_file(const int16_t* coef, int h, int w, int quality, uint8_t* out, size_t cap) {
    static const Tables tb;
    if (!coef || !out || h <= 0 || w <= 0 || h > 65535 || w > 65535 || cap < 1024) return -1;
    uint8_t* p = write_headers(out, h, w, quality);
    Writer wr{p, out + cap - 2};   // the two bytes of EOI stay free
    int last[3] = {0, 0, 0};
    const long mcus = (long)((h + 15) / 16) * ((w + 15) / 16);   // ragged sizes: the caller's coefficients include libjpeg's padding
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

static uint8_t* write_headers(uint8_t* p, int h, int w, int quality) {
    uint8_t lum[64], chr[64];
    lf_jpeg_quant_tables(quality, lum, chr);
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
    return p;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Reading: markers + Huffman decoding of a baseline 4:2:0 file into the same coefficient layout the encoder
// uses ([MCU][Y00 Y01 Y10 Y11 Cb Cr][64], zigzag order, still quantised) plus the file's two quantisation
// tables; dequantisation, IDCT, upsampling and colour conversion are lf_jpeg_idct_rgb_u8's (GPU).
// Everything this does not cover (progressive, other samplings, 12-bit, sizes that are not whole MCUs,
// arithmetic coding) returns 1 and the caller decodes with libjpeg as before.
// ---------------------------------------------------------------------------
namespace {

struct DHuff {
    bool present = false;
    uint8_t look_len[1 << 12], look_sym[1 << 12];   // 12-bit lookahead: at quality 95 most AC codes are 10-16 bits long
    int maxcode[18], valptr[17];
    uint8_t vals[256];
};

bool build_decoder(DHuff& t, const uint8_t* bits, const uint8_t* vals, int nvals) {
    memset(t.look_len, 0, sizeof(t.look_len));
    memcpy(t.vals, vals, (size_t)nvals);
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        t.valptr[l] = k;
        for (int i = 0; i < bits[l - 1]; ++i, ++k, ++code) {
            if (k >= nvals || code >= (1 << l)) return false;
            if (l <= 12) {
                const int first = code << (12 - l);
                for (int f = 0; f < (1 << (12 - l)); ++f) {
                    t.look_len[first + f] = (uint8_t)l;
                    t.look_sym[first + f] = vals[k];
                }
            }
        }
        t.maxcode[l] = bits[l - 1] ? code - 1 : -1;
        code <<= 1;
    }
    t.maxcode[17] = 0x7fffffff;
    // mincode folded into valptr: symbol index = valptr[l] + code - mincode[l]
    code = 0;
    k = 0;
    for (int l = 1; l <= 16; ++l) {
        t.valptr[l] = k - code;   // so that index = valptr[l] + code
        k += bits[l - 1];
        code = (code + bits[l - 1]) << 1;
    }
    t.present = true;
    return true;
}

struct Reader {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t buf = 0;
    int n = 0;
    int pad = 0;           // zero bits fed after the data ran out / a marker came: always the LAST `pad` bits of buf
    bool marker = false;   // ran into a marker: zeros are fed from here on
    // true once the decoder has CONSUMED bits that were never in the file (the scan ended early: a truncated file,
    // or a marker in the middle of the entropy-coded segment).  Look-ahead padding that is still in the buffer
    // does not count: the last MCU of an intact file is decoded with zeros queued behind its bits.
    bool starved() const { return n < pad; }
    void fill() {
        if (!marker && p + 8 <= end && n <= 56) {   // fast path: no 0xFF among the next eight bytes
            uint64_t v;
            memcpy(&v, p, 8);
            v = __builtin_bswap64(v);
            if (!(v & 0x8080808080808080ull & ~(v + 0x0101010101010101ull))) {
                const int take = (64 - n) >> 3;   // whole bytes that fit: 1..8
                const uint64_t kept = take == 8 ? v : v & ~(~0ull >> (8 * take));
                buf |= kept >> n;
                p += take;
                n += 8 * take;
                return;
            }
        }
        while (n <= 56) {
            unsigned b = 0;
            bool real = false;
            if (!marker && p < end) {
                b = *p;
                real = true;
                if (b == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) {
                        p += 2;
                    } else {
                        marker = true;
                        real = false;
                        b = 0;
                    }
                } else {
                    ++p;
                }
            }
            if (!real) pad += 8;
            buf |= (uint64_t)b << (56 - n);
            n += 8;
        }
    }
    unsigned peek(int k) { return (unsigned)(buf >> (64 - k)); }
    void skip(int k) {
        buf <<= k;
        n -= k;
    }
    void reset() {
        buf = 0;
        n = 0;
        pad = 0;
        marker = false;
    }
    // The first marker at or after the reader's position in the FILE (bits still queued in `buf` are bytes the
    // file pointer has passed): 0 when the data ends without one.  Stuffed 0xFF00 pairs and fill bytes are skipped.
    unsigned next_marker() const {
        const uint8_t* q = p;
        while (q + 1 < end) {
            if (q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF) return q[1];
            ++q;
        }
        return 0;
    }
};

inline int decode_symbol(Reader& r, const DHuff& t) {   // the caller keeps >= 32 bits in the buffer
    const unsigned look = r.peek(12);
    int l = t.look_len[look];
    if (l) {
        r.skip(l);
        return t.look_sym[look];
    }
    int code = (int)r.peek(13);
    for (l = 13; l <= 16; ++l) {
        if (code <= t.maxcode[l]) {
            r.skip(l);
            return t.vals[(t.valptr[l] + code) & 255];
        }
        code = (int)r.peek(l + 1);
    }
    return -1;
}

inline int receive_extend(Reader& r, int s) {
    const int v = (int)r.peek(s);
    r.skip(s);
    return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
}

inline bool decode_block(Reader& r, int16_t* zz, int& pred, const DHuff& dc, const DHuff& ac) {
    memset(zz, 0, 64 * sizeof(int16_t));
    if (r.n < 32) r.fill();   // a code (<= 16 bits) and its value bits (<= 11) per refill
    int s = decode_symbol(r, dc);
    if (s < 0 || s > 11) return false;
    if (s) pred += receive_extend(r, s);
    zz[0] = (int16_t)pred;
    for (int k = 1; k < 64;) {
        if (r.n < 32) r.fill();
        const int rs = decode_symbol(r, ac);
        if (rs < 0) return false;
        const int run = rs >> 4, size = rs & 15;
        if (size == 0) {
            if (run != 15) break;   // EOB
            k += 16;
            continue;
        }
        k += run;
        if (k > 63) return false;
        zz[k++] = (int16_t)receive_extend(r, size);
    }
    return true;
}

inline unsigned be16(const uint8_t* p) { return ((unsigned)p[0] << 8) | p[1]; }

}  // namespace

namespace {

// Everything in front of the entropy-coded segment of a baseline 4:2:0 file: tables, frame and scan headers.
struct Parsed {
    uint16_t qt[4][64];
    bool have_q[4] = {false, false, false, false};
    DHuff dc[4], ac[4];
    uint8_t dht[2][4][16 + 256];   // the DHT segments as they stood in the file: 16 counts, then the symbols
    int h = 0, w = 0, restart = 0;
    int comp_id[3] = {0, 0, 0}, comp_q[3] = {0, 0, 0};
    int td[3] = {0, 0, 0}, ta[3] = {0, 0, 0};
    bool have_sof = false;
    size_t scan = 0;   // offset of the first entropy-coded byte
};

// 0: `out.scan` is where the scan's bytes start; 1: a kind of file this path does not cover; -1: corrupt.
int parse_until_scan(const uint8_t* data, size_t len, Parsed& P) {
    if (!data || len < 4 || data[0] != 0xFF || data[1] != 0xD8) return -1;
    size_t pos = 2;
    while (pos + 4 <= len) {
        if (data[pos] != 0xFF) return -1;
        const unsigned m = data[pos + 1];
        if (m == 0xFF) {   // fill byte
            ++pos;
            continue;
        }
        pos += 2;
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) return -1;   // EOI before any scan
        if (pos + 2 > len) return -1;
        const size_t seg = be16(data + pos);
        if (seg < 2 || pos + seg > len) return -1;
        const uint8_t* q = data + pos + 2;
        const size_t n = seg - 2;
        if (m == 0xDB) {
            size_t i = 0;
            while (i < n) {
                const int pq = q[i] >> 4, tq = q[i] & 15;
                if (tq > 3) return -1;
                if (pq != 0) return 1;   // 16-bit tables: not baseline
                if (i + 65 > n) return -1;
                for (int k = 0; k < 64; ++k) P.qt[tq][kNatural[k]] = q[i + 1 + k];   // stored row-major
                P.have_q[tq] = true;
                i += 65;
            }
        } else if (m == 0xC4) {
            size_t i = 0;
            while (i < n) {
                if (i + 17 > n) return -1;
                const int tc = q[i] >> 4, th = q[i] & 15;
                if (tc > 1 || th > 3) return -1;
                int total = 0;
                for (int k = 0; k < 16; ++k) total += q[i + 1 + k];
                if (total > 256 || i + 17 + (size_t)total > n) return -1;
                if (!build_decoder(tc ? P.ac[th] : P.dc[th], q + i + 1, q + i + 17, total)) return -1;
                memset(P.dht[tc][th], 0, sizeof(P.dht[tc][th]));
                memcpy(P.dht[tc][th], q + i + 1, 16 + (size_t)total);
                i += 17 + total;
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (n < 15 || q[0] != 8) return 1;
            P.h = (int)be16(q + 1);
            P.w = (int)be16(q + 3);
            if (q[5] != 3) return 1;   // grey or CMYK: libjpeg's business
            for (int c = 0; c < 3; ++c) {
                P.comp_id[c] = q[6 + 3 * c];
                const int hv = q[7 + 3 * c];
                P.comp_q[c] = q[8 + 3 * c];
                if (hv != (c == 0 ? 0x22 : 0x11) || P.comp_q[c] > 3) return 1;   // 4:2:0 only
            }
            if (P.h <= 0 || P.w <= 0 || P.h % 16 || P.w % 16) return 1;
            P.have_sof = true;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            return 1;   // progressive, lossless, arithmetic
        } else if (m == 0xDD) {
            if (n < 2) return -1;
            P.restart = (int)be16(q);
        } else if (m == 0xDA) {
            if (!P.have_sof || n < 10 || q[0] != 3) return P.have_sof ? 1 : -1;
            for (int c = 0; c < 3; ++c) {
                if (q[1 + 2 * c] != P.comp_id[c]) return 1;   // components in another order
                P.td[c] = q[2 + 2 * c] >> 4;
                P.ta[c] = q[2 + 2 * c] & 15;
                if (P.td[c] > 3 || P.ta[c] > 3 || !P.dc[P.td[c]].present || !P.ac[P.ta[c]].present ||
                    !P.have_q[P.comp_q[c]])
                    return -1;
            }
            if (q[7] != 0 || q[8] != 63) return 1;
            for (int k = 0; k < 64; ++k)
                if (P.qt[P.comp_q[2]][k] != P.qt[P.comp_q[1]][k]) return 1;   // Cb and Cr with different tables
            P.scan = pos + seg;
            return 0;
        }
        pos += seg;
    }
    return -1;
}

}  // namespace

extern "C" int lf_jpeg_read_file(const uint8_t* data, size_t len, int16_t* coef, size_t coef_cap, uint16_t* qtab128,
                                 int* h_out, int* w_out) {
    if (!data || !coef || !qtab128 || !h_out || !w_out) return -1;
    Parsed P;
    const int rc = parse_until_scan(data, len, P);
    if (rc != 0) return rc;
    const int h = P.h, w = P.w, restart = P.restart;
    const long mcus = (long)(h / 16) * (w / 16);
    if ((size_t)mcus * 384 > coef_cap) return -1;
    for (int k = 0; k < 64; ++k) {
        qtab128[k] = P.qt[P.comp_q[0]][k];
        qtab128[64 + k] = P.qt[P.comp_q[1]][k];
    }
    Reader r{data + P.scan, data + len};
    int pred[3] = {0, 0, 0};
    int until_restart = restart;
    for (long mcu = 0; mcu < mcus; ++mcu) {
        if (restart && until_restart == 0) {
            if (r.starved()) return -1;   // the interval before this marker ended inside an MCU
            // byte-align, expect RSTn
            const uint8_t* mp = r.p;
            while (mp + 1 < r.end && mp < r.p + 8 && !(mp[0] == 0xFF && mp[1] >= 0xD0 && mp[1] <= 0xD7)) ++mp;
            if (!(mp + 1 < r.end && mp[0] == 0xFF && mp[1] >= 0xD0 && mp[1] <= 0xD7)) return -1;
            r.p = mp + 2;
            r.reset();
            pred[0] = pred[1] = pred[2] = 0;
            until_restart = restart;
        }
        int16_t* b = coef + mcu * 384;
        for (int k = 0; k < 4; ++k)
            if (!decode_block(r, b + 64 * k, pred[0], P.dc[P.td[0]], P.ac[P.ta[0]])) return -1;
        if (!decode_block(r, b + 256, pred[1], P.dc[P.td[1]], P.ac[P.ta[1]])) return -1;
        if (!decode_block(r, b + 320, pred[2], P.dc[P.td[2]], P.ac[P.ta[2]])) return -1;
        if (restart) --until_restart;
    }
    // A file cut inside its scan decodes "successfully" on padding zeros; Pillow raises "image file is
    // truncated" for it (image_utils.py:19-33 then counts the task as failed / skips the file).  Such a
    // file — bits consumed that were never there, or no EOI behind the scan — is handed back (-1) and the
    // caller's libjpeg path gives the reference's verdict.
    if (r.starved() || r.next_marker() != 0xD9) return -1;
    *h_out = h;
    *w_out = w;
    return 0;
}

// The same file prepared for the GPU's Huffman decoder (lf_jpeg_huffman_u8, lf_jpeg_huff.hip) instead of being
// decoded here.  The markers are parsed, the entropy-coded segment is freed of what a one-thread-per-image decoder
// would stumble over — the 0xFF00 stuffing is undone and the RSTn markers are taken out, their places kept as
// offsets — and `slot` receives
//   [0, 256)                 the luminance and chrominance quantisation tables (64 uint16 each, row-major)
//   [256, 256 + 3hw)         left alone: the GPU writes the coefficients there
//   [aux, aux + 32)          aux = lf_jpeg_scan_aux_offset(h, w): magic "LFSC", h, w (uint16), restart interval,
//                            number of restart intervals N (1 without restarts), the 64-bit FNV-1a of the four Huffman
//                            tables, offset of the data from aux, length of the data (uint32 each)
//   [aux + 32, aux + 1120)   the DHT segments of the luminance DC, luminance AC, chrominance DC, chrominance AC
//                            tables: 16 counts + 256 symbols each
//   [aux + 1120, ...)        N + 1 uint32: where each interval's bytes start in the data, and where the last one ends
//   [aux + data offset, ...) the unstuffed bytes of all intervals, then 16 bytes of zeros
// Returns 0; 1 when the file is to be decoded on the host (the kinds lf_jpeg_read_file hands back, Cr tables that
// differ from the Cb tables, a restart-marker count other than the frame calls for, a scan that does not end in
// EOI, a file that does not fit `cap`: lf_jpeg_read_file or libjpeg then gives the verdict); -1 when the markers
// are corrupt.
extern "C" size_t lf_jpeg_scan_aux_offset(int h, int w) { return ((size_t)256 + (size_t)3 * h * w + 15) / 16 * 16; }

extern "C" int lf_jpeg_scan_prepare(const uint8_t* data, size_t len, uint8_t* slot, size_t cap, int* h_out, int* w_out,
                                    uint64_t* hash_out) {
    if (!data || !slot || !h_out || !w_out) return -1;
    Parsed P;
    const int rc = parse_until_scan(data, len, P);
    if (rc != 0) return rc;
    if (P.td[2] != P.td[1] || P.ta[2] != P.ta[1] || P.h > 65535 || P.w > 65535) return 1;
    const long mcus = (long)(P.h / 16) * (P.w / 16);
    const long nint = P.restart ? (mcus + P.restart - 1) / P.restart : 1;
    const size_t aux = lf_jpeg_scan_aux_offset(P.h, P.w), raw_len = len - P.scan;
    const size_t data_off = 1120 + ((size_t)4 * (nint + 1) + 15) / 16 * 16;
    if (raw_len > 0x7FFFFFF0u || aux + data_off + raw_len + 16 > cap) return 1;
    uint8_t* a = slot + aux;
    uint32_t* offs = reinterpret_cast<uint32_t*>(a + 1120);
    uint8_t* out = a + data_off;
    const uint8_t* p = data + P.scan;
    const uint8_t* const end = data + len;
    size_t o = 0;
    long iv = 0;
    offs[0] = 0;
    while (true) {
        const uint8_t* f = p < end ? static_cast<const uint8_t*>(memchr(p, 0xFF, (size_t)(end - p))) : nullptr;
        if (!f) return 1;   // the data ran out without a marker: no EOI
        memcpy(out + o, p, (size_t)(f - p));
        o += (size_t)(f - p);
        if (f + 1 >= end) return 1;
        const unsigned m = f[1];
        if (m == 0x00) {
            out[o++] = 0xFF;
            p = f + 2;
        } else if (P.restart && m >= 0xD0 && m <= 0xD7 && iv + 1 < nint) {
            offs[++iv] = (uint32_t)o;
            p = f + 2;
        } else {   // the entropy-coded segment ends here; what follows must lead to EOI (Reader::next_marker)
            const uint8_t* q = f;
            unsigned next = 0;
            for (; q + 1 < end; ++q)
                if (q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF) {
                    next = q[1];
                    break;
                }
            if (next != 0xD9 || iv + 1 != nint) return 1;
            break;
        }
    }
    offs[nint] = (uint32_t)o;
    memset(out + o, 0, 16);
    uint16_t* qtab128 = reinterpret_cast<uint16_t*>(slot);
    for (int k = 0; k < 64; ++k) {
        qtab128[k] = P.qt[P.comp_q[0]][k];
        qtab128[64 + k] = P.qt[P.comp_q[1]][k];
    }
    uint8_t* t = a + 32;
    memcpy(t, P.dht[0][P.td[0]], 272);
    memcpy(t + 272, P.dht[1][P.ta[0]], 272);
    memcpy(t + 544, P.dht[0][P.td[1]], 272);
    memcpy(t + 816, P.dht[1][P.ta[1]], 272);
    uint64_t hash = 1469598103934665603ull;
    for (int i = 0; i < 1088; ++i) hash = (hash ^ t[i]) * 1099511628211ull;
    const uint32_t magic = 0x4353464Cu;   // "LFSC"
    const uint16_t hw[2] = {(uint16_t)P.h, (uint16_t)P.w};
    const uint32_t rs = (uint32_t)P.restart, ni = (uint32_t)nint, dof = (uint32_t)data_off, dl = (uint32_t)o;
    memcpy(a, &magic, 4);
    memcpy(a + 4, hw, 4);
    memcpy(a + 8, &rs, 4);
    memcpy(a + 12, &ni, 4);
    memcpy(a + 16, &hash, 8);
    memcpy(a + 24, &dof, 4);
    memcpy(a + 28, &dl, 4);
    *h_out = P.h;
    *w_out = P.w;
    if (hash_out) *hash_out = hash;
    return 0;
}

// ---------------------------------------------------------------------------
// The distortion op's noise plane: np.random.RandomState(seed).normal(0, scale, n).astype(np.uint8)
// (srcs/preprocessing/image_augmenter.py:121-123), restated so that a codec worker makes it in a third less time
// than numpy: MT19937 (init_genrand seeding, 53-bit doubles from two draws), numpy's legacy Gaussian — the polar
// method, an accepted pair giving f*x2 first and f*x1 second, f = sqrt(-2 log(r2) / r2) with libm's log and sqrt,
// the very functions numpy's legacy-distributions.c calls — then loc + scale * g and numpy's float64 -> uint8 cast
// (truncate to int32, keep the low byte).  tests/test_jpeg_codec.py compares it with numpy for many seeds.
// ---------------------------------------------------------------------------
namespace {

struct MT {
    uint32_t mt[624];
    int pos;
    explicit MT(uint32_t seed) {
        mt[0] = seed;
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        pos = 624;
    }
    void refill() {
        for (int k = 0; k < 624; ++k) {
            const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
            mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        pos = 0;
    }
    uint32_t next() {
        if (pos >= 624) refill();
        uint32_t y = mt[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    double next_double() {
        const int32_t a = (int32_t)(next() >> 5), b = (int32_t)(next() >> 6);
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
};

}  // namespace

extern "C" int lf_legacy_normal_u8(uint32_t seed, double loc, double scale, size_t n, uint8_t* out8, double* out64) {
    if (!out8 && !out64) return -1;
    MT g(seed);
    size_t i = 0;
    while (i < n) {
        double x1, x2, r2;
        do {
            x1 = 2.0 * g.next_double() - 1.0;
            x2 = 2.0 * g.next_double() - 1.0;
            r2 = x1 * x1 + x2 * x2;
        } while (r2 >= 1.0 || r2 == 0.0);
        const double f = sqrt(-2.0 * log(r2) / r2);
        const double pair[2] = {f * x2, f * x1};   // legacy_gauss returns f*x2 and keeps f*x1 for the next call
        for (int k = 0; k < 2 && i < n; ++k, ++i) {
            const double v = loc + scale * pair[k];
            if (out64) out64[i] = v;
            if (out8) out8[i] = (uint8_t)(int32_t)v;
        }
    }
    return 0;
}
