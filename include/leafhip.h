/*
 * leafhip.h — C ABI of libleafhip.so: the MI355X (gfx950) hot path of leaffliction.
 *
 * The reference (Kiripiro/leaffliction) is pure Python and has no FFI; the
 * boundary each entry point replaces is the third-party library call the
 * reference makes on its hot path.  Each declaration cites
 * the reference call site (path:line under the reference tree) it stands in for.
 *
 * Conventions (last row):
 *  - every buffer is a caller-owned DEVICE pointer (torch tensors on the host
 *    side); the library never allocates, frees or synchronises;
 *  - all work is enqueued on the hipStream_t passed as `stream` (void* here so
 *    that the header needs no HIP include; 0 = the null stream);
 *  - return value: 0 = LF_OK, negative = error (see lf_last_error());
 *    nothing throws or exits across the ABI; the library is re-entrant and
 *    keeps no mutable global state beyond the thread-local error string;
 *  - integer / byte results are bit-exact with the reference's CPU path;
 *    float tolerances are stated per function.
 *
 * Image layouts: "HWC u8" = [N][H][W][3] uint8 interleaved RGB (what
 * np.array(PIL.Image) yields); "NCHW f32" = [N][C][H][W] float.
 */
#ifndef LEAFHIP_H
#define LEAFHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LF_OK 0
#define LF_ERR_INVALID (-1)   /* bad argument (null pointer, non-positive dim, unsupported size) */
#define LF_ERR_LAUNCH (-2)    /* HIP reported a launch error */
#define LF_ERR_WORKSPACE (-3) /* workspace too small */

#define LF_VERSION 100

typedef void* lf_stream_t;

int lf_version(void);
/* Thread-local description of the last error returned on this thread ("" if none). */
const char* lf_last_error(void);

/* The first `width` bytes of `rows` rows between a page-locked host slab and its device mirror (asynchronous on
 * `stream`; to_host: 1 = device -> host, 0 = host -> device): the balancer moves the used front of each fixed-size
 * slot, not the slot (dataset_balancer.py's pixels never left host memory; here they cross PCIe twice). */
int lf_copy_rows(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width, size_t rows, int to_host,
                 lf_stream_t stream);

/* ------------------------------------------------------------------------- */
/* A1 — augmentation / input side (uint8, bit-exact)                          */
/* ------------------------------------------------------------------------- */

/* u8 HWC -> f32 NCHW with exact x/255.0f, optional per-channel (x-mean)/denom.
 * Replaces np.array(img) + ImageTransforms.normalize_array
 * (srcs/utils/image_utils.py:117-130) and, when mean/denom are non-null, the
 * keras Normalization layer (srcs/model/cnn.py:84-86): denom[c] =
 * max(sqrt(var[c]), eps) is computed by the caller; mean3/denom3 are HOST pointers to
 * 3 floats (model constants passed by value into the launch).  Bit-exact vs numpy f32. */
int lf_pack_hwc_u8_to_nchw_f32(const uint8_t* in, float* out, int n, int h, int w,
                               const float* mean3, const float* denom3, lf_stream_t stream);

/* Per-image, per-channel 256-bin histogram: hist[n][c][v] (int32).
 * Replaces PIL Image.histogram() inside ImageOps.autocontrast
 * (srcs/preprocessing/image_augmenter.py:127).  `hist` is overwritten. */
int lf_hist_u8(const uint8_t* in, int32_t* hist, int n, int h, int w, lf_stream_t stream);

/* autocontrast LUT from histograms: lut[n][c][256] u8, cutoff[n] in percent.
 * Restates PIL ImageOps.autocontrast's cut / lo / hi / scale / offset
 * arithmetic in IEEE double (image_augmenter.py:127).  Bit-exact. */
int lf_autocontrast_lut(const int32_t* hist, const double* cutoff, uint8_t* lut, int n,
                        lf_stream_t stream);

/* out[n][y][x][c] = lut[n][c][in[n][y][x][c]]  (PIL Image.point(lut)). */
int lf_lut_apply_u8(const uint8_t* in, const uint8_t* lut, uint8_t* out, int n, int h, int w,
                    lf_stream_t stream);

/* Batch assembly from a device-resident dataset (the loader's cache=True, sequence.py:47-58,
 * kept in HBM instead of host RAM): dst[i][:] = src[index[i]][:] for rows of row_bytes bytes
 * (one resized uint8 image each).  index: int32 device array, values the caller has checked. */
int lf_gather_rows_u8(const uint8_t* src, const int32_t* index, uint8_t* dst, int n_out,
                      size_t row_bytes, lf_stream_t stream);

/* ImageAugmenter.flip (image_augmenter.py:20-31): mode[n] = 0 -> FLIP_LEFT_RIGHT,
 * 1 -> FLIP_TOP_BOTTOM. */
int lf_flip_u8(const uint8_t* in, uint8_t* out, const int32_t* mode, int n, int h, int w,
               lf_stream_t stream);

/* ImageAugmenter.distortion's noise add (image_augmenter.py:121-124):
 * out = (u8)(in + (u8)(int)noise) with uint8 wrap-around, noise in float64
 * exactly as np.random.normal returned it (same shape as the image). */
int lf_noise_wrap_add_u8(const uint8_t* in, const double* noise, uint8_t* out, size_t nbytes,
                         lf_stream_t stream);

/* The same wrap-around add when the noise was already cast to uint8 on the host (numpy's own
 * astype, image_augmenter.py:121-123; the codec worker processes of DatasetBalancer do that next
 * to the JPEG decode): out = in + add mod 256, bytewise.  nbytes % 4 == 0, 4-byte aligned. */
int lf_add_wrap_u8(const uint8_t* in, const uint8_t* add, uint8_t* out, size_t nbytes,
                   lf_stream_t stream);

/* Same op with the noise drawn on the device: counter-based Philox4x32-10 +
 * Box-Muller N(0, sigma) keyed by (seed, byte index).  Statistically, not
 * bit-wise, equal to the numpy stream; used for the synthetic C3 pass. */
int lf_noise_philox_add_u8(const uint8_t* in, uint8_t* out, size_t nbytes, uint64_t seed,
                           float sigma, lf_stream_t stream);

/* ImageAugmenter.distortion (image_augmenter.py:121-131), first half in one pass: out = in + noise
 * (mod 256) TOGETHER WITH the per-image, per-channel histogram of `out` that ImageOps.autocontrast
 * takes next (hist [N][3][256] int32, as lf_hist_u8 gives it) — the noisy image is not read back
 * just to be counted.  add != NULL: the uint8 noise plane of lf_add_wrap_u8; add == NULL: the
 * Philox noise of lf_noise_philox_add_u8 (same seed -> same bytes).  h*w*3 % 16 == 0, 16-byte
 * aligned buffers. */
int lf_noise_hist_u8(const uint8_t* in, const uint8_t* add, uint8_t* out, int32_t* hist, int n, int h,
                     int w, uint64_t seed, float sigma, lf_stream_t stream);

/* apply_mask (srcs/utils/mask_utils.py:67-79) and blur.py:74-75:
 * out = mask > 127 ? img : color (color 0 or 255), mask is [N][H][W] u8. */
int lf_mask_composite_u8(const uint8_t* img, const uint8_t* mask, uint8_t* out, int n, int h,
                         int w, int color, lf_stream_t stream);

/* cv2.cvtColor(rgb, COLOR_RGB2HSV) on uint8 (hist.py:184, blur.py:44):
 * OpenCV's 8-bit fixed-point path, H in [0,180).  Parity unpinned (no cv2). */
int lf_rgb2hsv_u8(const uint8_t* rgb, uint8_t* hsv, size_t npixels, lf_stream_t stream);

/* cv2.cvtColor(rgb, COLOR_RGB2GRAY) on uint8 (blur.py:27):
 * (4899 R + 9617 G + 1868 B + 8192) >> 14. */
int lf_rgb2gray_u8(const uint8_t* rgb, uint8_t* gray, size_t npixels, lf_stream_t stream);

/* cv2.GaussianBlur(img, (k,k), sigma) on uint8 with BORDER_REFLECT_101
 * (blur.py:61,72): separable, both passes fused through LDS, OpenCV's 8.8
 * fixed-point kernel (kq: HOST array of ksize uint16 taps, sum 256).  channels = 1 or 3,
 * ksize odd <= 31; odd ksize 3..15 with every tap <= 255 takes the dot-product fast path. */
int lf_gauss_blur_u8(const uint8_t* in, uint8_t* out, int n, int h, int w, int channels,
                     const uint16_t* kq, int ksize, lf_stream_t stream);

/* HSV colour-region statistics of apply_histogram_filter (hist.py:22-67,
 * 181-189, 248-256): per image, with leaf = (s>10)&(v>15)&(v<245):
 * counts[n][0] = leaf pixels, [1..8] = the 8 predicate counts,
 * [9..13] = the 5 hue-range counts; hsv_hist[n][3][256] = histograms of H,S,V
 * over leaf pixels.  Input is RGB HWC u8 (the HSV conversion is fused). int64-free:
 * all counters int32. */
#define LF_HSV_NCOUNTS 14
int lf_hsv_region_stats(const uint8_t* rgb, int32_t* counts, int32_t* hsv_hist, int n, int h,
                        int w, lf_stream_t stream);

/* apply_blur_filter (srcs/transform/filters/blur.py:18-79) given the leaf mask its
 * make_mask_func returned (leaf = mask > 0): gray -> Canny(50,150,L2) dilated by the 3x3
 * MORPH_ELLIPSE element (x0.4) + uint8(min-max normalised Sobel magnitude) (x0.3) + brown
 * regions (hue_lo <= H <= hue_hi, S >= s_min, V <= v_max, inside the leaf; closed, then dilated
 * twice) (x0.6, skipped when use_brown = 0) + min-max normalised mean |rgb - GaussianBlur15|
 * (x0.2) -> min-max normalised to uint8 -> GaussianBlur 5x5 -> kept under the leaf mask,
 * replicated to RGB.  rgb/out [n][h][w][3], leaf_mask [n][h][w]; kq15 / kq5: HOST arrays of
 * the 8.8 fixed-point Gaussian taps (15 taps for sigma 0 -> 2.6, 5 taps for cfg.gaussian_sigma).
 * Parity unpinned (no cv2); every step follows oracle/cv_ops.py:blur_saliency bit for bit. */
size_t lf_blur_saliency_workspace(int n, int h, int w);
int lf_blur_saliency_u8(const uint8_t* rgb, const uint8_t* leaf_mask, uint8_t* out, int n, int h,
                        int w, int use_brown, int hue_lo, int hue_hi, int s_min, int v_max,
                        const uint16_t* kq15, const uint16_t* kq5, void* workspace,
                        size_t ws_bytes, lf_stream_t stream);

/* _create_inclusive_mask (srcs/transform/filters/mask.py:727-831), the default strategy of make_mask
 * (mask.py:548-582, config.yaml:6 "inclusive"), on the working image: colour predicates in 8-bit HSV and
 * L*a*b* (mask.py:735-770), gray / purple / untextured background removal (:772-789), Canny(30, 100)
 * edges dilated 3x3 (:791-794), open 3x3 / close 9x9 / close 7x7 with cv2's MORPH_ELLIPSE elements
 * (:806-815), largest 8-connected component (:817-824), close 5x5 (:826-829).
 * rgb [N,H,W,3] -> mask [N,H,W] 0 / 255.  green_lo / green_hi: cfg.green_hue_range (config.yaml:10);
 * kq15: HOST pointer to the 15 Q8.8 taps of GaussianBlur(gray, (15, 15), 0).
 * The upscale before it (_prepare_working_image, mask.py:29-50) and GrabCut / brown extension after it
 * (:307-392) are not part of this call.  Parity unpinned (no cv2); follows oracle/cv_ops.py:inclusive_mask. */
size_t lf_inclusive_mask_workspace(int n, int h, int w);
int lf_inclusive_mask_u8(const uint8_t* rgb, uint8_t* mask, int n, int h, int w, int green_lo,
                         int green_hi, const uint16_t* kq15, void* workspace, size_t ws_bytes,
                         lf_stream_t stream);

/* ------------------------------------------------------------------------- */
/* JPEG encode (the file Pillow's Image.save(path, quality=q) writes)          */
/* ------------------------------------------------------------------------- */
/* Replaces ImageLoader.save_pil_image / save_array (srcs/utils/image_utils.py:49-56; the balancer's output
 * step, dataset_balancer.py:201-207) for images of any size: baseline, 4:2:0, Annex K Huffman
 * tables, JFIF 1.01 — libjpeg-turbo's integer pipeline restated (jccolor, jcsample h2v2, jfdctint,
 * jcdctmgr, jchuff, jcmarker), so the bytes equal Pillow's.
 * lf_jpeg_fdct_quant_u8 (GPU): rgb [N,H,W,3] -> coef int16 [N][ceil(H/16) * ceil(W/16) MCUs][Y00 Y01 Y10 Y11 Cb Cr][64],
 *   quantised, in zigzag order (768 bytes per MCU, 16-byte aligned).  Sizes that are not whole MCUs are padded as
 *   libjpeg pads them (replicated edges, dummy blocks: jcprepct.c, jcsample.c, jccoefct.c).
 * lf_jpeg_write_file (HOST, no GPU call in it; also exported by libleafcodec.so for the codec worker
 *   processes): one image's coefficients -> the complete file in `out`; returns its length, -1 on bad
 *   arguments or if `cap` (take lf_jpeg_file_bound) is too small.
 * lf_jpeg_quant_tables (HOST): jpeg_set_quality(quality, force_baseline) tables, row-major. */
int lf_jpeg_fdct_quant_u8(const uint8_t* rgb, int16_t* coef, int n, int h, int w, int quality,
                          lf_stream_t stream);
void lf_jpeg_quant_tables(int quality, uint8_t* lum64, uint8_t* chroma64);
size_t lf_jpeg_file_bound(int h, int w);
long lf_jpeg_write_file(const int16_t* coef, int h, int w, int quality, uint8_t* out, size_t cap);
/* The entropy coding on the GPU as well: lf_jpeg_entropy_u8 turns the coefficients of N images (image i at
 * coef + i*coef_stride bytes, the layout above) into their Huffman-coded, byte-stuffed scans: row i of `out`
 * (out_stride bytes) = int32 length, then the bytes; length -1 when a scan does not fit its row.
 * lf_jpeg_wrap_scan (HOST, also in libleafcodec.so) puts the markers around such a scan: the complete file. */
size_t lf_jpeg_entropy_workspace(int n, size_t out_stride);
int lf_jpeg_entropy_u8(const void* coef, size_t coef_stride, uint8_t* out, size_t out_stride, int n, int h,
                       int w, void* workspace, size_t ws_bytes, lf_stream_t stream);
long lf_jpeg_wrap_scan(const uint8_t* scan, size_t scan_len, int h, int w, int quality, uint8_t* out, size_t cap);
/* The same two GPU steps for images of DIFFERENT sizes in one launch each (the balancer's rotated canvases: a rotation
 * with expand=True gives every output its own size; dataset_balancer.py:201-207 saves each with Image.save):
 * items[i] (device memory) = where image i's pixels start in rgb_base (bytes; any alignment), where its coefficients
 * go in coef_base (int16 elements, a multiple of 8), where its scan goes in out_base (bytes, a multiple of 4; the same
 * place as its pixels is fine: the coefficients are complete before the scan is written), the number of the first of
 * its lf_jpeg_fdct_groups(h, w) passes (running sum over the images before it), its size, and 6 x its MCUs.
 * out_room: the bytes each scan may take (int32 length first, -1 when it did not fit), as out_stride above. */
typedef struct {
    int64_t rgb_off, coef_off, out_off, group_start;
    int32_t h, w, nblocks, reserved;
} lf_jpeg_item;
long lf_jpeg_fdct_groups(int h, int w);
int lf_jpeg_fdct_quant_items_u8(const uint8_t* rgb_base, int16_t* coef_base, const lf_jpeg_item* items, int n,
                                long total_groups, int quality, lf_stream_t stream);
int lf_jpeg_entropy_items_u8(const void* coef_base, const lf_jpeg_item* items, uint8_t* out_base, size_t out_room, int n,
                             void* workspace, size_t ws_bytes, lf_stream_t stream);
/* Decoding, the same split the other way round (Image.open(path).convert("RGB"), image_utils.py:19-33 — the
 * balancer's input step and the loader's):
 * lf_jpeg_read_file (HOST, also in libleafcodec.so): markers + Huffman decoding of a baseline 4:2:0 file of
 *   whole MCUs -> coef (the layout above, still quantised; coef_cap in int16 elements) and qtab128 (the file's
 *   luminance and chrominance tables, 64 uint16 each, row-major).  Returns 0; 1 when the file is of a kind it
 *   does not cover (progressive, other samplings, grey, ragged size: decode with libjpeg); -1 when it is corrupt.
 * lf_jpeg_idct_rgb_u8 (GPU): dequantisation + jidctint islow IDCT + h2v2 fancy upsampling + YCbCr->RGB for N
 *   images of one size; image i's coefficients at coef + i*coef_stride bytes, its tables at qtab + i*qtab_stride. */
int lf_jpeg_read_file(const uint8_t* data, size_t len, int16_t* coef, size_t coef_cap, uint16_t* qtab128,
                      int* h, int* w);
/* The same decoding with the Huffman step on the GPU as well (jdhuff.c's decode_mcu; the balancer's input step):
 * lf_jpeg_scan_prepare (HOST, also in libleafcodec.so): markers only.  The slot receives the quantisation tables
 *   at [0, 256), and at lf_jpeg_scan_aux_offset(h, w) = align16(256 + 3hw) a 32-byte header, the four Huffman
 *   tables as they stood in the file, the offsets of the restart intervals and the entropy-coded bytes with the
 *   0xFF00 stuffing undone and the RSTn markers taken out (layout: lf_jpeg_host.cpp).  *hash = FNV-1a of the
 *   Huffman tables: images decoded in one launch must share it.  Returns 0; 1 = decode this file on the host
 *   (lf_jpeg_read_file / libjpeg); -1 = corrupt markers.
 * lf_jpeg_huffman_u8 (GPU): N prepared slots of one size -> each slot's coefficient area [256, 256 + 3hw) in
 *   lf_jpeg_read_file's layout (then lf_jpeg_idct_rgb_u8 as before).  mode 0: one workgroup per image decodes 256
 *   subsequences of the scan at once (self-synchronising; lf_jpeg_huff.hip), and the images that kernel does not
 *   take (scans of a megabyte and more) go through the one-lane-per-image kernel; mode 1: that kernel for all.  (A scan
 *   up to 96 KB is staged in LDS, a longer one is read where it lies; restart intervals are decoded one per thread.)
 *   status[i] (int32, device): 0 decoded; 1 the scan is malformed or ends early (what lf_jpeg_read_file answers
 *   with -1: give the file to libjpeg for the reference's verdict); 2 (one-lane-per-image kernel only: it shares
 *   one set of tables among 64 images) the image's hash differs from that of the first image of its group; 3 no
 *   prepared scan in the slot. */
size_t lf_jpeg_scan_aux_offset(int h, int w);
int lf_jpeg_scan_prepare(const uint8_t* data, size_t len, uint8_t* slot, size_t cap, int* h, int* w, uint64_t* hash);
int lf_jpeg_huffman_u8(void* slots, size_t stride, int n, int h, int w, int* status, int mode, lf_stream_t stream);
size_t lf_jpeg_decode_workspace(int n, int h, int w);
int lf_jpeg_idct_rgb_u8(const void* coef, size_t coef_stride, const void* qtab, size_t qtab_stride,
                        uint8_t* rgb, int n, int h, int w, void* workspace, size_t ws_bytes,
                        lf_stream_t stream);

/* HOST (also in libleafcodec.so): np.random.RandomState(seed).normal(loc, scale, n) — the distortion op's noise
 * plane (srcs/preprocessing/image_augmenter.py:121-123) — from MT19937 and numpy's legacy polar Gaussian with
 * libm's log / sqrt: out64 (optional) the float64 values, bit for bit; out8 (optional) their numpy astype(uint8). */
int lf_legacy_normal_u8(uint32_t seed, double loc, double scale, size_t n, uint8_t* out8, double* out64);
/* The same planes for N seeds at once on the GPU (lf_noise.hip; one workgroup per plane): out + i*out_stride receives
 * RandomState(seeds[i]).normal(loc, scale, count).astype(uint8).  flags[i] (int32, device): 0 = the plane is numpy's
 * byte for byte; 1 = some value lay within 1e-9 of an integer, where the last bit of log() decides the cast — make
 * that plane with lf_legacy_normal_u8; 2 = ran out of attempts (cannot happen for count >= 16; same remedy). */
int lf_legacy_normal_batch_u8(const uint32_t* seeds, double loc, double scale, size_t count, uint8_t* out,
                              size_t out_stride, int n, int* flags, lf_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Geometric ops (Pillow semantics, bit-exact; coordinates in IEEE double)    */
/* ------------------------------------------------------------------------- */

/* Image.transform(size, AFFINE|PERSPECTIVE, coeffs, BICUBIC) (image_augmenter.py:44-94).
 * coeffs[n][8] double (affine uses the first 6, a6 = a7 = 0), output same size,
 * outside pixels black.  `perspective` bit 0: PERSPECTIVE (vs AFFINE) map; bit 1: hint that
 * the maps are axis-aligned scales (a1 = a3 = 0, e.g. ImageAugmenter.skew) — selects a kernel
 * that reuses horizontally interpolated rows down a column; the hint is verified per image
 * and never changes results. */
int lf_warp_bicubic_u8(const uint8_t* in, uint8_t* out, const double* coeffs, int perspective,
                       int n, int h, int w, lf_stream_t stream);

/* Image.rotate(angle, expand=True, fillcolor="white") NEAREST path
 * (image_augmenter.py:37): Pillow's 16.16 fixed-point affine.  fix6[n][6] are the
 * int32 fixed-point coefficients a0,a1,a2',a3,a4,a5' computed on the host exactly as
 * Pillow's affine_fixed does.  in: [n][h][w][3].  The output is a ragged batch (the
 * expanded canvas depends on the angle): image i writes ohw[i] = (oh, ow) pixels,
 * packed RGB, at byte offset out_off[i] of `out`; pixels whose source falls outside
 * get `fill`.  max_out_pixels = max_i oh*ow sizes the grid. */
int lf_affine_nearest_fixed_u8(const uint8_t* in, uint8_t* out, const int32_t* fix6,
                               const int32_t* ohw, const int64_t* out_off, int n, int h, int w,
                               int max_out_pixels, int fill, lf_stream_t stream);

/* Pillow two-pass separable resample with 8-bit intermediates and 22-bit
 * fixed-point coefficients (Image.resize(..., LANCZOS), image_utils.py:109-114,
 * image_augmenter.py:110).  bounds/coefficients are computed on the host exactly as
 * Resample.c's precompute_coeffs + normalize_coeffs_8bpc do; a crop box is folded into
 * them (xmin/ymin index the uncropped image).
 * Horizontal pass: in [n][h][w][3] -> tmp [n][h][ow][3]; vertical: -> out [n][oh][ow][3].
 * xbounds [ow][2] = (xmin, count), xk [ow][kx]; ybounds [oh][2], yk [oh][ky] (int32);
 * with per_image_coeffs != 0 each table has a leading [n] dimension.  Coefficients must
 * satisfy |k| < 2^23 (normalised 22-bit weights always do): the kernels multiply with
 * full-rate 24-bit operands. */
int lf_resample_u8(const uint8_t* in, uint8_t* tmp, uint8_t* out, int n, int h, int w, int oh,
                   int ow, const int32_t* xbounds, const int32_t* xk, int kx,
                   const int32_t* ybounds, const int32_t* yk, int ky, int per_image_coeffs,
                   lf_stream_t stream);

/* The same resample with both passes fused in one kernel (32x32 output tiles, the tile's input
 * window and the 8-bit intermediate kept in LDS; no tmp buffer).  Preconditions: kx, ky <= 10,
 * ow % 4 == 0, and the windows of any 32 consecutive outputs span at most 48 inputs on either
 * axis (crop -> resize back, scales up to ~1.25); the host checks the last one on its tables
 * before choosing this entry (ops.crop_resize_plan).  Bit-identical to lf_resample_u8. */
int lf_resample_tile_u8(const uint8_t* in, uint8_t* out, int n, int h, int w, int oh, int ow,
                        const int32_t* xbounds, const int32_t* xk, int kx, const int32_t* ybounds,
                        const int32_t* yk, int ky, int per_image_coeffs, lf_stream_t stream);

/* ------------------------------------------------------------------------- */
/* A2 — leaf_cnn conv stack (fp32, NCHW activations)                           */
/* ------------------------------------------------------------------------- */
/* Conv weights are kept in "IKO" layout [Cin][k*k][Cout] (tap = ky*k + kx); the keras
 * HWIO kernel [k][k][Cin][Cout] maps to it by a transpose at the artifact boundary. */

/* keras Conv2D(cout, ksize, padding="same", use_bias=False) forward
 * (srcs/model/cnn.py:27-29,44): y[n][co] = sum_{ci,tap} w[ci][tap][co] * x[n][ci] (cross-
 * correlation, zero padding).  Optional fused input prologue (both pointers non-null):
 * x' = x*in_scale[ci] + in_shift[ci], then relu if in_relu — the producer's
 * BatchNorm(+ReLU) applied while staging (cnn.py:30-31); padding stays exactly zero.
 * Implicit GEMM on v_mfma_f32_32x32x2_f32: result equals an fp32 fmaf chain over
 * k = (ci, tap) in ascending order.  Tolerance vs fp32 torch conv2d: 1e-4 relative.
 * dgrad is the same call on dy with lf_conv2d_dgrad_weights_f32's output; accumulate != 0
 * adds into y (residual gradient joins) instead of overwriting it. */
int lf_conv2d_f32(const float* x, const float* w, float* y, int n, int cin, int h, int wd, int cout,
                  int ksize, const float* in_scale, const float* in_shift, int in_relu,
                  int accumulate, lf_stream_t stream);

/* The same forward convolution with both operands rounded to bf16 (round to nearest even) while
 * staging and fp32 accumulation on v_mfma_f32_32x32x16_bf16 — the reduced-precision inference mode
 * (the reference predicts under Keras' mixed_float16 policy by default, train.py:53-117;
 * BASELINE configs[4]).  x and y stay fp32 NCHW, so every other kernel of the forward pass is
 * shared with the fp32 path.  Weights are packed once per model with
 * lf_conv2d_bf16_prep_weights: fp32 IKO [cin][k*k][cout] -> bf16 [ceil(cin/16)][k*k][cout][16]
 * (lf_conv2d_bf16_weight_elems uint16 elements, channels past cin zero).  Requires w % 4 == 0,
 * cout % 32 == 0, ksize 1 or 3.  Tolerance vs lf_conv2d_f32: bf16 operand rounding, 2^-8
 * relative per product (tests bound the error by 2e-2 of the output's scale). */
size_t lf_conv2d_bf16_weight_elems(int cin, int cout, int ksize);
int lf_conv2d_bf16_prep_weights(const float* w_iko, uint16_t* wprep, int cin, int cout, int ksize,
                                lf_stream_t stream);
int lf_conv2d_bf16_f32(const float* x, const uint16_t* wprep, float* y, int n, int cin, int h, int wd,
                       int cout, int ksize, const float* in_scale, const float* in_shift, int in_relu,
                       lf_stream_t stream);

/* The reduced-precision forward pass with bf16 ACTIVATION STORAGE as well (what mixed_float16
 * keeps between layers): the same convolution reading and / or writing bf16 NCHW tensors
 * (x_bf16 / y_bf16 flags; the fused prologue and the accumulation stay fp32), and the two plane
 * kernels of the block on bf16 tensors — lf_gap_f32's mean of relu?(x*scale+shift) (no mask
 * sums) and lf_block_tail_fwd_f32's maxpool2x2(relu(shortcut' + relu(BN(y)) * gate)) without the
 * route bytes and dropout a backward pass would need.  hw % 4 == 0 / w % 4 == 0, h even.
 * out_scale / out_shift / out_relu: optional epilogue v*out_scale[co]+out_shift[co] (+ReLU) on the
 * fp32 accumulators — at inference the layer's folded BatchNorm(+ReLU), so that what is stored is
 * the activation itself and the consumer needs no prologue (a bf16 input without prologue is staged
 * by interleaving the stored bits, no arithmetic).  The tail's a_scale / a_shift may then be null
 * (y holds relu(BN(.)) already). */
int lf_conv2d_bf16_act(const void* x, int x_bf16, const uint16_t* wprep, void* y, int y_bf16, int n,
                       int cin, int h, int wd, int cout, int ksize, const float* in_scale,
                       const float* in_shift, int in_relu, const float* out_scale,
                       const float* out_shift, int out_relu, lf_stream_t stream);
/* The block's second convolution at inference TOGETHER WITH the squeeze of its SE gate (cnn.py:33-41:
 * GlobalAveragePooling2D over relu(BN(conv2))): stores the bf16 activation like lf_conv2d_bf16_act and
 * leaves means[n][co] = mean over the plane of the STORED (rounded) values — summed in the
 * convolution's epilogue, per image, so the activation is not read back from memory for the pool
 * (lf_gap_bf16 remains for everything else).  Workspace: lf_conv2d_bf16_act_mean_workspace bytes. */
size_t lf_conv2d_bf16_act_mean_workspace(int n, int cin, int h, int wd, int cout, int ksize, int x_bf16);
int lf_conv2d_bf16_act_mean(const void* x, int x_bf16, const uint16_t* wprep, uint16_t* y, int n, int cin,
                            int h, int wd, int cout, int ksize, const float* in_scale,
                            const float* in_shift, int in_relu, const float* out_scale,
                            const float* out_shift, int out_relu, float* means, void* workspace,
                            size_t ws_bytes, lf_stream_t stream);
int lf_gap_bf16(const uint16_t* x, float* out, int n, int c, int hw, const float* scale,
                const float* shift, int relu, lf_stream_t stream);
int lf_block_tail_fwd_bf16(const uint16_t* y, const float* a_scale, const float* a_shift,
                           const float* s, const uint16_t* sc, const float* sc_scale,
                           const float* sc_shift, int sc_relu, uint16_t* pooled, int n, int c, int h,
                           int w, lf_stream_t stream);

/* ------------------------------------------------------------------------- */
/* A2 — the mixed-precision TRAINING step (bf16 storage, fp32 arithmetic)      */
/* ------------------------------------------------------------------------- */
/* The reference trains under keras.mixed_precision.set_global_policy("mixed_float16") unless
 * --no-mixed-precision is given (srcs/cli/train.py:179-190): layer outputs and gradients are
 * 16-bit, variables, BatchNorm statistics, softmax / loss and the optimizer are fp32.  BASELINE
 * configs[3] asks for that step in bf16.  Here: every activation / gradient tensor in HBM is bf16
 * NCHW, every MFMA operand is bf16 (v_mfma_f32_32x32x16_bf16, fp32 accumulators), all other
 * arithmetic is fp32, master weights / Adam state / BatchNorm state are fp32.  A value is rounded
 * (nearest even) exactly where it is stored or staged as an operand, and every statistic a later
 * kernel relies on is taken over the ROUNDED values.  Tolerance vs the fp32 step: bf16 operand
 * rounding (2^-9 relative per element); vs oracle/cnn_ref.py evaluated with the same rounding
 * points the tests bound each gradient tensor's error norm (tests/test_train_bf16_gpu.py).
 *
 * lf_conv2d_bf16_train: forward convolution (Conv2D, cnn.py:27-29) or input-gradient convolution
 * (the same call on dY with lf_conv2d_dgrad_weights_f32 -> lf_conv2d_bf16_prep_weights) writing
 * bf16; x is fp32 (x_bf16 = 0: the normalised network input) or bf16; optional prologue
 * relu?(x*in_scale+in_shift) = the producer's BatchNorm(+ReLU) (cnn.py:30-31); accumulate != 0:
 * y = bf16(conv + y) (residual gradient joins).  tile_part (may be null) receives per-(channel,
 * part) sums, part < lf_conv2d_bf16_stats_tiles(...) — one per tile, or one per workgroup on the
 * streaming path that serves Cin, Cout <= 64 —, layout [cout][parts][2]:
 *   mask_y == null: BatchNormalization forward statistics {sum (y-pivot), sum (y-pivot)^2}
 *     (feed lf_bn_train_stats_tiles_f32; pivot = the moving mean, may be null);
 *   mask_y != null: the backward sums of the BatchNorm this gradient feeds, {sum d, sum d*mask_y}
 *     with d = y*[mask_y*mask_scale+mask_shift > 0 or !mask_relu] (feed lf_bn_bwd_sums_tiles_f32).
 * Requires w % 4 == 0, cout % 32 == 0, ksize 1 or 3, 16-byte aligned x / wprep. */
long long lf_conv2d_bf16_stats_tiles(int n, int cin, int h, int w, int cout, int ksize, int x_bf16);
int lf_conv2d_bf16_train(const void* x, int x_bf16, const uint16_t* wprep, uint16_t* y, int n, int cin,
                         int h, int w, int cout, int ksize, const float* in_scale,
                         const float* in_shift, int in_relu, int accumulate, float* tile_part,
                         size_t tile_part_bytes, const float* pivot, const uint16_t* mask_y,
                         const float* mask_scale, const float* mask_shift, int mask_relu,
                         lf_stream_t stream);

/* Conv2D weight gradient dw[cin][k*k][cout] (fp32, overwritten) from bf16 tensors, K = pixels on
 * the bf16 MFMA with fp32 partial slabs summed in a fixed order (deterministic).  x: the conv's
 * input as stored (bf16; fp32 when cin*9 <= 32, the stem) with the optional prologue
 * relu?(x*in_scale+in_shift); g: dY itself, or — with bn_y — the gradient w.r.t. the output of the
 * BatchNormalization(+ReLU) that follows the conv, in which case the BatchNorm backward is formed
 * while staging exactly as lf_conv2d_wgrad_bn_f32 does (alpha_nc / add_nc [n][cout] optional,
 * coef [5][cout] from lf_bn_bwd_sums*_f32) and dY is also written to dy_out (bf16, may be null)
 * for the input-gradient convolution.  Requires w % 4 == 0, cout % 32 == 0, cin % 4 == 0 (or the
 * stem), 8-byte aligned tensors.  workspace >= lf_conv2d_wgrad_bf16_workspace(...) bytes. */
size_t lf_conv2d_wgrad_bf16_workspace(int n, int cin, int h, int w, int cout, int ksize);
int lf_conv2d_wgrad_bf16(const void* x, const uint16_t* g, const uint16_t* bn_y, const float* alpha_nc,
                         const float* add_nc, const float* coef, int bn_relu, uint16_t* dy_out,
                         float* dw, int n, int cin, int h, int w, int cout, int ksize,
                         const float* in_scale, const float* in_shift, int in_relu, void* workspace,
                         size_t ws_bytes, lf_stream_t stream);

/* The plane kernels of the training step on bf16 tensors — the arithmetic of lf_gap_f32 (with
 * mask_sums), lf_block_tail_fwd_f32 (route bytes, SpatialDropout2D keep-scales),
 * lf_block_tail_bwd_f32 and lf_bcast_planes_f32 (cnn.py:35-49,94-101), fp32 after widening;
 * pooled / dr / out are rounded to bf16 where stored and the per-plane sums are over the rounded
 * gradient.  hw % 4 == 0, w % 4 == 0, h even. */
int lf_gap_stats_bf16(const uint16_t* x, float* out, float* mask_sums, int n, int c, int hw,
                      const float* scale, const float* shift, int relu, lf_stream_t stream);
int lf_block_tail_fwd_train_bf16(const uint16_t* y, const float* a_scale, const float* a_shift,
                                 const float* s, const uint16_t* sc, const float* sc_scale,
                                 const float* sc_shift, int sc_relu, const float* drop, uint8_t* route,
                                 uint16_t* pooled, int n, int c, int h, int w, lf_stream_t stream);
int lf_block_tail_bwd_bf16(const uint16_t* dp, const uint8_t* route, const uint16_t* y,
                           const float* a_scale, const float* a_shift, const float* drop, uint16_t* dr,
                           float* ds, float* plane_sums, const uint16_t* sc_y, float* sc_sums, int n,
                           int c, int h, int w, lf_stream_t stream);
int lf_bcast_planes_bf16(const float* v, uint16_t* out, int planes, int hw, float scale,
                         lf_stream_t stream);
/* fp32 <-> bf16 (round to nearest even) of a flat buffer: the data-parallel gradient bucket
 * crosses xGMI as bf16 (2.5 MB instead of 5 MB). */
int lf_cast_f32_bf16(const float* in, uint16_t* out, size_t count, lf_stream_t stream);
int lf_cast_bf16_f32(const uint16_t* in, float* out, size_t count, lf_stream_t stream);

/* Which tile variant (template instantiation) the dispatcher picks for a shape — used by
 * bench.py to attribute measured launch durations to kernel names.  (For H = 28 the 28x8
 * variant walks two images as one strip; lf_conv2d_stats_tiles accounts for that.) */
int lf_conv2d_variant(int h, int wd, int cout, int ksize);
int lf_conv2d_wgrad_variant(int n, int cin, int h, int wd, int cout, int ksize);

/* Conv2D followed by BatchNormalization in training mode (cnn.py:28-33,40-45): the same
 * convolution, and in its epilogue the per-tile sums of (y - pivot[co]) and (y - pivot[co])^2 for
 * every output channel -> tile_part[co][tile][2] (tile < lf_conv2d_stats_tiles(...)).  pivot
 * (device, [cout], may be null) only conditions the sum of squares; pass the layer's moving
 * mean.  lf_bn_train_stats_tiles_f32 (below) turns the tile sums into the batch statistics, so
 * the activation is not read again. */
long long lf_conv2d_stats_tiles(int n, int cin, int h, int w, int cout, int ksize);
int lf_conv2d_stats_f32(const float* x, const float* w, float* y, int n, int cin, int h, int wd,
                        int cout, int ksize, const float* in_scale, const float* in_shift,
                        int in_relu, const float* pivot, float* tile_part, size_t tile_part_bytes,
                        lf_stream_t stream);

/* Input-gradient convolution whose output g feeds a BatchNormalization backward (mask_y = that
 * BN's input, same shape as y): besides y (= conv, or y += conv with accumulate) the epilogue
 * leaves per-tile {sum d, sum d*mask_y}, d = y*[mask_y*mask_scale[co]+mask_shift[co] > 0 or
 * !mask_relu], in tile_part[co][tile][2] for lf_bn_bwd_sums_tiles_f32 — the BN backward's
 * reduction pass over g and its input disappears. */
int lf_conv2d_bnbwd_f32(const float* x, const float* w, float* y, int n, int cin, int h, int wd,
                        int cout, int ksize, int accumulate, const float* mask_y,
                        const float* mask_scale, const float* mask_shift, int mask_relu,
                        float* tile_part, size_t tile_part_bytes, lf_stream_t stream);

/* w [Cin][k*k][Cout] -> wt [Cout][k*k (flipped)][Cin]: the weights with which
 * lf_conv2d_f32(dy, wt, dx, n, cout, h, w, cin, k, ...) is the input gradient. */
int lf_conv2d_dgrad_weights_f32(const float* w, float* wt, int cin, int ksize, int cout,
                                lf_stream_t stream);

/* Weight gradient dw[ci][tap][co] = sum_{n,y,x} x'[n][ci][y+ky-1][x+kx-1] * dy[n][co][y][x]
 * (x' = optional prologue as above) in two steps: lf_conv2d_wgrad_f32 writes one partial
 * slab per workgroup into a workspace of lf_conv2d_wgrad_workspace(...) bytes;
 * lf_conv2d_wgrad_reduce_f32 sums them in a fixed order (two-stage when there are many):
 * dw = beta*dw + sum (deterministic, no float atomics; the workspace tail is scratch). */
size_t lf_conv2d_wgrad_workspace(int n, int cin, int h, int wd, int cout, int ksize);
int lf_conv2d_wgrad_f32(const float* x, const float* dy, int n, int cin, int h, int wd, int cout,
                        int ksize, const float* in_scale, const float* in_shift, int in_relu,
                        void* workspace, size_t ws_bytes, lf_stream_t stream);
int lf_conv2d_wgrad_reduce_f32(void* workspace, float* dw, int n, int cin, int h, int wd,
                               int cout, int ksize, float beta, lf_stream_t stream);
/* Weight gradient whose dY operand is a BatchNormalization backward, formed on the fly:
 * dY = coef2*dz + coef3*bn_y + coef4, dz = (g*alpha_nc+add_nc)*[bn_y*coef0+coef1 > 0 or !bn_relu]
 * (coef from lf_bn_bwd_sums_f32), also written to dy_out [n][cout][h][w] (may be null: the stem
 * has no input gradient) for the input-gradient convolution that follows.  3x3 (the small-Cin
 * stem kernel included) and 1x1, shapes for which lf_conv2d_wgrad_bn_supported() != 0; same
 * workspace and reduce step as lf_conv2d_wgrad_f32. */
int lf_conv2d_wgrad_bn_supported(int n, int cin, int h, int w, int cout, int ksize);
int lf_conv2d_wgrad_bn_f32(const float* x, const float* g, const float* bn_y,
                           const float* alpha_nc, const float* add_nc, const float* coef,
                           int bn_relu, float* dy_out, int n, int cin, int h, int w, int cout,
                           int ksize, const float* in_scale, const float* in_shift, int in_relu,
                           void* workspace, size_t ws_bytes, lf_stream_t stream);

/* ---- input stage ----------------------------------------------------------- */
/* u8 HWC -> f32 NCHW with the model's train-time augmentation fused (cnn.py:74-86):
 * keras RandomFlip("horizontal") -> RandomRotation (bilinear, fill_mode="reflect") ->
 * RandomContrast on the [0,1] image, then Normalization (x-mean)/denom.  aug4[n] =
 * {flip 0/1, cos, sin, contrast factor} (device, drawn by the host RNG); mean3/denom3 are
 * HOST pointers (or both null); means_ws is a device scratch of n*24 floats
 * (8 partial channel sums per image).
 * Stochastic layers: statistical parity with keras, exact parity with oracle/cnn_ref.py. */
int lf_input_stage_f32(const uint8_t* in, float* out, int n, int h, int w, const float* aug4,
                       const float* mean3, const float* denom3, float* means_ws,
                       lf_stream_t stream);

/* ---- BatchNorm (keras BatchNormalization: momentum 0.99, eps 1e-3; cnn.py:30,45) ---- */
/* out = act(x*scale[c] + shift[c]) over [n][c][hw] (BN apply, optional ReLU). */
int lf_scale_shift_act_f32(const float* x, float* out, int n, int c, int hw, const float* scale,
                           const float* shift, int relu, lf_stream_t stream);
size_t lf_bn_workspace(int c);
/* Training statistics of y [n][c][hw]: batch mean / biased variance per channel; writes
 * mean, invstd = 1/sqrt(var+eps), scale = gamma*invstd, shift = beta - mean*scale, and
 * updates moving_mean/var <- moving*momentum + batch*(1-momentum). */
int lf_bn_train_stats_f32(const float* y, int n, int c, int hw, const float* gamma,
                          const float* beta, float* moving_mean, float* moving_var, float momentum,
                          float eps, float* mean, float* invstd, float* scale, float* shift,
                          void* workspace, size_t ws_bytes, lf_stream_t stream);
/* Same outputs as lf_bn_train_stats_f32 from the tile sums of lf_conv2d_stats_f32 (taken about
 * moving_mean, which must not have changed in between); tiles = lf_conv2d_stats_tiles(...). */
int lf_bn_train_stats_tiles_f32(const float* tile_part, long long tiles, int n, int c, int hw,
                                const float* gamma, const float* beta, float* moving_mean,
                                float* moving_var, float momentum, float eps, float* mean,
                                float* invstd, float* scale, float* shift, void* workspace,
                                size_t ws_bytes, lf_stream_t stream);
/* Inference scale/shift from the moving statistics. */
int lf_bn_infer_scale_shift_f32(int c, const float* gamma, const float* beta,
                                const float* moving_mean, const float* moving_var, float eps,
                                float* scale, float* shift, lf_stream_t stream);
/* BatchNorm backward with the upstream chain folded in: dz = g*alpha_nc[n][c] + add_nc[n][c]
 * (both optional), zeroed where the forward's ReLU was inactive when relu != 0 — the mask is
 * recomputed as y*scale[c]+shift[c] > 0 from the pre-BN tensor, so the activation is never
 * stored; dy = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)); dgamma = sum dz*xhat,
 * dbeta = sum dz.  plane_g / plane_m (optional, [n][c][2], relu case only): when the producer
 * of g already left per-plane sums {sum g*mask, sum g*mask*y} (lf_block_tail_bwd_f32) and the
 * forward left {sum mask, sum mask*y} (lf_gap_f32; needed with add_nc), the two channel sums
 * (with relu == 0 the plane sums are unmasked and alpha_nc / add_nc must be null)
 * come from those and g / y are read once (for dy) instead of twice.  have_sums != 0: dgamma /
 * dbeta already hold the sums (lf_bn_bwd_sums_tiles_f32) and only the apply pass runs. */
int lf_bn_bwd_f32(const float* g, const float* alpha_nc, const float* add_nc, const float* y,
                  const float* mean, const float* invstd, const float* scale, const float* shift,
                  int relu, const float* gamma, float* dy, float* dgamma, float* dbeta,
                  const float* plane_g, const float* plane_m, int have_sums, int n, int c, int hw,
                  void* workspace, size_t ws_bytes, lf_stream_t stream);

/* The two channel sums of lf_bn_bwd_f32 without the apply pass: dgamma, dbeta and coef [5][c] =
 * {scale, shift, P, Q, R} such that dy = P*dz + Q*y + R with dz = (g*alpha+add)*[y*scale+shift>0
 * or !relu].  lf_conv2d_wgrad_bn_f32 forms dy from g and y with these while it computes the
 * weight gradient, so the standalone apply pass (2 reads + 1 write) disappears. */
int lf_bn_bwd_sums_f32(const float* g, const float* alpha_nc, const float* add_nc, const float* y,
                       const float* mean, const float* invstd, const float* scale,
                       const float* shift, int relu, const float* gamma, float* dgamma, float* dbeta,
                       float* coef, const float* plane_g, const float* plane_m, int n, int c, int hw,
                       void* workspace, size_t ws_bytes, lf_stream_t stream);

/* lf_bn_bwd_sums_f32 from the tile sums of lf_conv2d_bnbwd_f32 (no alpha/add). */
int lf_bn_bwd_sums_tiles_f32(const float* tile_part, long long tiles, const float* mean,
                             const float* invstd, const float* scale, const float* shift,
                             const float* gamma, float* dgamma, float* dbeta, float* coef, int n,
                             int c, int hw, void* workspace, size_t ws_bytes, lf_stream_t stream);

/* ---- pooling / broadcast ---- */
/* out[p] = mean over hw of act(x[p][:]*scale[c]+shift[c]), c = p % C (GlobalAveragePooling2D,
 * cnn.py:13,98; scale/shift null = plain mean; relu applies with the prologue).  mask_sums
 * (optional, [planes][2]) receives {count of x*scale+shift > 0, sum of x over those}. */
int lf_gap_f32(const float* x, float* out, int planes, int hw, int c, const float* scale,
               const float* shift, int relu, float* mask_sums, lf_stream_t stream);
/* out[p][:] = v[p]*scale (GAP backward). */
int lf_bcast_planes_f32(const float* v, float* out, int planes, int hw, float scale,
                        lf_stream_t stream);

/* ---- Squeeze-Excite (cnn.py:9-17): s = sigmoid(relu(m w1 + b1) w2 + b2) ---- */
/* m [n][c], w1 [c][cr], w2 [cr][c]; z1 = hidden activations (saved for backward).
 * Backward writes dm = dL/dm * dm_scale (dm_scale = 1/(H*W) folds the GlobalAveragePooling2D
 * the squeeze came through). */
int lf_se_fwd_f32(const float* m, const float* w1, const float* b1, const float* w2,
                  const float* b2, float* z1, float* s, int n, int c, int cr, lf_stream_t stream);
size_t lf_se_bwd_workspace(int n, int c, int cr);
int lf_se_bwd_f32(const float* ds, const float* m, const float* z1, const float* s,
                  const float* w1, const float* w2, float* dm, float* dw1, float* db1, float* dw2,
                  float* db2, int n, int c, int cr, float dm_scale, void* workspace,
                  size_t ws_bytes, lf_stream_t stream);

/* ---- residual tail (cnn.py:47-48,94-96): Add -> ReLU -> SpatialDropout2D -> MaxPool2D(2) ---- */
/* r = relu(sc' + a*s[n][c]) with a = relu(y*a_scale[c]+a_shift[c]) (BN2+ReLU fused; a = y when
 * a_scale is null) and sc' = act(sc*sc_scale[c]+sc_shift[c]) (projection BN, or BN+ReLU of the
 * producer when sc_relu) or sc;  p = drop[n][c] * maxpool2x2(r) (drop = 0 or 1/(1-rate)).
 * r itself is not stored: route [n][c][h/2][w/2] (one byte per pooled value) records where the
 * value came from (bits 0-1: first maximum in scan order) and whether it was > 0 (bit 2). */
int lf_block_tail_fwd_f32(const float* y, const float* a_scale, const float* a_shift,
                          const float* s, const float* sc, const float* sc_scale,
                          const float* sc_shift, int sc_relu, const float* drop, uint8_t* route,
                          float* p, int n, int c, int h, int w, lf_stream_t stream);
/* dr = gradient wrt (sc' + a*s): dp*drop routed to the recorded position of each window whose
 * maximum was > 0; ds[n][c] = sum_hw dr*a (SE gate gradient); plane_sums [n][c][2] =
 * {sum dr*[a>0], sum dr*[a>0]*y} for lf_bn_bwd_f32 (y goes with ds / plane_sums; plane_sums
 * needs a_scale); sc_y (the projection shortcut's pre-BN tensor, optional) with sc_sums
 * [n][c][2] = {sum dr, sum dr*sc_y}: the same for the shortcut's BatchNormalization (no mask). */
int lf_block_tail_bwd_f32(const float* dp, const uint8_t* route, const float* y,
                          const float* a_scale, const float* a_shift, const float* drop, float* dr,
                          float* ds, float* plane_sums, const float* sc_y, float* sc_sums, int n,
                          int c, int h, int w, lf_stream_t stream);

/* ---- head (cnn.py:98-101; train/utils.py:30-35) ---- */
/* probs = softmax(feat w + b), w [f][c]; loss[n] = -sum_j ytrue[n][j] log(clip(probs)). */
int lf_head_fwd_f32(const float* feat, const float* w, const float* b, const float* ytrue,
                    float* probs, float* loss, int n, int f, int c, lf_stream_t stream);
int lf_head_bwd_f32(const float* feat, const float* w, const float* probs, const float* ytrue,
                    float* dlogits, float* dfeat, float* dw, float* db, int n, int f, int c,
                    float inv_n, lf_stream_t stream);
int lf_mul_f32(const float* a, const float* b, float* out, size_t count, lf_stream_t stream);

/* ---- optimizer (train/utils.py:17-27 AdamW + clipnorm; :44-57 EMA) ---- */
/* Flat buffers; tensor t occupies [offsets[t], offsets[t+1]).  Per tensor: g' = g + 2*l2[t]*w
 * (kernel_regularizer gradient), clip_by_norm(g', clipnorm) per tensor (0 = off), decoupled
 * decay w -= lr*wd*w, Adam with bias correction at `step` (1-based), then
 * ema = copy ? w : decay*ema + (1-decay)*w (ema may be null).  norms_out: ntensors floats (the
 * pre-clip gradient norms, for logging); workspace: lf_adamw_workspace(ntensors) bytes. */
size_t lf_adamw_workspace(int ntensors);
int lf_adamw_step_f32(float* param, const float* grad, float* m, float* v, float* ema,
                      const long long* offsets, const float* l2, int ntensors, long long max_count,
                      float lr, float beta1, float beta2, float eps, float weight_decay,
                      float clipnorm, long long step, float ema_decay, int ema_copy,
                      float* norms_out, void* workspace, size_t ws_bytes, lf_stream_t stream);
int lf_ema_update_f32(float* ema, const float* w, size_t count, float decay, int copy,
                      lf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LEAFHIP_H */
