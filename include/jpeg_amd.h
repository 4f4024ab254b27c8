/* jpeg_amd.h -- C ABI of the MI355X (gfx950) spectral pipeline.
 *
 * Drop-in boundary for the one data-parallel hot path of tayloraswift/jpeg
 * (reference @ 2024_08_07; citations below are sources/jpeg/<file>:<lines>):
 *
 *   decode   Spectral.idct()              decode.swift:4154-4165  (per plane :4101-4133)
 *            Planar.interleaved(cosite:)  decode.swift:4182-4276
 *            Rectangular.unpack(as:)      decode.swift:4291-4298  (jpeg.swift:441-453, 493-572)
 *   encode   Rectangular.pack(...)        encode.swift:453-464    (jpeg.swift:463-478, 527-599)
 *            Rectangular.decomposed()     encode.swift:389-425
 *            Planar.fdct(quanta:)         encode.swift:353-370    (per plane :199-248)
 *
 * The reference has no FFI for this path (it is pure Swift); these entry points are
 * what a Swift shim binds with @_silgen_name / a module map.
 * Results are bit-identical to the reference: the float32 operation order of the
 * reference is reproduced exactly (kernels are built with -ffp-contract=off).
 *
 * Conventions
 *   - plain C, no exceptions, no aborts: every call returns 0 (JPEG_AMD_OK) or a
 *     negative jpeg_amd_status; contract violations the reference traps on
 *     (precondition failures) come back as JPEG_AMD_EINVAL.
 *   - `d_` parameters are DEVICE pointers (HBM); `h_` parameters are HOST pointers.
 *     Quantisation tables are tiny and are HOST pointers unless named `d_`.
 *   - a ctx owns one HIP stream (or borrows the caller's), scratch memory and
 *     timing events.  A ctx is single-threaded; different ctxs may run concurrently.
 *     All device entry points are asynchronous on the ctx stream.
 *   - coefficient planes: int16 [units_y][units_x][64], ZIGZAG order inside a block
 *     (decode.swift:1434, 1466).  Spatial planes: uint16 [8*units_y][8*units_x]
 *     (decode.swift:1548-1598).  Rectangular: uint16 [H][W][nplanes]
 *     (decode.swift:1650-1718).  Colours: uint8 [H*W][3] (jpeg.swift:160-269).
 */
#ifndef JPEG_AMD_H
#define JPEG_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JPEG_AMD_VERSION 100  /* 0.1.0 */
#define JPEG_AMD_MAX_PLANES 4

typedef enum jpeg_amd_status {
    JPEG_AMD_OK      = 0,
    JPEG_AMD_EINVAL  = -1, /* bad argument / violated precondition of the reference */
    JPEG_AMD_ENOMEM  = -2, /* device or host allocation failed */
    JPEG_AMD_EHIP    = -3, /* HIP runtime error; see jpeg_amd_last_hip_error */
    JPEG_AMD_ENODEV  = -4, /* no such device / no gfx950 code object for it */
    JPEG_AMD_ENOSUP  = -5  /* valid in the reference but not implemented here */
} jpeg_amd_status;

typedef struct jpeg_amd_ctx jpeg_amd_ctx;

/* Image geometry = the part of JPEG.Layout<Format> + Spectral/Planar sizes the
 * hot path reads (jpeg.swift:1084-1635, decode.swift:2181-2190, 2456-2495).
 * Only RECOGNISED planes are listed; scale is the max sampling factor over ALL
 * components of the frame (it can exceed every listed factor). */
typedef struct jpeg_amd_layout {
    int32_t width, height;                   /* image size in pixels, > 0 */
    int32_t precision;                       /* Format.precision, 1..16 */
    int32_t nplanes;                         /* 1..JPEG_AMD_MAX_PLANES */
    int32_t scale_x, scale_y;                /* Layout.scale */
    int32_t factor_x[JPEG_AMD_MAX_PLANES];   /* Component.factor */
    int32_t factor_y[JPEG_AMD_MAX_PLANES];
    int32_t units_x[JPEG_AMD_MAX_PLANES];    /* Plane.units (data units per row / column) */
    int32_t units_y[JPEG_AMD_MAX_PLANES];
    int32_t qi[JPEG_AMD_MAX_PLANES];         /* Plane.q: index into the quanta array */
} jpeg_amd_layout;

/* colour targets of Rectangular.unpack / pack (the built-in JPEG.Color types) */
typedef enum jpeg_amd_color {
    JPEG_AMD_COLOR_YCC8 = 0,  /* JPEG.YCbCr  jpeg.swift:493-539 */
    JPEG_AMD_COLOR_RGB8 = 1   /* JPEG.RGB    jpeg.swift:551-599 */
} jpeg_amd_color;

/* ---- library / context --------------------------------------------------------- */
int         jpeg_amd_version(void);
const char *jpeg_amd_strerror(int status);
int         jpeg_amd_device_count(int *count);
/* stream: the hipStream_t to launch on (borrowed, e.g. torch's current stream; NULL is
 * the device's default stream).  With JPEG_AMD_CTX_OWN_STREAM in flags, `stream` is
 * ignored and the ctx creates (and later destroys) a private non-blocking stream. */
#define JPEG_AMD_CTX_OWN_STREAM 1
int         jpeg_amd_ctx_create(int device, void *stream, int flags, jpeg_amd_ctx **ctx);
int         jpeg_amd_ctx_destroy(jpeg_amd_ctx *ctx);
int         jpeg_amd_ctx_synchronize(jpeg_amd_ctx *ctx);
int         jpeg_amd_last_hip_error(const jpeg_amd_ctx *ctx); /* raw hipError_t */
/* fill in units_x/units_y = ceil(size * factor / (8 * scale))  decode.swift:2606-2616 */
int         jpeg_amd_layout_units(jpeg_amd_layout *layout);

/* ---- device memory + timing (so a non-HIP host language can keep data resident) -- */
int jpeg_amd_malloc(jpeg_amd_ctx *ctx, size_t bytes, void **d_ptr);
int jpeg_amd_free(jpeg_amd_ctx *ctx, void *d_ptr);
int jpeg_amd_memcpy_h2d(jpeg_amd_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int jpeg_amd_memcpy_d2h(jpeg_amd_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
/* HIP events on the ctx stream; end() synchronises and returns elapsed ms */
int jpeg_amd_timer_begin(jpeg_amd_ctx *ctx);
int jpeg_amd_timer_end(jpeg_amd_ctx *ctx, float *elapsed_ms);

/* ---- decode stages (device-resident) ---------------------------------------------- */

/* Spectral.Plane.idct(quanta:precision:)  decode.swift:4101-4133 (+ modulate :3984-4017,
 * load :4020-4039, idct8 :4042-4093, idct8x8 :4095-4099).  One plane. */
int jpeg_amd_idct_plane(jpeg_amd_ctx *ctx, const int16_t *d_coef, int units_x, int units_y,
                        const uint16_t h_quanta_zigzag[64], int precision,
                        uint16_t *d_plane);

/* Spectral.idct()  decode.swift:4154-4165: every plane of an image.
 * h_quanta: [ntables][64] zigzag; plane p uses table layout->qi[p]. */
int jpeg_amd_spectral_idct(jpeg_amd_ctx *ctx, const jpeg_amd_layout *layout,
                           const int16_t *const d_coef[], const uint16_t *h_quanta,
                           int ntables, uint16_t *const d_planes[]);

/* Planar.interleaved(cosite:)  decode.swift:4182-4276 */
int jpeg_amd_planar_interleaved(jpeg_amd_ctx *ctx, const jpeg_amd_layout *layout,
                                const uint16_t *const d_planes[], int cosited,
                                uint16_t *d_rect);

/* Rectangular.unpack(as:) for the built-in 8-bit colour targets  decode.swift:4291-4298.
 * nplanes = 1 (y8 / nonconforming1x8) or 3 (ycc8 / nonconforming3x8). */
int jpeg_amd_rectangular_unpack(jpeg_amd_ctx *ctx, const uint16_t *d_rect, size_t npixels,
                                int nplanes, jpeg_amd_color color, uint8_t *d_pixels);

/* Fused Spectral -> pixels: == idct().interleaved(cosite:).unpack(as:) bit for bit,
 * without materialising Planar / Rectangular in HBM (8-bit formats, 1 or 3 planes).
 * n_images images of identical layout; image i reads d_coef[p] + i*coef_stride[p]
 * (int16 elements), table set i*quanta_stride (uint16 elements) of d_quanta and writes
 * d_pixels + i*pixel_stride (bytes).  d_quanta: DEVICE pointer [..][ntables][64]. */
int jpeg_amd_decode_batch(jpeg_amd_ctx *ctx, const jpeg_amd_layout *layout, int n_images,
                          const int16_t *const d_coef[], const size_t coef_stride[],
                          const uint16_t *d_quanta, size_t quanta_stride, int ntables,
                          int cosited, jpeg_amd_color color,
                          uint8_t *d_pixels, size_t pixel_stride);
/* single image, host tables */
int jpeg_amd_decode(jpeg_amd_ctx *ctx, const jpeg_amd_layout *layout,
                    const int16_t *const d_coef[], const uint16_t *h_quanta, int ntables,
                    int cosited, jpeg_amd_color color, uint8_t *d_pixels);

/* Fused Spectral -> Rectangular: == idct().interleaved(cosite:) bit for bit (decode.swift:4154-4165, 4182-4276) -- what
 * Rectangular.decompress(stream:cosite:) runs behind the entropy decoder (decode.swift:4367-4374) -- for ANY JPEG.Format
 * (jpeg.swift:21-56; examples/custom-color/main.swift:41-63): precision 1 .. 16, 1 .. 4 planes, centred or cosited.
 * Layouts whose planes lie at the image's scale or at half of it per axis (factors 1 | 2, scale <= 2) take ONE launch with no
 * Planar intermediate in HBM (kernels_generic.hip; the reference's literal operation sequence: true division, .rounded());
 * every other layout (factors 3, 4 ...) runs the staged kernels through the context's scratch -- same result either way.
 * d_rect: uint16 [H][W][nplanes]; batch strides as in jpeg_amd_decode_batch (rect_stride in uint16 elements). */
int jpeg_amd_spectral_rectangular_batch(jpeg_amd_ctx *ctx, const jpeg_amd_layout *layout, int n_images,
                                        const int16_t *const d_coef[], const size_t coef_stride[],
                                        const uint16_t *d_quanta, size_t quanta_stride, int ntables,
                                        int cosited, uint16_t *d_rect, size_t rect_stride);
/* single image, host tables */
int jpeg_amd_spectral_rectangular(jpeg_amd_ctx *ctx, const jpeg_amd_layout *layout,
                                  const int16_t *const d_coef[], const uint16_t *h_quanta, int ntables,
                                  int cosited, uint16_t *d_rect);

/* Fused Rectangular -> Spectral: == decomposed().fdct(quanta:) bit for bit (encode.swift:389-425, 199-248) -- what
 * Rectangular.compress(stream:quanta:) runs in front of the entropy coder (encode.swift:2031) -- for ANY JPEG.Format, the
 * mirror image of jpeg_amd_spectral_rectangular: the layouts that take one launch there take one here (no Planar
 * intermediate in HBM, the reference's literal operation sequence), every other layout runs the staged kernels through the
 * context's scratch.  d_rect: uint16 [H][W][nplanes]; d_coef[p]: int16 [units_y][units_x][64], zigzag. */
int jpeg_amd_rectangular_spectral_batch(jpeg_amd_ctx *ctx, const jpeg_amd_layout *layout, int n_images,
                                        const uint16_t *d_rect, size_t rect_stride, const uint16_t *d_quanta,
                                        size_t quanta_stride, int ntables, int16_t *const d_coef[],
                                        const size_t coef_stride[]);
/* single image, host tables */
int jpeg_amd_rectangular_spectral(jpeg_amd_ctx *ctx, const jpeg_amd_layout *layout, const uint16_t *d_rect,
                                  const uint16_t *h_quanta, int ntables, int16_t *const d_coef[]);

/* ---- encode stages (device-resident) ---------------------------------------------- */

/* Rectangular.pack(size:layout:metadata:pixels:)  encode.swift:453-464 */
int jpeg_amd_rectangular_pack(jpeg_amd_ctx *ctx, const uint8_t *d_pixels, size_t npixels,
                              int nplanes, jpeg_amd_color color, uint16_t *d_rect);

/* Rectangular.decomposed()  encode.swift:389-425 */
int jpeg_amd_rectangular_decomposed(jpeg_amd_ctx *ctx, const jpeg_amd_layout *layout,
                                    const uint16_t *d_rect, uint16_t *const d_planes[]);

/* Spectral.Plane.fdct(_:quanta:precision:)  encode.swift:199-248 */
int jpeg_amd_fdct_plane(jpeg_amd_ctx *ctx, const uint16_t *d_plane, int units_x, int units_y,
                        const uint16_t h_quanta_zigzag[64], int precision, int16_t *d_coef);

/* Planar.fdct(quanta:)  encode.swift:353-370 */
int jpeg_amd_planar_fdct(jpeg_amd_ctx *ctx, const jpeg_amd_layout *layout,
                         const uint16_t *const d_planes[], const uint16_t *h_quanta,
                         int ntables, int16_t *const d_coef[]);

/* Fused pixels -> Spectral: == pack(...).decomposed().fdct(quanta:) bit for bit
 * (8-bit formats, 1 or 3 planes).  Strides as in jpeg_amd_decode_batch. */
int jpeg_amd_encode_batch(jpeg_amd_ctx *ctx, const jpeg_amd_layout *layout, int n_images,
                          const uint8_t *d_pixels, size_t pixel_stride, jpeg_amd_color color,
                          const uint16_t *d_quanta, size_t quanta_stride, int ntables,
                          int16_t *const d_coef[], const size_t coef_stride[]);
int jpeg_amd_encode(jpeg_amd_ctx *ctx, const jpeg_amd_layout *layout,
                    const uint8_t *d_pixels, jpeg_amd_color color,
                    const uint16_t *h_quanta, int ntables, int16_t *const d_coef[]);

/* ---- host-buffer conveniences: what the Swift shim calls ---------------------------
 * Same semantics as the calls above with every buffer in HOST memory: the library
 * uploads inputs, runs the kernels and downloads outputs (synchronous). */
int jpeg_amd_host_spectral_idct(jpeg_amd_ctx *ctx, const jpeg_amd_layout *layout,
                                const int16_t *const h_coef[], const uint16_t *h_quanta,
                                int ntables, uint16_t *const h_planes[]);
int jpeg_amd_host_planar_interleaved(jpeg_amd_ctx *ctx, const jpeg_amd_layout *layout,
                                     const uint16_t *const h_planes[], int cosited,
                                     uint16_t *h_rect);
int jpeg_amd_host_rectangular_unpack(jpeg_amd_ctx *ctx, const uint16_t *h_rect,
                                     size_t npixels, int nplanes, jpeg_amd_color color,
                                     uint8_t *h_pixels);
int jpeg_amd_host_decode(jpeg_amd_ctx *ctx, const jpeg_amd_layout *layout,
                         const int16_t *const h_coef[], const uint16_t *h_quanta, int ntables,
                         int cosited, jpeg_amd_color color, uint8_t *h_pixels);
/* idct().interleaved(cosite:) with host buffers: ONE crossing of the link each way, one launch on the device for formats
 * whose planes lie at the image's scale or at half of it (jpeg_amd_spectral_rectangular). */
int jpeg_amd_host_spectral_rectangular(jpeg_amd_ctx *ctx, const jpeg_amd_layout *layout,
                                       const int16_t *const h_coef[], const uint16_t *h_quanta, int ntables,
                                       int cosited, uint16_t *h_rect);
/* decomposed().fdct(quanta:) with host buffers, likewise (jpeg_amd_rectangular_spectral) */
int jpeg_amd_host_rectangular_spectral(jpeg_amd_ctx *ctx, const jpeg_amd_layout *layout, const uint16_t *h_rect,
                                       const uint16_t *h_quanta, int ntables, int16_t *const h_coef[]);
int jpeg_amd_host_rectangular_pack(jpeg_amd_ctx *ctx, const uint8_t *h_pixels, size_t npixels,
                                   int nplanes, jpeg_amd_color color, uint16_t *h_rect);
int jpeg_amd_host_rectangular_decomposed(jpeg_amd_ctx *ctx, const jpeg_amd_layout *layout,
                                         const uint16_t *h_rect, uint16_t *const h_planes[]);
int jpeg_amd_host_planar_fdct(jpeg_amd_ctx *ctx, const jpeg_amd_layout *layout,
                              const uint16_t *const h_planes[], const uint16_t *h_quanta,
                              int ntables, int16_t *const h_coef[]);
int jpeg_amd_host_encode(jpeg_amd_ctx *ctx, const jpeg_amd_layout *layout,
                         const uint8_t *h_pixels, jpeg_amd_color color,
                         const uint16_t *h_quanta, int ntables, int16_t *const h_coef[]);

/* ---- host side of the path's INPUT ("next" rows) -------------------------
 * Huffman entropy decoding stays on the host CPU (north_star); these entry points turn a JPEG
 * byte stream into the Spectral containers the kernels consume, like
 * JPEG.Data.Spectral.decompress(stream:) does through JPEG.Context (decode.swift:3554-3961,
 * 4315).  Baseline / extended sequential and progressive Huffman, restart intervals, 1..4
 * components, 8-bit or 12-bit.  No GPU is needed for the first two. */
typedef struct jpeg_amd_frame_info {
    int32_t width, height, precision, ncomponents;
    int32_t process;                          /* 0 baseline, 1 extended sequential, 2 progressive */
    int32_t scale_x, scale_y;                 /* Layout.scale */
    int32_t id[JPEG_AMD_MAX_PLANES];          /* component identifiers, frame order */
    int32_t factor_x[JPEG_AMD_MAX_PLANES], factor_y[JPEG_AMD_MAX_PLANES];
    int32_t units_x[JPEG_AMD_MAX_PLANES], units_y[JPEG_AMD_MAX_PLANES];
    int32_t nscans, restart_interval;
} jpeg_amd_frame_info;

/* JPEG.Table.Huffman<Symbol>.Decoder subscript  decode.swift:1243-1261 (table construction :1008-1240): build the
 * decoder of a DHT table -- counts[l] codewords of length l + 1, `values` in codeword order -- and decode the
 * codeword at the top of the 16-bit window `window` (big-endian, left-aligned).  A window that is no codeword gives
 * symbol 0 and length 16: the reference renders damaged streams that way instead of failing, and so does
 * jpeg_amd_jpeg_decode_spectral.  EINVAL for tables the reference's initialiser rejects (over-subscribed lengths,
 * more than 256 values).  Known-answer vectors: tests/unit/tests.swift:141-461.  Host only. */
int jpeg_amd_huffman_lookup(const uint8_t counts[16], const uint8_t *values, int nvalues,
                            uint16_t window, int32_t *symbol, int32_t *length);
/* JPEG.Table.Huffman<Symbol>.init(frequencies:target:)  encode.swift:597-760: the code Spectral.compress builds for
 * a scan from its symbol frequencies (optimal lengths, the reference's tie-breaking and its 16-bit limiter), as a
 * DHT table: counts[16], values[] in codeword order (*nvalues of them, <= 256).  freq[v] <= 0: symbol unused. */
int jpeg_amd_huffman_build(const int64_t freq[256], uint8_t counts[16], uint8_t values[256], int32_t *nvalues);

/* parse the headers (and walk the scans) without decoding: geometry for buffer allocation */
int jpeg_amd_jpeg_inspect(const uint8_t *h_jpeg, size_t nbytes, jpeg_amd_frame_info *info);
/* entropy-decode every scan into caller-allocated planes h_coef[c]: int16 [units_y][units_x][64]
 * zigzag (zeroed here first); h_quanta[c] receives the table bound to component c (zigzag).
 * Damaged entropy-coded data is decoded the way the reference decodes it, not refused: a 16-bit window that matches
 * no codeword is symbol 0 of length 16 (decode.swift:1255-1258), a stream that ends early is padded with 1-bits. */
int jpeg_amd_jpeg_decode_spectral(const uint8_t *h_jpeg, size_t nbytes, int16_t *const h_coef[],
                                  uint16_t h_quanta[][64], jpeg_amd_frame_info *info);
/* The same with the restart intervals of every scan decoded by `nthreads` host threads
 * (<= 0: all cores): intervals are independent bit streams (DC predictors and EOB runs reset at
 * RSTn, decode.swift:3210, 3500-3502).  Files without DRI, or with a damaged marker sequence,
 * take the sequential path. */
int jpeg_amd_jpeg_decode_spectral_mt(const uint8_t *h_jpeg, size_t nbytes, int16_t *const h_coef[],
                                     uint16_t h_quanta[][64], jpeg_amd_frame_info *info, int nthreads);
/* The image as it stands after the first `max_scans` scans (0 = all): what JPEG.Context hands out
 * between scans (decode.swift:3554-3961, examples/decode-online) -- progressive previews.
 * Components no scan has reached yet are all zero and get a table of ones. */
int jpeg_amd_jpeg_decode_spectral_partial(const uint8_t *h_jpeg, size_t nbytes, int16_t *const h_coef[],
                                          uint16_t h_quanta[][64], jpeg_amd_frame_info *info,
                                          int nthreads, int max_scans);
/* A decoder fed by a GROWING byte stream -- JPEG.Context driven by a Bytestream.Source that runs
 * dry (decode.swift:3554-3961; examples/decode-online): push whatever bytes have arrived; every
 * segment and every scan that is complete by then is consumed, incomplete ones wait for the next
 * push.  *scans_done counts the scans decoded so far, *finished is set at EOI.  The decoder owns
 * the planes; jpeg_amd_stream_snapshot copies them (and the table of every component, ones for a
 * component no scan has reached yet) into caller buffers sized from jpeg_amd_stream_info.
 * Because it owns the planes it refuses frames of more than 32 Mi blocks in all (ENOMEM; larger frames go through
 * the one-shot entry points, where the caller allocates), and its first error is final: every later push returns
 * the same status without touching the decoder's state. */
/* The same file as SPARSE coefficients: one 32-bit entry per nonzero coefficient (every block's DC has one) --
 * bits 0-15 the int16 coefficient, bits 16-21 its zigzag index, bit 31 set on the last entry of its block -- the entries of a
 * block in a row, and per block of the frame (planes in frame order, plane c's block (x, y) at
 * sum(units_x * units_y of the planes before c) + y * units_x + x) the index of its first entry in h_desc, 0xFFFFFFFF for a
 * block no scan reached.  What the host has to write and PCIe has to carry is an eighth of the planes for a typical file;
 * jpeg_amd_spectral_expand rebuilds the planes on the device.  Sequential files with every restart marker in place only:
 * JPEG_AMD_ENOSUP for anything else (progressive, a damaged marker sequence) and when `capacity` entries do not suffice --
 * decode those with jpeg_amd_jpeg_decode_spectral. */
int jpeg_amd_jpeg_decode_sparse(const uint8_t *data, size_t nbytes, uint32_t *h_desc, size_t ndesc,
                                uint32_t *h_entries, size_t capacity, size_t *nentries,
                                uint16_t h_quanta[][64], jpeg_amd_frame_info *info);

/* Sparse coefficients -> Spectral planes on the device (the other half of jpeg_amd_jpeg_decode_sparse): image i reads its
 * descriptors at d_desc + i * desc_stride and its entries at d_entries + i * entries_stride (elements) and has every block
 * of its planes d_coef[p] + i * coef_stride[p] written; d_skip (optional, one byte per image): nonzero = the image's planes
 * are left as they are.  Asynchronous on the context's stream. */
int jpeg_amd_spectral_expand_batch(jpeg_amd_ctx *ctx, const jpeg_amd_layout *layout, int n_images,
                                   const uint32_t *d_desc, size_t desc_stride, const uint32_t *d_entries,
                                   size_t entries_stride, const uint8_t *d_skip, int16_t *const d_coef[],
                                   const size_t coef_stride[]);

typedef struct jpeg_amd_stream jpeg_amd_stream;
jpeg_amd_stream *jpeg_amd_stream_create(void);
void jpeg_amd_stream_destroy(jpeg_amd_stream *stream);
int jpeg_amd_stream_push(jpeg_amd_stream *stream, const uint8_t *h_bytes, size_t nbytes, int *scans_done,
                         int *finished);
int jpeg_amd_stream_info(const jpeg_amd_stream *stream, jpeg_amd_frame_info *info);
int jpeg_amd_stream_snapshot(const jpeg_amd_stream *stream, int16_t *const h_coef[], uint16_t h_quanta[][64]);
/* Rectangular.decompress(stream:cosite:) + unpack(as:)  (decode.swift:4367, os.swift:375):
 * JPEG bytes in, H*W colours of 3 bytes out (host memory); 8-bit images of 1 or 3 components. */
int jpeg_amd_decompress(jpeg_amd_ctx *ctx, const uint8_t *h_jpeg, size_t nbytes, int cosited,
                        jpeg_amd_color color, uint8_t *h_pixels, size_t pixel_capacity,
                        jpeg_amd_frame_info *info);

/* Rectangular<Format>.decompress(stream:cosite:) for ANY JPEG.Format  (decode.swift:4367-4374; what
 * examples/custom-color/main.swift:190-200 does with its 12-bit four-component format): JPEG bytes in host memory ->
 * entropy decoding on the host (restart intervals on `nthreads` threads, <= 0: all cores) -> idct() + interleaved(cosite:)
 * on the GPU -> the samples, uint16 [H][W][n], in host memory.  n = nrecognized, the first nrecognized components of the
 * frame (0: all of them); the components behind them are non-recognised by the format (jpeg.swift:21-56): they take part in
 * the image's scale and in nothing else.  Precision 1 .. 16, 1 .. 4 components, sequential or progressive, any sampling
 * factors.  rect_capacity: the size of h_rect in samples (H * W * n are written; EINVAL if it is smaller). */
int jpeg_amd_decompress_rectangular(jpeg_amd_ctx *ctx, const uint8_t *h_jpeg, size_t nbytes, int cosited,
                                    int nrecognized, int nthreads, uint16_t *h_rect, size_t rect_capacity,
                                    jpeg_amd_frame_info *info);

/* The same for n_images files of ONE frame geometry (a burst, the frames of an MJPEG stream):
 * `nthreads` host threads (<= 0: all cores) entropy-decode into pinned buffers, the device decodes
 * a chunk of images per launch.  h_pixels: image i at h_pixels + i * pixel_stride (0 = W*H*3).
 * This is the restart-interval / image-level parallelism of SURVEY.md 8f-1 on the host side.
 * h_pixels may be pageable (downloaded into the context's pinned slots, copied out by the host threads) or page-locked
 * (hipHostMalloc / hipHostRegister: downloaded straight into it).
 * Threads: the calling thread directs (it submits chunks to the device and polls their events); `nthreads` threads beside it
 * entropy-decode, and for pageable output up to min(nthreads, 16) more copy pixels out of the pinned slots -- all of them kept
 * in the context between calls.  If no helper thread can be started the call runs synchronously on the calling thread. */
int jpeg_amd_decompress_batch(jpeg_amd_ctx *ctx, const uint8_t *const h_jpeg[], const size_t nbytes[],
                              int n_images, int nthreads, int cosited, jpeg_amd_color color,
                              uint8_t *h_pixels, size_t pixel_stride, jpeg_amd_frame_info *info);
/* The same, with the pixels LEFT ON THE DEVICE (image i at d_pixels + i * pixel_stride): what crosses PCIe is the sparse form
 * of the coefficients (jpeg_amd_jpeg_decode_sparse) -- about 0.8 MB instead of 6.2 MB for a typical 1080p file -- and nothing
 * comes back.  Returns when the last chunk's kernels have finished. */
int jpeg_amd_decompress_batch_device(jpeg_amd_ctx *ctx, const uint8_t *const h_jpeg[], const size_t nbytes[],
                                     int n_images, int nthreads, int cosited, jpeg_amd_color color,
                                     uint8_t *d_pixels, size_t pixel_stride, jpeg_amd_frame_info *info);

/* ---- host side of the path's OUTPUT ("next" row) ------------------------------
 * The Huffman entropy encoder and file writer behind JPEG.Data.Spectral.compress(stream:)
 * (encode.swift:1918-1972): optimised Huffman tables per scan (:700-760), sequential scans,
 * interleaved or not (:962-1011, 1211-1384), table definitions grouped like
 * JPEG.Layout.definitions (jpeg.swift:1383-1442).  Output is byte-identical to the reference's
 * files for the same coefficients.  Sequential and progressive (DC / AC, first pass and
 * refinement, EOB runs: encode.swift:1013-1206, 1386-1557) Huffman scans. */
typedef struct jpeg_amd_scan {                /* JPEG.Header.Scan (jpeg.swift:1640-1760) */
    int32_t ncomponents;
    int32_t component[JPEG_AMD_MAX_PLANES];   /* plane indices (frame order), ascending */
    int32_t dc[JPEG_AMD_MAX_PLANES];          /* Huffman table selectors 0..1 (baseline) / 0..3 */
    int32_t ac[JPEG_AMD_MAX_PLANES];
    /* progressive process only; all zero = a sequential scan (.sequential(...)):
     *   band_lo = 0, band_hi = 1, refine = 0: .progressive(..., bits: bit...)     DC, first pass
     *   band_lo = 0, band_hi = 1, refine = 1: .progressive(..., bit: bit)         DC, one more bit
     *   band_lo >= 1,             refine = 0: .progressive(c, band:, bits: bit...) AC, first pass
     *   band_lo >= 1,             refine = 1: .progressive(c, band:, bit: bit)     AC, one more bit */
    int32_t band_lo, band_hi;                 /* zigzag band [band_lo, band_hi) */
    int32_t bit, refine;
} jpeg_amd_scan;

typedef struct jpeg_amd_jfif {                /* JPEG.JFIF */
    int32_t version_minor;                    /* 1.0, 1.1, 1.2 -> 0, 1, 2 */
    int32_t unit;                             /* 0 none, 1 dots per inch, 2 dots per centimetre */
    int32_t density_x, density_y;
} jpeg_amd_jfif;

typedef struct jpeg_amd_metadata {            /* JPEG.Metadata record, written after SOI in order */
    int32_t kind;                             /* 0 .jfif, 1 .application(app, data:), 2 .comment(data:) */
    int32_t app;                              /* kind 1: 0..15 */
    jpeg_amd_jfif jfif;                       /* kind 0 */
    const uint8_t *data;                      /* kinds 1, 2: segment payload */
    size_t size;
} jpeg_amd_metadata;

/* frame: width, height, precision, process (0 baseline, 1 extended), ncomponents, id[] (ascending),
 * factor_*[], units_*[] of the planes in h_coef[] (int16 [units_y][units_x][64], zigzag).
 * quanta_key[c]: quantisation-table key of component c (JPEG.Table.Quantization.Key);
 * h_quanta / h_quanta_keys: ntables tables of 64 zigzag values and their keys.
 * process 2 (progressive) takes progressive scans, 0 / 1 sequential ones.
 * frame->restart_interval > 0 (an extension: the reference's writer never emits DRI) cuts every
 * scan into restart intervals of that many MCUs, which the decoder then takes on several threads.
 * h_out == NULL only computes *nbytes. */
int jpeg_amd_jpeg_encode_spectral(const jpeg_amd_frame_info *frame, const int32_t *quanta_key,
                                  const int16_t *const h_coef[], const uint16_t *h_quanta,
                                  const int32_t *h_quanta_keys, int ntables,
                                  const jpeg_amd_scan *scans, int nscans,
                                  const jpeg_amd_metadata *metadata, int nmetadata, uint8_t *h_out,
                                  size_t capacity,
                                  size_t *nbytes);
/* The same writer fed with SPARSE coefficients (the format of jpeg_amd_jpeg_decode_sparse: h_desc one descriptor per block of the
 * frame, planes in frame order; h_entries / nentries the arena; a block's entries in ascending zigzag index, the DC first):
 * sequential scans only (JPEG_AMD_ENOSUP for a progressive frame).  Byte-identical to jpeg_amd_jpeg_encode_spectral on the
 * planes those entries expand to.  jpeg_amd_compress_batch brings the coefficients down from the device in this form. */
int jpeg_amd_jpeg_encode_sparse(const jpeg_amd_frame_info *frame, const int32_t *quanta_key, const uint32_t *h_desc,
                                const uint32_t *h_entries, size_t nentries, const uint16_t *h_quanta,
                                const int32_t *h_quanta_keys, int ntables, const jpeg_amd_scan *scans, int nscans,
                                const jpeg_amd_metadata *metadata, int nmetadata, uint8_t *h_out, size_t capacity,
                                size_t *nbytes);
/* Rectangular.pack(...).compress(stream:quanta:)  (encode.swift:456, 2031; os.swift:412): H*W
 * colours of 3 bytes in host memory -> colour conversion, downsampling, FDCT and quantisation on
 * the GPU -> entropy coding on the host -> JPEG bytes.  8-bit, 1 or 3 components, ids in
 * frame->id.  frame->units_* are filled in. */
int jpeg_amd_compress(jpeg_amd_ctx *ctx, jpeg_amd_frame_info *frame, const uint8_t *h_pixels,
                      jpeg_amd_color color, const int32_t *quanta_key, const uint16_t *h_quanta,
                      const int32_t *h_quanta_keys, int ntables, const jpeg_amd_scan *scans,
                      int nscans, const jpeg_amd_metadata *metadata, int nmetadata, uint8_t *h_out,
                      size_t capacity,
                      size_t *nbytes);

/* Rectangular<Format>.compress(stream:quanta:) for ANY JPEG.Format  (encode.swift:2031; examples/custom-color/
 * main.swift:132-188): samples uint16 [H][W][frame->ncomponents] in host memory -> decomposed() + fdct(quanta:) on the GPU
 * -> entropy coding on the host -> JPEG bytes.  frame: width, height, precision (1 .. 16), process, ncomponents (1 .. 4),
 * id[] (ascending), factor_*[]; units_* and scale_* are filled in.  Tables, scans and metadata as for jpeg_amd_compress
 * (quanta above 255 are written as 16-bit tables).  h_out == NULL only computes *nbytes. */
int jpeg_amd_compress_rectangular(jpeg_amd_ctx *ctx, jpeg_amd_frame_info *frame, const uint16_t *h_rect,
                                  const int32_t *quanta_key, const uint16_t *h_quanta,
                                  const int32_t *h_quanta_keys, int ntables, const jpeg_amd_scan *scans,
                                  int nscans, const jpeg_amd_metadata *metadata, int nmetadata, uint8_t *h_out,
                                  size_t capacity, size_t *nbytes);

/* The same for n_images pictures of ONE geometry, tables and scan progression: image i at
 * h_pixels + i * pixel_stride (0 = W*H*3); file i is written to h_out + i * out_stride and is
 * nbytes[i] long (EINVAL with nbytes[i] > out_stride: that buffer was too small).  One fused
 * encode launch per chunk, the entropy coding on `nthreads` host threads (<= 0: all cores). */
int jpeg_amd_compress_batch(jpeg_amd_ctx *ctx, jpeg_amd_frame_info *frame, const uint8_t *h_pixels,
                            size_t pixel_stride, int n_images, jpeg_amd_color color,
                            const int32_t *quanta_key, const uint16_t *h_quanta,
                            const int32_t *h_quanta_keys, int ntables, const jpeg_amd_scan *scans,
                            int nscans, const jpeg_amd_metadata *metadata, int nmetadata, int nthreads,
                            uint8_t *h_out, size_t out_stride, size_t nbytes[]);
/* The same with the pixels ALREADY ON THE DEVICE (picture i at d_pixels + i * pixel_stride): nothing is uploaded, the
 * coefficients come down and the files are written into host memory as above. */
int jpeg_amd_compress_batch_device(jpeg_amd_ctx *ctx, jpeg_amd_frame_info *frame, const uint8_t *d_pixels,
                            size_t pixel_stride, int n_images, jpeg_amd_color color,
                            const int32_t *quanta_key, const uint16_t *h_quanta,
                            const int32_t *h_quanta_keys, int ntables, const jpeg_amd_scan *scans,
                            int nscans, const jpeg_amd_metadata *metadata, int nmetadata, int nthreads,
                            uint8_t *h_out, size_t out_stride, size_t nbytes[]);

#ifdef __cplusplus
}
#endif
#endif /* JPEG_AMD_H */
