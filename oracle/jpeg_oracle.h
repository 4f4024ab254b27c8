/* jpeg_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the 8x8-block spectral pipeline of tayloraswift/jpeg
 * (reference @ 2024_08_07).  Every function cites the reference lines it follows
 * (paths relative to the reference checkout, sources/jpeg/...).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (jpeg_amd/) never links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement
 * against the reference's own committed outputs (regression golds, example
 * dumps, encode-basic coefficient streams) -- see tests/golden/MANIFEST.json.
 *
 * Build: see oracle/Makefile.  MUST be compiled with -ffp-contract=off and
 * without -ffast-math: the reference's float32 operation order is the contract.
 */
#ifndef JPEG_ORACLE_H
#define JPEG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* decode.swift:1289-1298  JPEG.Table.Quantization.z(k:h:) */
int orc_zigzag(int k, int h);

/* decode.swift:3984-4017  Spectral.Plane.modulate(quanta:scale:)
 * q_zz: 64 quanta in zigzag order; out[8*h + k]. */
void orc_modulate(const uint16_t q_zz[64], float scale, float out[64]);

/* decode.swift:4020-4133  Spectral.Plane.idct(quanta:precision:)
 * coef: [uy][ux][64] zigzag int16; out: [(8*uy)][(8*ux)] uint16.
 * rows [by0, by1) of blocks only (for banded multi-thread timing). */
void orc_idct_plane_rows(const int16_t *coef, int ux, int uy,
                         const uint16_t q_zz[64], int precision,
                         uint16_t *out, int by0, int by1);
void orc_idct_plane(const int16_t *coef, int ux, int uy,
                    const uint16_t q_zz[64], int precision, uint16_t *out);

/* decode.swift:4182-4276  Planar.interleaved(cosite:)
 * planes[p]: [(8*uy[p])][(8*ux[p])] uint16; (sx, sy) = layout.scale
 * (decode.swift:2181-2190); out: [H][W][count] uint16.
 * rows [y0, y1) of the output only. */
void orc_interleave_rows(const uint16_t *const *planes, const int *ux,
                         const int *uy, const int *fx, const int *fy,
                         int count, int sx, int sy, int W, int H, int cosited,
                         uint16_t *out, int y0, int y1);
void orc_interleave(const uint16_t *const *planes, const int *ux,
                    const int *uy, const int *fx, const int *fy, int count,
                    int sx, int sy, int W, int H, int cosited, uint16_t *out);

/* jpeg.swift:551-572 RGB.unpack, :441-453 YCbCr.rgb, :343-354 Common.clamp
 * in: npx*ncomp interleaved uint16 (ncomp 1 or 3); out: npx*3 uint8 (r,g,b). */
void orc_unpack_rgb8(const uint16_t *in, size_t npx, int ncomp, uint8_t *out);
/* jpeg.swift:493-514 YCbCr.unpack; out: npx*3 uint8 (y,cb,cr). */
void orc_unpack_ycc8(const uint16_t *in, size_t npx, int ncomp, uint8_t *out);

/* jpeg.swift:584-599 RGB.pack, :463-478 RGB.ycc
 * in: npx*3 uint8 (r,g,b); out: npx*ncomp uint16 interleaved. */
void orc_pack_rgb8(const uint8_t *in, size_t npx, int ncomp, uint16_t *out);
/* jpeg.swift:527-539 YCbCr.pack; in: npx*3 uint8 (y,cb,cr). */
void orc_pack_ycc8(const uint8_t *in, size_t npx, int ncomp, uint16_t *out);

/* encode.swift:389-425 Rectangular.decomposed() for ONE plane p.
 * in: [H][W][count]; out: [(8*uy)][(8*ux)] where
 * ux = ceil(W*fx / (8*sx)), uy = ceil(H*fy / (8*sy))  (decode.swift:2606-2616) */
void orc_decompose_plane(const uint16_t *in, int W, int H, int count, int p,
                         int fx, int fy, int sx, int sy, int ux, int uy,
                         uint16_t *out);

/* encode.swift:80-99, 104-248  Spectral.Plane.fdct(_:quanta:precision:)
 * plane: [(8*uy)][(8*ux)] uint16; coef: [uy][ux][64] zigzag int16. */
void orc_fdct_plane_rows(const uint16_t *plane, int ux, int uy,
                         const uint16_t q_zz[64], int precision,
                         int16_t *coef, int by0, int by1);
void orc_fdct_plane(const uint16_t *plane, int ux, int uy,
                    const uint16_t q_zz[64], int precision, int16_t *coef);

/* encode.swift:260-333 CompressionLevel.quanta (host constant tables; used by
 * the encode golden tests to rebuild the tables examples/encode-basic used).
 * kind 0 = luminance, 1 = chrominance; out zigzag order. */
void orc_compression_quanta(int kind, double level, uint16_t out_zz[64]);

#ifdef __cplusplus
}
#endif
#endif
