// kernels.hpp -- internal launch interface between the C ABI (capi.hip) and the gfx950
// kernels (kernels_*.hip).  Everything here is device-resident and asynchronous on `stream`.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/jpeg_amd.h"

namespace jpeg_amd {

// Streaming output (written once, never re-read by the launch chain): the `nt` hint keeps it
// from displacing data that IS re-read (chroma intermediates, tables) in L2 and the Infinity
// Cache.  Only for instructions that write WHOLE 128-byte lines: on accesses that touch a line
// 16 bytes at a time (one block per work-item loads / stores) `nt` defeats the merging of the
// pieces and halves the throughput (measured: IDCT-only 184 -> 351 us, encode 38 -> 71 us).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_nt16(void *p, const uint4 &v)
{
    __builtin_nontemporal_store(u32x4_t{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4_t *>(p));
}

// Per-plane description of a batch of identically laid out images.
// image i of plane p lives at ptr[p] + i * stride[p] (elements).
struct PlaneSet {
    const void *ptr[JPEG_AMD_MAX_PLANES];
    size_t      stride[JPEG_AMD_MAX_PLANES];
};
struct PlaneSetMut {
    void  *ptr[JPEG_AMD_MAX_PLANES];
    size_t stride[JPEG_AMD_MAX_PLANES];
};

// Quantisation tables in HBM: uint16 [n_images or 1][ntables][64] zigzag.
struct QuantaRef {
    const uint16_t *d_quanta;
    size_t          image_stride;  // uint16 elements between image table sets (0 = shared)
};

enum class PixelKind : int { Rect16 = 0, YCC8 = 1, RGB8 = 2 };

// ---- decode -------------------------------------------------------------------------
// a3..a7: dequantise + IDCT of one plane (batch of n_images), output uint16 or uint8 samples.
hipError_t launch_idct_plane(hipStream_t stream, int n_images, const int16_t *d_coef,
                             size_t coef_stride, QuantaRef q, int qi, int ux, int uy,
                             int precision, void *d_plane, size_t plane_stride,
                             bool out_u8);

// Sparse coefficients (entropy.cpp, jpeg_amd_jpeg_decode_sparse) -> the planes.  d_skip: optional per-image flags, nonzero =
// leave that image's planes alone.  d_packed (optional): the images' records [descriptors][entries] lie packed in d_desc, image i's
// at element d_packed[i] (the strides and d_entries are then not used).
hipError_t launch_expand_sparse(hipStream_t stream, int n_images, const jpeg_amd_layout &layout, const uint32_t *d_desc,
                                size_t desc_stride, const uint32_t *d_entries, size_t entries_stride, const uint8_t *d_skip,
                                const PlaneSetMut &coef, const uint64_t *d_packed = nullptr);

// ... and the planes -> sparse coefficients (for jpeg_amd_jpeg_encode_sparse).  d_cursor: one uint32 per image, on return the
// number of entries the image has; larger than `capacity`: the image did not fit and its entries are not valid.
hipError_t launch_sparsify(hipStream_t stream, int n_images, const jpeg_amd_layout &layout, const PlaneSet &coef, uint32_t *d_desc,
                           size_t desc_stride, uint32_t *d_entries, size_t entries_stride, uint32_t capacity, uint32_t *d_cursor);

// a9 (+ a11/a12): upsample + interleave, written as Rectangular uint16, or colour-converted
// straight to YCbCr / RGB bytes.  Planes are uint16 (or uint8 when planes_u8).
hipError_t launch_planar_to_pixels(hipStream_t stream, int n_images,
                                   const jpeg_amd_layout &layout, const PlaneSet &planes,
                                   bool planes_u8, bool cosited, PixelKind kind,
                                   void *d_out, size_t out_stride_bytes);

// ... and back: Rectangular -> Spectral in one launch (decomposed() + fdct(quanta:), encode.swift:389-425, 199-248), same layouts.
bool       generic_encode_supported(const jpeg_amd_layout &layout);
hipError_t launch_generic_encode(hipStream_t stream, int n_images, const jpeg_amd_layout &layout, const uint16_t *d_rect,
                                 size_t rect_stride, QuantaRef q, const PlaneSetMut &coef);

// a10..a12: Rectangular.unpack(as:) for YCbCr / RGB.
hipError_t launch_unpack(hipStream_t stream, const uint16_t *d_rect, size_t npixels,
                         int nplanes, jpeg_amd_color color, uint8_t *d_pixels);

// Fused Spectral -> YCbCr / RGB bytes (kernels_fused.hip): 8-bit y8 images, and ycc8 images
// with full-factor luma and 1x / 2x subsampled chroma, centred upsampling.
bool       fused_decode_supported(const jpeg_amd_layout &layout, bool cosited);
// No intermediate in HBM for any layout.  `d_walk_counters`: a device dword that the caller keeps for the stream
// (jpeg_amd_ctx owns one): the 4:2:0 walk of a long call hands its stacks out through it (zeroed in front of the launch,
// on the stream); nullptr: always the static walk.
hipError_t launch_fused_decode(hipStream_t stream, int n_images, const jpeg_amd_layout &layout,
                               const PlaneSet &coef, QuantaRef q, bool rgb, uint32_t *d_walk_counters,
                               uint8_t *d_pixels, size_t pixel_stride);

// ycc8 4:2:0 in ONE launch with no chroma intermediate in HBM (kernels_quad.hip): the waves of a workgroup decode a
// stack of vertically adjacent strips together and share one chroma tile in LDS.  Any image size; launch_fused_decode
// sends every 4:2:0 call here.
bool       quad_decode_supported(const jpeg_amd_layout &layout);
hipError_t launch_quad_decode(hipStream_t stream, int n_images, const jpeg_amd_layout &layout,
                              const PlaneSet &coef, QuantaRef q, bool rgb, uint32_t *d_walk_counters,
                              uint8_t *d_pixels, size_t pixel_stride);

// Spectral -> Rectangular (uint16 [H][W][count]) in one launch for the JPEG.Format plug-in path (kernels_generic.hip): any
// precision 1 .. 16, 1 .. 4 planes, centred or cosited, every plane at the image's scale or at half of it per axis.
bool       generic_fused_supported(const jpeg_amd_layout &layout);
// `d_walk_counter`: TWO device dwords the caller keeps for the stream, zero when handed over (calls of several rounds draw their tiles
// through them and leave them zero; nullptr: planned)
hipError_t launch_generic_fused(hipStream_t stream, int n_images, const jpeg_amd_layout &layout, const PlaneSet &coef,
                                QuantaRef q, bool cosited, uint32_t *d_walk_counter, uint16_t *d_rect, size_t rect_stride);

// ---- encode -------------------------------------------------------------------------
// a13: Rectangular.pack
hipError_t launch_pack(hipStream_t stream, const uint8_t *d_pixels, size_t npixels,
                       int nplanes, jpeg_amd_color color, uint16_t *d_rect);

// a14: Rectangular.decomposed() for every plane; input is Rectangular uint16, or pixel
// bytes (RGB8 / YCC8) colour-converted on the fly.
hipError_t launch_decompose(hipStream_t stream, int n_images, const jpeg_amd_layout &layout,
                            const void *d_in, size_t in_stride_bytes, PixelKind in_kind,
                            const PlaneSetMut &planes);

// Fused pixels -> Spectral (kernels_encode.hip): 8-bit y8 images, and ycc8 images with
// full-factor luma and 1x / 2x subsampled chroma.
bool       fused_encode_supported(const jpeg_amd_layout &layout);
hipError_t launch_fused_encode(hipStream_t stream, int n_images, const jpeg_amd_layout &layout,
                               const uint8_t *d_pixels, size_t pixel_stride, bool rgb, QuantaRef q,
                               const PlaneSetMut &coef);

// a15..a17: FDCT + quantise of one plane.
hipError_t launch_fdct_plane(hipStream_t stream, int n_images, const uint16_t *d_plane,
                             size_t plane_stride, QuantaRef q, int qi, int ux, int uy,
                             int precision, int16_t *d_coef, size_t coef_stride);

}  // namespace jpeg_amd
