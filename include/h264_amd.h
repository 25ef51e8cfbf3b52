/**
 * libh264_amd -- batched MI355X engine behind the libh264 C API (additive).
 *
 * The reference library is single threaded and handles one NAL unit per call
 * (src/h264_reader.c:117-162).  Its C API stays available unchanged in
 * <h264/h264.h>; this header adds the batched, device-resident form of the
 * same hot path so that thousands of independent slice NAL units are parsed
 * by one kernel launch (one slice per wavefront):
 *
 *   h264_amd_scan()            <- h264_find_nalu loop   src/h264_bitstream.c:87-184,
 *                                                        src/h264_reader.c:133-140
 *   h264_amd_decode_slices()   <- _h264_read_nalu for nal_unit_type 1/5:
 *                                 slice_header           src/h264_syntax.h:1239-1387
 *                                 slice_data / MB loop   src/h264_syntax_slice_data.h:701-811
 *                                 CAVLC residual         src/h264_syntax_slice_data.h:103-419,
 *                                                        src/h264_slice_data.c:839-1416
 *   h264_amd_write_slices()    <- h264_write_nalu for slices
 *                                                        src/h264_writer.c:240-243,
 *                                                        src/h264_syntax_slice_data.h:813-830
 *   h264_amd_encode_synth_slices() <- h264_write_grey_i_slice /
 *                                 h264_write_skipped_p_slice (CAVLC + CABAC)
 *                                                        src/h264_writer.c:49-309
 *   h264_amd_cabac_init_states() <- h264_cabac_init_states
 *                                                        src/h264_cabac_ctx_tables.c:4441-4468
 *
 * Plain C ABI: pointers, sizes, POD structs.  Pointers named d_* are DEVICE
 * pointers (hipMalloc / torch CUDA tensors), everything else is host memory.
 * All launches are asynchronous on the engine's HIP stream unless stated.
 * Every function returns 0 or a negative errno; -ENODEV when no usable GPU or
 * kernel image is present (there is no CPU fallback).
 */
#ifndef H264_AMD_H
#define H264_AMD_H

#include <h264/h264.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------ */
/* Record formats (shared by the kernels, the host replay and the oracle)    */
/* ------------------------------------------------------------------------ */

/* One NAL unit found by the scan: payload is buf[start, end). */
struct h264_amd_nal {
	uint64_t start;
	uint64_t end;
};

/* One slice NAL unit to decode: bytes d_buf[offset, offset+size) hold the NAL
 * unit WITHOUT start code and WITH emulation-prevention bytes, exactly what
 * h264_reader_parse_nalu() takes.  `epoch` selects the parameter-set snapshot
 * that was current at this NAL's position in the stream. */
struct h264_amd_slice_desc {
	uint64_t offset;
	uint32_t size;
	uint32_t epoch;
};

/* Macroblock output tiers */
enum h264_amd_tier {
	H264_AMD_TIER_NONE = 0, /* per-slice result + checksum only */
	H264_AMD_TIER_T0 = 1, /* 4 B/MB: first dword of the summary */
	H264_AMD_TIER_T1 = 2, /* 64 B/MB: struct h264_amd_mb_summary */
	H264_AMD_TIER_T2 = 3, /* full syntax, h264_amd_t2_stride() B/MB */
};

/* h264_amd_mb_summary.flags */
#define H264_AMD_MB_SKIPPED 0x01
#define H264_AMD_MB_TRANSFORM_8X8 0x02
#define H264_AMD_MB_FIELD 0x04

/* 64-byte macroblock summary = bytes [0,64) of a T2 record.
 * Mirrors the fields of the reference's private struct h264_macroblock
 * (src/h264_macroblock.h:105-167) that are not coefficient arrays. */
struct h264_amd_mb_summary {
	uint8_t mb_type; /* enum h264_mb_type */
	uint8_t cbp; /* CodedBlockPatternLuma | CodedBlockPatternChroma << 4 */
	uint8_t flags; /* H264_AMD_MB_* */
	uint8_t raw_mb_type; /* mb_type as coded, low 8 bits */
	int16_t mb_qp_delta;
	uint8_t intra_chroma_pred_mode;
	uint8_t intra16x16_pred_mode;
	/* intra: 16 (Intra4x4) or 4 (Intra8x8) nibbles, low nibble first,
	 * nibble = pred_mode + 1 (0 = "predicted", the reference's -1);
	 * inter with 4 partitions: bytes 0..3 = raw sub_mb_type[0..3] */
	uint8_t pred[8];
	/* TotalCoeff per 4x4 block as the reference keeps it for nC
	 * derivation: [comp*16 + blkIdx], comp 0 Y, 1 Cb, 2 Cr
	 * (src/h264_slice_data.c:1211-1232,1353-1358) */
	uint8_t nz_coeff[48];
};

/* T2 record = 256-byte head + coefficient area of int16:
 *   [  0, 64)  struct h264_amd_mb_summary
 *   [ 64, 68)  ref_idx_l0[4]      [ 68, 72)  ref_idx_l1[4]
 *   [ 72,128)  zero
 *   [128,192)  mvd_l0[4][4][2] int16     [192,256)  mvd_l1[4][4][2] int16
 *   [256, ...) coefficient area, int16 units:
 *      luma  [0,256):  I_16x16: Intra16x16DCLevel[16] then
 *                      Intra16x16ACLevel[16][15]; otherwise LumaLevel4x4[16][16]
 *      ChromaArrayType 1/2 (n = 4 or 8 chroma blocks per component):
 *              ChromaDCLevel[2][n], then ChromaACLevel[2][n][15]
 *      ChromaArrayType 3: Cb [256,512), Cr [512,768) laid out like luma
 *      I_PCM: pcm_sample_luma[256] then pcm_sample_chroma[2][MbWidthC*MbHeightC]
 *              as uint16
 * Coefficient counts: ChromaArrayType 0: 256, 1: 384, 2: 512, 3: 768. */
#define H264_AMD_T2_HEAD 256u
static inline uint32_t h264_amd_t2_coefs(uint32_t chroma_array_type)
{
	return chroma_array_type == 0   ? 256u
	       : chroma_array_type == 1 ? 384u
	       : chroma_array_type == 2 ? 512u
					: 768u;
}
static inline uint32_t h264_amd_t2_stride(uint32_t chroma_array_type)
{
	return H264_AMD_T2_HEAD + 2u * h264_amd_t2_coefs(chroma_array_type);
}

/* h264_amd_slice_result.flags */
#define H264_AMD_SR_HEADER_OK 0x0001 /* slice header parsed */
#define H264_AMD_SR_DATA_PARSED 0x0002 /* slice_data walked (CAVLC) */
#define H264_AMD_SR_CABAC 0x0004 /* CABAC slice: data kept raw */

/* Engine-specific status values besides negative errno */
#define H264_AMD_E_UNSUPPORTED (-ENOSYS) /* CABAC beyond the subset; absurd widths; slice groups on pictures beyond level 6.2 */
#define H264_AMD_E_MB_OVERFLOW (-EOVERFLOW) /* more MBs than mb_capacity */
#define H264_AMD_E_BAD_DESC (-EFAULT) /* descriptor outside d_buf_len */
/* Most macroblocks one slice may deliver (64 pictures of 4096 x 2304): where the
 * drop-in reader stops growing a slice's record area, and where the oracle
 * stops too.  Only a damaged mb_skip_run gets anywhere near.  Under slice
 * groups the macroblock ADDRESS runs ahead of the count: a slice also ends with
 * H264_AMD_E_MB_OVERFLOW when its next address lies this far behind its first. */
#define H264_AMD_MAX_SLICE_MBS (1u << 22)

/* Widest picture whose slice data is walked, in macroblocks (16896 pixels).
 * Level 6.2, the highest there is, allows PicWidthInMbs <= sqrt(8 * MaxFS) =
 * 1055; the reference itself accepts any 32-bit width (and divides by a width
 * of 0, src/h264_macroblock.c:318-330).  Slices of wider pictures, and of
 * pictures of width 0, end with H264_AMD_E_UNSUPPORTED in product and oracle
 * alike; their headers are still parsed.  Bounds the per-wave row context. */
#define H264_AMD_MAX_PIC_WIDTH_MBS 1056u
/* The same for CABAC slice data (H264_AMD_READER_FLAGS_CABAC): 14 080 pixels.
 * A macroblock column keeps 40 bytes of row context there (coded_block_flag,
 * |mvd|, ref_idx of its bottom row) against 16 for CAVLC, and four slices'
 * rows share one compute unit's LDS.  Wider pictures (level 6.2 would allow
 * 1055 macroblocks; 8K is 512): H264_AMD_E_UNSUPPORTED behind the header, in
 * product and oracle alike, not a failed launch. */
#define H264_AMD_MAX_CABAC_WIDTH_MBS 880u

/* Slice groups (FMO): the macroblock-to-slice-group map of a slice is built
 * for pictures of up to H264_AMD_FMO_MAX_PIC_MBS macroblocks (what level 6.2
 * allows) and, for the box-out type (slice_group_map_type 3), of up to
 * H264_AMD_FMO_MAX_BOX_OUT_UNITS map units; the slice data of larger pictures
 * -- only a damaged SPS asks for them -- is not walked:
 * H264_AMD_E_UNSUPPORTED in product and oracle, header still parsed.  (The
 * reference allocates and fills a map of any size, up to 2^32 entries.)
 * Type 0 run lengths are summed in 64 bits (the reference's 32-bit sum wraps
 * for run_length_minus1 = 2^32 - 2 and its loop need not end). */
#define H264_AMD_FMO_MAX_PIC_MBS 139264u
#define H264_AMD_FMO_MAX_BOX_OUT_UNITS 32768u
/* The box-out walk (8.2.2.4) steps over units it has filled already once the
 * picture's edge clips the box; in a picture of extreme aspect ratio nearly
 * all steps are of that kind (1056 x 6 macroblocks: 177 steps per unit, n^2 at
 * worst).  The walk ends after this many steps, units not reached stay in
 * slice group 1 -- enough for every aspect ratio up to 14 : 1; product and
 * oracle alike (found by the hostile-parameter fuzz, round 3). */
#define H264_AMD_FMO_BOX_OUT_STEPS(n) (8ull * (n) + 64ull)

/* Per-slice outcome, 64 bytes.  status mirrors what _h264_read_nalu would
 * have returned for this NAL (0, -EIO, -EINVAL ...). */
struct h264_amd_slice_result {
	int32_t status;
	uint32_t flags; /* H264_AMD_SR_* */
	uint32_t nal_ref_idc;
	uint32_t nal_unit_type;
	int32_t slice_type; /* enum h264_slice_type, -1 if header failed */
	uint32_t first_mb_in_slice;
	uint32_t mb_count; /* macroblocks delivered (skipped + coded) */
	int32_t slice_qp; /* SliceQPLuma */
	/* ctx->slice.hdr_len: header length in RAW bits, NAL header byte and
	 * escape bytes included (src/h264_syntax.h:1383) */
	uint32_t hdr_len_bits;
	/* ctx->slice.rawdata (src/h264_syntax_slice_data.h:803-806): offset and
	 * length of the raw slice-data bytes inside the NAL unit, plus the
	 * <= 7 leftover bits of the header's last byte */
	uint32_t rawdata_off;
	uint32_t rawdata_len;
	uint8_t partial;
	uint8_t partialbits;
	uint16_t chroma_array_type;
	/* RBSP bit position reached (escape bytes not counted); diagnostic */
	uint32_t end_bit;
	uint32_t pic_width_in_mbs;
	/* order-independent checksum over the T2 byte image of every delivered
	 * macroblock, see h264_amd_checksum_term() */
	uint64_t checksum;
};

/* Checksum: sum over macroblock index m and dword index j (j < 512) of the T2
 * record of term(m * 512 + j, dword j of the record, little endian: 16-bit word
 * 2j in its low half).  Zero dwords contribute nothing, so the sum can be taken
 * over any partition of the record, in any order.
 * Low half: the two words weighted by odd 16-bit position weights that are
 * affine in the index (a kernel steps them with one packed add; two words cost
 * ONE v_dot2_u32_u16); high half: the sum of the two squares times an odd
 * 32-bit position weight.  All arithmetic modulo 2^32. */
#define H264_AMD_CHECKSUM_KP 0x9E35u /* odd */
#define H264_AMD_CHECKSUM_KB 0xC2B2AE3Du /* odd */
static inline uint64_t h264_amd_checksum_term(uint32_t g, uint32_t dword)
{
	uint32_t w0 = dword & 0xffffu, w1 = dword >> 16;
	uint32_t p0 = ((4u * g + 1u) * H264_AMD_CHECKSUM_KP) & 0xffffu;
	uint32_t p1 = ((4u * g + 3u) * H264_AMD_CHECKSUM_KP) & 0xffffu;
	uint32_t b = (2u * g + 1u) * H264_AMD_CHECKSUM_KB;
	uint32_t lo = w0 * p0 + w1 * p1;
	uint32_t hi = (w0 * w0 + w1 * w1) * b;
	return ((uint64_t)hi << 32) | lo;
}
/* the two 32-bit halves are summed independently (no carry between them) */
static inline uint64_t h264_amd_checksum_add(uint64_t acc, uint64_t term)
{
	uint32_t lo = (uint32_t)acc + (uint32_t)term;
	uint32_t hi = (uint32_t)(acc >> 32) + (uint32_t)(term >> 32);
	return ((uint64_t)hi << 32) | lo;
}

/* ------------------------------------------------------------------------ */
/* Engine                                                                     */
/* ------------------------------------------------------------------------ */

struct h264_amd_engine;
struct h264_amd_params;

/* device: HIP device ordinal.  hip_stream: a hipStream_t to launch on, or NULL
 * for an engine-owned stream.  -ENODEV if the device or the gfx950 code
 * object is unusable. */
H264_API int h264_amd_engine_new(int device,
				 void *hip_stream,
				 struct h264_amd_engine **ret);
H264_API int h264_amd_engine_destroy(struct h264_amd_engine *eng);
H264_API int h264_amd_engine_sync(struct h264_amd_engine *eng);
H264_API void *h264_amd_engine_stream(struct h264_amd_engine *eng);
/* 1 if a GPU usable by this library is present (counts devices only) */
H264_API int h264_amd_available(void);

/* Plain device memory helpers so that C callers need no HIP headers. */
H264_API int h264_amd_malloc(struct h264_amd_engine *eng, size_t size, void **d_ptr);
H264_API int h264_amd_free(struct h264_amd_engine *eng, void *d_ptr);
H264_API int h264_amd_upload(struct h264_amd_engine *eng,
			     void *d_dst,
			     const void *src,
			     size_t size);
H264_API int h264_amd_download(struct h264_amd_engine *eng,
			       void *dst,
			       const void *d_src,
			       size_t size);
/* Page-locked host memory and a download that does not hold the caller (nor
 * the engine's stream): the copy starts behind everything queued on the
 * engine's stream so far and runs on a stream of its own;
 * h264_amd_download_wait() returns when all such copies have landed.  What the
 * drop-in reader fetches record pages with while it replays the callbacks of
 * the page before (the reference has no counterpart: its parse and its
 * callbacks are one loop, src/h264_reader.c:117-143). */
H264_API int h264_amd_host_alloc(struct h264_amd_engine *eng, size_t size, void **ptr);
H264_API int h264_amd_host_free(struct h264_amd_engine *eng, void *ptr);
H264_API int h264_amd_download_async(struct h264_amd_engine *eng, void *dst,
				     const void *d_src, size_t size);
H264_API int h264_amd_download_wait(struct h264_amd_engine *eng);
/* device to device, on the engine's stream */
H264_API int h264_amd_copy(struct h264_amd_engine *e, void *d_dst,
			   const void *d_src, size_t size);
H264_API int h264_amd_memset(struct h264_amd_engine *eng,
			     void *d_dst,
			     int value,
			     size_t size);

/* ---- parameter-set snapshots ------------------------------------------- */
/* A slice binds to the SPS/PPS *versions* active at its stream position
 * (h264_ctx_set_pps overwrites table entries in place, src/h264_ctx.c:334-355).
 * The params object records every version and, per epoch, which version each
 * of the 32 SPS / 256 PPS ids pointed to. */
H264_API int h264_amd_params_new(struct h264_amd_params **ret);
H264_API int h264_amd_params_destroy(struct h264_amd_params *p);
H264_API int h264_amd_params_reset(struct h264_amd_params *p);
/* return the epoch that is current AFTER the update (>= 0) or -errno */
H264_API int h264_amd_params_put_sps(struct h264_amd_params *p,
				     const struct h264_sps *sps);
H264_API int h264_amd_params_put_pps(struct h264_amd_params *p,
				     const struct h264_pps *pps);
H264_API int h264_amd_params_epoch(const struct h264_amd_params *p);
/* the SPS / PPS an id names in the current epoch (borrowed) or NULL */
H264_API const struct h264_sps *
h264_amd_params_sps(const struct h264_amd_params *p, uint32_t sps_id);
H264_API const struct h264_pps *
h264_amd_params_pps(const struct h264_amd_params *p, uint32_t pps_id);
/* 1 if any PPS version seen so far has num_slice_groups_minus1 > 0 */
H264_API int h264_amd_params_has_slice_groups(const struct h264_amd_params *p);
/* largest PicSizeInMbs over every SPS version seen: a safe mb_capacity */
H264_API uint32_t h264_amd_params_max_mbs(const struct h264_amd_params *p);

/* ---- Annex-B scan -------------------------------------------------------- */
/* Finds every NAL unit of d_buf[0,len) with the reference's exact delimiting
 * rules.  Writes up to `cap` entries, sorted by
 * start, to d_nals; *count receives the total number found (may exceed cap);
 * *consumed receives what h264_reader_parse would report in *off.
 * Synchronous (returns after the table is complete). */
H264_API int h264_amd_scan(struct h264_amd_engine *eng,
			   const uint8_t *d_buf,
			   size_t len,
			   struct h264_amd_nal *d_nals,
			   size_t cap,
			   size_t *count,
			   size_t *consumed);

/* h264_byte_stream_to_avcc (ref src/h264.c:208-246) on a device buffer, in
 * place: every four-byte start code 00 00 00 01 is replaced by the big-endian
 * length of the NAL unit behind it (up to the next four-byte start code or the
 * end of the buffer); three-byte start codes are payload, a start code in the
 * buffer's last four bytes is left alone, bytes before the first start code
 * are not touched -- exactly the reference's loop.  d_buf 16-byte aligned.
 * Synchronous.  (The opposite direction is a chain of lengths, each found
 * only through the one before: it stays on the host, h264_avcc_to_byte_stream.) */
H264_API int h264_amd_byte_stream_to_avcc(struct h264_amd_engine *eng,
					  uint8_t *d_buf,
					  size_t len);

/* Head bytes of every NAL unit of a scan, as one dense table: d_heads[i][j] =
 * d_buf[nal[i].start + j] for j < min(head_bytes, nal length), 0 beyond.  What
 * the host control plane of a device-resident pipeline needs to see -- the NAL
 * header byte (nal_ref_idc, nal_unit_type: src/h264_syntax.h:1434-1443) of
 * every unit and the whole of short ones (parameter sets, AUD) -- in one
 * device-to-host copy instead of one per NAL unit.  head_bytes: 1..256. */
H264_API int h264_amd_gather_nal_heads(struct h264_amd_engine *eng,
				       const uint8_t *d_buf,
				       const struct h264_amd_nal *d_nals,
				       size_t n,
				       uint32_t head_bytes,
				       uint8_t *d_heads);

/* ---- slice decode -------------------------------------------------------- */
/* reader_flags bit beside H264_READER_FLAGS_SLICE_DATA: also walk the slice
 * data of CABAC slices.  The reference has no CABAC decoder (it returns before
 * CABAC slice data, src/h264_syntax_slice_data.h:715-717), so this is an
 * extension and off by default; the drop-in library never sets it.  The
 * decoder follows the standard (9.3: binarisations, context index rules,
 * arithmetic decoding engine) for I / SI / P / SP / B slices of frame and field
 * pictures, all chroma formats, 8x8 transform, I_PCM -- spec-only and therefore
 * "parity unpinned", except for what the reference's own writers emit
 * (h264_write_grey_i_slice / h264_write_skipped_p_slice), which it decodes
 * exactly.  Context variables are initialised as the REFERENCE does it
 * (SliceQPY clipped to 1..51, src/h264_bac.c:216-230).  MBAFF frames and slice
 * groups with CABAC end with H264_AMD_E_UNSUPPORTED.  Records are the T2
 * records of CAVLC with these conventions: an 8x8 block's coefficient 4k + j is
 * coefficient k of the 4x4 array j of its quadrant (the interleaving CAVLC
 * uses, ref src/h264_syntax_slice_data.h:323-326); nz_coeff[] counts the
 * non-zero coefficients of each 4x4 array, DC blocks not counted; a level
 * beyond 16 bits wraps, and so does a motion vector difference, whose
 * neighbours' contexts see the magnitude of the 16-bit value as stored; unary
 * codes are bounded (mb_qp_delta, ref_idx: 255;
 * Exp-Golomb escapes: 24 prefix bits) and end the slice with -EIO beyond; so
 * does a start of the decoding engine with codIOffset 510 or 511 (9.3.1.2).
 * Needs d_headers. */
#define H264_AMD_READER_FLAGS_CABAC 0x100u
#define H264_AMD_READER_FLAGS_CABAC_SUBSET H264_AMD_READER_FLAGS_CABAC /* round-2 name */
/* With H264_AMD_READER_FLAGS_CABAC: decode 4:4:4 streams written by x264 before
 * its build 151 (the version is in the stream's user-data SEI).  Those builds
 * chose the coded_block_flag context of an 8x8 block next to a macroblock
 * WITHOUT the 8x8 transform as if the neighbour were not available, where
 * 9.3.3.1.1.9 says "transBlockN not available: condTermFlagN = 0"; a decoder
 * that follows the standard loses such a stream at the first macroblock of
 * that kind (libavcodec reproduces the old behaviour for these streams).  Off
 * by default: the standard's rule. */
#define H264_AMD_READER_FLAGS_CABAC_X264_PRE151 0x200u
/* The x264 build number out of a user-data SEI NAL unit ("x264 - core NNN"), 0
 * if there is none: a caller sets the flag above for ChromaArrayType 3 streams
 * with 0 < build < 151. */
H264_API uint32_t h264_amd_x264_build(const uint8_t *sei_nal, size_t len);
/* With H264_AMD_READER_FLAGS_CABAC: initialise the context variables with
 * SliceQPY clipped to 0..51 as 9.3.1.1 says, not to 1..51 as the reference's
 * encoder does.  The two differ at SliceQPY 0 only (and below, which no
 * conforming 8-bit stream reaches): a slice the reference's writers -- or this
 * library's -- made at SliceQPY 0 needs the default, a slice of any other
 * encoder at SliceQPY 0 (lossless x264, qpprime_y_zero_transform_bypass_flag)
 * needs this flag. */
#define H264_AMD_READER_FLAGS_CABAC_STD_QP 0x400u
struct h264_amd_decode_args {
	const uint8_t *d_buf; /* NAL bytes */
	const struct h264_amd_slice_desc *d_descs; /* [n] */
	uint32_t n;
	uint32_t reader_flags; /* H264_READER_FLAGS_SLICE_DATA or 0 */
	uint32_t tier; /* enum h264_amd_tier */
	/* macroblock records per slice; a slice with more macroblocks ends with
	 * H264_AMD_E_MB_OVERFLOW.  0 is only accepted without macroblock output
	 * (tier NONE or d_mb_out NULL) and then means "no limit other than
	 * H264_AMD_MAX_SLICE_MBS" */
	uint32_t mb_capacity;
	struct h264_amd_slice_result *d_results; /* [n] */
	struct h264_slice_header *d_headers; /* [n] byte images, or NULL */
	uint8_t *d_mb_out; /* n * mb_capacity * mb_stride bytes, or NULL */
	uint32_t mb_stride; /* bytes per MB record for the tier */
	uint32_t reserved;
	/* size of the allocation behind d_buf, or 0 if the caller vouches for its
	 * descriptors.  When given, a descriptor reaching past it costs that slice
	 * H264_AMD_E_BAD_DESC instead of a faulting kernel (a GPU memory fault
	 * can take down more than the process). */
	uint64_t d_buf_len;
};
H264_API int h264_amd_decode_slices(struct h264_amd_engine *eng,
				    const struct h264_amd_params *params,
				    const struct h264_amd_decode_args *args);

/* ---- slice write-back (round trip) -------------------------------------- */
struct h264_amd_write_args {
	const uint8_t *d_src; /* original NAL bytes (raw slice data source) */
	const struct h264_amd_slice_desc *d_descs; /* [n] */
	const struct h264_amd_slice_result *d_results; /* [n] from decode */
	const struct h264_slice_header *d_headers; /* [n] from decode */
	uint32_t n;
	uint32_t wflags; /* H264_AMD_W_* */
	uint8_t *d_out; /* output NAL bytes */
	const uint64_t *d_out_offsets; /* [n] where each NAL goes in d_out */
	uint32_t *d_out_sizes; /* [n] bytes written, or negative errno cast */
	uint32_t out_capacity; /* per-NAL capacity in bytes */
	uint32_t reserved2;
	/* optional [n]: header length in bits as the reference's writer
	 * records it in ctx->slice.hdr_len (src/h264_syntax.h:1380-1381).  When
	 * a NAL ends inside a byte (no slice data given) the pending bits are
	 * stored, left aligned, in the byte following the last whole one. */
	uint32_t *d_out_hdr_bits;
	uint64_t d_src_len; /* as d_buf_len of h264_amd_decode_args */
};
/* wflags: bits 0-1 = number of zero bytes (0..2) that precede the output in
 * the caller's bitstream (they take part in the 00 00 0x escape rule),
 * bit 8 = do not insert emulation-prevention bytes */
#define H264_AMD_W_PREV_ZEROS(n) ((n)&3u)
#define H264_AMD_W_NO_ESCAPE 0x100u
H264_API int h264_amd_write_slices(struct h264_amd_engine *eng,
				   const struct h264_amd_params *params,
				   const struct h264_amd_write_args *args);

/* ---- synthetic slices (CAVLC and CABAC) --------------------------------- */
enum h264_amd_synth_kind {
	H264_AMD_SYNTH_GREY_I = 0, /* h264_write_grey_i_slice */
	H264_AMD_SYNTH_SKIP_P = 1, /* h264_write_skipped_p_slice */
};
struct h264_amd_synth_desc {
	uint32_t kind; /* enum h264_amd_synth_kind */
	uint32_t epoch;
	uint32_t mb_count;
	uint32_t nal_ref_idc;
	uint32_t nal_unit_type;
	uint32_t reserved;
};
struct h264_amd_synth_args {
	const struct h264_amd_synth_desc *d_descs; /* [n] */
	const struct h264_slice_header *d_headers; /* [n] */
	uint32_t n;
	uint32_t out_capacity; /* per-NAL capacity in bytes */
	uint8_t *d_out;
	const uint64_t *d_out_offsets; /* [n] */
	int32_t *d_out_sizes; /* [n] bytes written or negative errno */
	uint32_t *d_out_hdr_bits; /* optional [n], see h264_amd_write_args */
	uint32_t wflags; /* H264_AMD_W_* */
	uint32_t reserved;
};
H264_API int h264_amd_encode_synth_slices(struct h264_amd_engine *eng,
					  const struct h264_amd_params *params,
					  const struct h264_amd_synth_args *args);

/* CABAC context initialisation for n (slice_type, cabac_init_idc, SliceQPLuma)
 * triples: d_states[n][1024][2] = {pStateIdx, valMPS}. */
H264_API int h264_amd_cabac_init_states(struct h264_amd_engine *eng,
					const int32_t *d_triples, /* [n][3] */
					uint32_t n,
					uint8_t *d_states);

/* Kernel timing: duration in ms of the most recent launch of the named
 * kernel family ("scan" = the marking kernel that reads the stream,
 * "scan_all" = marking + prefix + placement, "decode", "write", "synth",
 * "cabac", "avcc" = the whole byte-stream-to-AVCC conversion),
 * measured with HIP events on the engine stream.  Synchronises. */
H264_API int h264_amd_last_kernel_ms(struct h264_amd_engine *eng,
				     const char *family,
				     float *ms);

/* Slice groups (FMO): h264_amd_decode_slices() builds the macroblock-to-slice-
 * group map of 8.2.2 per slice (the kernel needs it to find a slice's next
 * macroblock address, ref src/h264_fmo.c:307-319).  Macroblock records are in
 * slice order and carry no address; a caller that needs the addresses of a
 * slice under a PPS with slice groups fetches that slice's map -- one byte per
 * macroblock address of the picture, `size` <= PicSizeInMbs -- and walks it:
 * the next address of the same group above the current one.  `index`: the
 * slice's position in the LAST decode launch of this engine.  -ERANGE when
 * that launch built no map for it. */
H264_API int h264_amd_slice_group_map(struct h264_amd_engine *e, uint32_t index,
				      uint8_t *dst, size_t size);
/* The maps of `count` slices of the last launch from position `first` on,
 * `size` bytes of each into rows of `dst_pitch` bytes, in one copy; and the
 * room the last launch gave each map (0: it built none).  The maps live until
 * the engine's next decode launch: a caller that hands control to code that may
 * launch again (the drop-in reader's callbacks) fetches them first. */
H264_API int h264_amd_slice_group_maps(struct h264_amd_engine *e, uint32_t first,
				       uint32_t count, uint8_t *dst,
				       size_t dst_pitch, size_t size);
H264_API size_t h264_amd_slice_group_map_pitch(const struct h264_amd_engine *e);

/* ---- engine group: one batch, several devices --------------------------------- */
/* Slices are independent entropy-coded units: a batch shards
 * by slice.  A group is N engines of ONE process, one per device; a decode call
 * takes the same arguments as h264_amd_decode_slices() with every buffer on the
 * FIRST member's device and
 *   1. plans on the host: the members that get work -- min(N, ceil(n / 1024)), a
 *      batch that fills less than a quarter of one chip is not scattered -- and a
 *      size-balanced assignment, longest slice first to the least loaded member;
 *   2. packs each other member's slices into one contiguous piece on the first
 *      device (gather kernel) and sends it with ONE device-to-device copy per
 *      member -- point to point: on xGMI every peer has its own link;
 *   3. decodes on every member at once (the first member reads the caller's
 *      buffer in place);
 *   4. brings the result records (and header images / macroblock records when
 *      asked for) back and scatters them into the caller's arrays in the
 *      caller's order.
 * The reference has no counterpart (it is single threaded, one NAL unit per
 * call, src/h264_reader.c:117-162).  libh264.so uses a group when
 * H264_AMD_DEVICES=0,1,... names more than one device.  Slice groups (FMO) maps
 * stay with the member that built them: h264_amd_slice_group_map() is not
 * available through a group; a caller that needs them decodes on one engine. */
struct h264_amd_group;

/* The device layer a group runs on.  h264_amd_group_new() uses the library's
 * own (HIP: an engine per device, hipMemcpyPeerAsync between them); a test
 * substitutes host memory and its own decoder through
 * h264_amd_group_new_with_ops().  `member` is the index in the group; every call
 * returns 0 or -errno; copies and launches may be asynchronous on the member's
 * stream, sync() waits for all of that member's work. */
struct h264_amd_group_ops {
	void *ctx;
	int (*alloc)(void *ctx, uint32_t member, size_t size, void **d_ptr);
	int (*free)(void *ctx, uint32_t member, void *d_ptr);
	int (*upload)(void *ctx, uint32_t member, void *d_dst, const void *src,
		      size_t size);
	int (*download)(void *ctx, uint32_t member, void *dst, const void *d_src,
			size_t size);
	/* d_dst on dst_member <- d_src on src_member */
	int (*copy)(void *ctx, uint32_t dst_member, void *d_dst, uint32_t src_member,
		    const void *d_src, size_t size);
	/* on `member`: for i < n  d_dst[d_dst_off[i] ..] = d_src[d_src_off[i] .. +
	 * d_sizes[i]) */
	int (*pack)(void *ctx, uint32_t member, uint8_t *d_dst, const uint8_t *d_src,
		    const uint64_t *d_src_off, const uint64_t *d_dst_off,
		    const uint32_t *d_sizes, uint32_t n);
	/* on `member`: for j < n  row d_index[j] of d_dst = row j of d_src; rows
	 * of row_bytes bytes, a multiple of 16 */
	int (*scatter_rows)(void *ctx, uint32_t member, uint8_t *d_dst,
			    const uint8_t *d_src, const uint32_t *d_index, uint32_t n,
			    size_t row_bytes);
	int (*decode)(void *ctx, uint32_t member, const struct h264_amd_params *params,
		      const struct h264_amd_decode_args *args);
	int (*sync)(void *ctx, uint32_t member);
};

/* devices: n HIP ordinals (the first one holds the caller's buffers) */
H264_API int h264_amd_group_new(const int *devices, uint32_t n,
				struct h264_amd_group **ret);
H264_API int h264_amd_group_new_with_ops(const struct h264_amd_group_ops *ops,
					 uint32_t n, struct h264_amd_group **ret);
H264_API int h264_amd_group_destroy(struct h264_amd_group *g);
H264_API uint32_t h264_amd_group_size(const struct h264_amd_group *g);
/* member i's engine (NULL for a group on foreign ops) */
H264_API struct h264_amd_engine *h264_amd_group_engine(struct h264_amd_group *g,
						       uint32_t i);

/* The plan alone (host arithmetic, no device): members used and, per slice, the
 * member that gets it.  sizes[n] in bytes; owner[n] out. */
H264_API uint32_t h264_amd_group_plan(const uint32_t *sizes, uint32_t n,
				      uint32_t members, uint32_t *owner);
/* the same with the threshold spelled out: members used = min(members,
 * ceil(n / min_per_member)); h264_amd_group_plan() uses 1024 */
H264_API uint32_t h264_amd_group_plan_ex(const uint32_t *sizes, uint32_t n,
					 uint32_t members, uint32_t min_per_member,
					 uint32_t *owner);
/* The group's gather kernel on one engine, for a caller that transports the
 * pieces itself (one process per GPU over RCCL: libh264_amd/pipeline.py): for
 * i < n  d_dst[d_dst_off[i] ..] = d_src[d_src_off[i] .. + d_sizes[i]); all
 * arrays on the device, d_dst_off 16-byte aligned.  Asynchronous. */
H264_API int h264_amd_pack(struct h264_amd_engine *eng, uint8_t *d_dst,
			   const uint8_t *d_src, const uint64_t *d_src_off,
			   const uint64_t *d_dst_off, const uint32_t *d_sizes,
			   uint32_t n);

/* Like h264_amd_decode_slices() on the first member's device, spread over the
 * group.  h_descs: the host's copy of args->d_descs if the caller has one (else
 * NULL: fetched).  *members_used (optional) reports how many members got work.
 * Synchronous. */
H264_API int h264_amd_group_decode_slices(struct h264_amd_group *g,
					  const struct h264_amd_params *params,
					  const struct h264_amd_decode_args *args,
					  const struct h264_amd_slice_desc *h_descs,
					  uint32_t *members_used);

/* ---- calibration and identification ---------------------------------------- */

/* Time one pure streaming read of d_buf[0 .. len) with the scan kernel's work
 * mapping (HIP events on the engine's stream): the measured read ceiling of
 * the box, next to which the HBM-bound kernels' rates are reported. */
H264_API int h264_amd_probe_read(struct h264_amd_engine *e, const uint8_t *d_buf,
				 size_t len, float *ms);

/* Identity of the sources this library was built from (hex sha256 over
 * libh264_amd/csrc and include, see libh264_amd/build.py), "unidentified" for
 * a build made by hand. */
H264_API const char *h264_amd_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* H264_AMD_H */
