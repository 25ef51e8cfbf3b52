/**
 * libh264_amd -- serialisation entry points of the libh264 C API (drop-in for
 * the reference's include/h264/h264_writer.h:31-50).  There is no writer
 * object: every call serialises the NAL unit currently described by `ctx` into
 * the caller's bitstream, without a start code.  The batched, device-side
 * counterparts are h264_amd_write_slices() / h264_amd_encode_synth_slices()
 * in <h264_amd.h>.
 */
#ifndef _H264_WRITER_H_
#define _H264_WRITER_H_

/* NAL header + payload for ctx->nalu.hdr.nal_unit_type.  For slices: header
 * fields re-encoded one by one, then the saved partial byte and the raw
 * slice-data bytes verbatim (ref src/h264_syntax_slice_data.h:813-830). */
H264_API int h264_write_nalu(struct h264_bitstream *bs, struct h264_ctx *ctx);

/* Synthetic slices for concealment (ref src/h264_writer.c:254-309): mb_count
 * I_16x16 DC-predicted macroblocks without residual, resp. mb_count skipped
 * macroblocks; CAVLC or CABAC according to the active PPS. */
H264_API int h264_write_grey_i_slice(struct h264_bitstream *bs,
				     struct h264_ctx *ctx,
				     uint32_t mb_count);
H264_API int h264_write_skipped_p_slice(struct h264_bitstream *bs,
					struct h264_ctx *ctx,
					uint32_t mb_count);

/* In-place header patch; the new header must have the same bit length or
 * -EPROTO is returned and ctx is restored (ref src/h264_writer.c:312-370). */
H264_API int h264_rewrite_slice_header(struct h264_bitstream *bs,
				       struct h264_ctx *ctx,
				       const struct h264_slice_header *sh);

#endif /* !_H264_WRITER_H_ */
