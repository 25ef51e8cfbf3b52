/**
 * libh264_amd -- parsing context and event callbacks of the libh264 C API.
 *
 * Drop-in for the reference's include/h264/h264_ctx.h.  struct h264_ctx_cbs
 * (18 function pointers, 144 bytes, ref :34-152) is copied by value when a
 * reader is created, so member ORDER is ABI.  struct h264_ctx stays opaque.
 *
 * Delivery contract (ref src/h264_syntax.h:1446-1604, kept by the GPU reader):
 * callbacks fire synchronously on the caller's thread, in bitstream order:
 *   nalu_begin -> { slice_data_begin, slice_data_mb x N, slice_data_end, slice
 *                 | sps | pps | aud | sei (+ typed sei_*) }
 *              -> au_end (for the PREVIOUS access unit, when this NAL starts a
 *                 new one; never for the last AU of a stream)
 *              -> nalu_end.
 * A NAL that fails to parse keeps the callbacks already fired and skips the
 * rest (no nalu_end).  Pointers handed to callbacks are borrowed.
 * With libh264_amd the per-macroblock events of a whole batch of slices are
 * computed on the GPU first and then replayed here in order.
 */
#ifndef _H264_CTX_H_
#define _H264_CTX_H_

struct h264_ctx;

struct h264_ctx_cbs {
	void (*au_end)(struct h264_ctx *ctx, void *userdata);

	void (*nalu_begin)(struct h264_ctx *ctx,
			   enum h264_nalu_type type,
			   const uint8_t *buf,
			   size_t len,
			   const struct h264_nalu_header *nh,
			   void *userdata);

	void (*nalu_end)(struct h264_ctx *ctx,
			 enum h264_nalu_type type,
			 const uint8_t *buf,
			 size_t len,
			 const struct h264_nalu_header *nh,
			 void *userdata);

	void (*slice)(struct h264_ctx *ctx,
		      const uint8_t *buf,
		      size_t len,
		      const struct h264_slice_header *sh,
		      void *userdata);

	void (*slice_data_begin)(struct h264_ctx *ctx,
				 const struct h264_slice_header *sh,
				 void *userdata);

	void (*slice_data_end)(struct h264_ctx *ctx,
			       const struct h264_slice_header *sh,
			       uint32_t mb_count,
			       void *userdata);

	void (*slice_data_mb)(struct h264_ctx *ctx,
			      const struct h264_slice_header *sh,
			      uint32_t mb_addr,
			      enum h264_mb_type mb_type,
			      void *userdata);

	void (*sps)(struct h264_ctx *ctx,
		    const uint8_t *buf,
		    size_t len,
		    const struct h264_sps *sps,
		    void *userdata);

	void (*pps)(struct h264_ctx *ctx,
		    const uint8_t *buf,
		    size_t len,
		    const struct h264_pps *pps,
		    void *userdata);

	void (*aud)(struct h264_ctx *ctx,
		    const uint8_t *buf,
		    size_t len,
		    const struct h264_aud *aud,
		    void *userdata);

	void (*sei)(struct h264_ctx *ctx,
		    enum h264_sei_type type,
		    const uint8_t *buf,
		    size_t len,
		    void *userdata);

	void (*sei_buffering_period)(struct h264_ctx *ctx,
				     const uint8_t *buf,
				     size_t len,
				     const struct h264_sei_buffering_period *sei,
				     void *userdata);

	void (*sei_pic_timing)(struct h264_ctx *ctx,
			       const uint8_t *buf,
			       size_t len,
			       const struct h264_sei_pic_timing *sei,
			       void *userdata);

	void (*sei_pan_scan_rect)(struct h264_ctx *ctx,
				  const uint8_t *buf,
				  size_t len,
				  const struct h264_sei_pan_scan_rect *sei,
				  void *userdata);

	void (*sei_filler_payload)(struct h264_ctx *ctx,
				   const uint8_t *buf,
				   size_t len,
				   const struct h264_sei_filler_payload *sei,
				   void *userdata);

	void (*sei_user_data_registered)(
		struct h264_ctx *ctx,
		const uint8_t *buf,
		size_t len,
		const struct h264_sei_user_data_registered *sei,
		void *userdata);

	void (*sei_user_data_unregistered)(
		struct h264_ctx *ctx,
		const uint8_t *buf,
		size_t len,
		const struct h264_sei_user_data_unregistered *sei,
		void *userdata);

	void (*sei_recovery_point)(struct h264_ctx *ctx,
				   const uint8_t *buf,
				   size_t len,
				   const struct h264_sei_recovery_point *sei,
				   void *userdata);
};

/* Lifecycle (ref :155-168; src/h264_ctx.c:226-281) */
H264_API int h264_ctx_new(struct h264_ctx **ret_obj);
H264_API int h264_ctx_destroy(struct h264_ctx *ctx); /* NULL -> 0 */
H264_API int h264_ctx_clear(struct h264_ctx *ctx); /* drops SPS/PPS too */
H264_API int h264_ctx_clear_nalu(struct h264_ctx *ctx); /* keeps SPS/PPS */

/* Describe the NAL a writer call will serialise (ref :171-196) */
H264_API int h264_ctx_set_nalu_header(struct h264_ctx *ctx,
				      const struct h264_nalu_header *nh);
H264_API int h264_ctx_is_nalu_unknown(struct h264_ctx *ctx);
H264_API int h264_ctx_set_aud(struct h264_ctx *ctx, const struct h264_aud *aud);
/* copy into the per-id table (32 SPS / 256 PPS) and make it the active one */
H264_API int h264_ctx_set_sps(struct h264_ctx *ctx, const struct h264_sps *sps);
H264_API int h264_ctx_set_pps(struct h264_ctx *ctx, const struct h264_pps *pps);
H264_API int h264_ctx_set_filler(struct h264_ctx *ctx, size_t len);
H264_API const struct h264_sps *h264_ctx_get_sps(struct h264_ctx *ctx);
H264_API const struct h264_pps *h264_ctx_get_pps(struct h264_ctx *ctx);

/* SEI table of the current NAL (ref :199-215) */
H264_API int h264_ctx_add_sei(struct h264_ctx *ctx, const struct h264_sei *sei);
H264_API int h264_ctx_get_sei_count(struct h264_ctx *ctx);
H264_API uint64_t
h264_ctx_sei_pic_timing_to_ts(struct h264_ctx *ctx,
			      const struct h264_sei_pic_timing *sei);
H264_API uint64_t
h264_ctx_sei_pic_timing_to_us(struct h264_ctx *ctx,
			      const struct h264_sei_pic_timing *sei);

/* (ref :218-224) */
H264_API int h264_ctx_set_slice_header(struct h264_ctx *ctx,
				       const struct h264_slice_header *sh);
/* -EAGAIN until both an SPS and a PPS are active */
H264_API int h264_ctx_get_info(struct h264_ctx *ctx, struct h264_info *info);

#endif /* !_H264_CTX_H_ */
