/**
 * libh264_amd -- reader object of the libh264 C API (drop-in for the
 * reference's include/h264/h264_reader.h:31-85).
 *
 * In libh264_amd the reader is a thin host orchestrator over the MI355X
 * kernels: h264_reader_parse() uploads the buffer, runs the Annex-B scan
 * kernel, parses the (tiny) non-VCL NAL units on the host, decodes ALL slice
 * NAL units of the buffer in one one-slice-per-wavefront launch and replays
 * the callbacks in bitstream order.  There is no CPU path for slice NAL units:
 * without a usable GPU the calls fail with -ENODEV.
 */
#ifndef _H264_READER_H_
#define _H264_READER_H_

struct h264_reader;

/* Parse slice_data() (CAVLC only; CABAC slice data is recorded as a raw extent,
 * as in the reference: src/h264_syntax_slice_data.h:715-717) */
#define H264_READER_FLAGS_SLICE_DATA 0x01

/* cbs is copied; cbs == NULL -> -EINVAL (ref src/h264_reader.c:58-89) */
H264_API int h264_reader_new(const struct h264_ctx_cbs *cbs,
			     void *userdata,
			     struct h264_reader **ret_obj);
H264_API int h264_reader_destroy(struct h264_reader *reader);
H264_API struct h264_ctx *h264_reader_get_ctx(struct h264_reader *reader);
/* cooperative cancel, honoured between NAL units */
H264_API int h264_reader_stop(struct h264_reader *reader);

/* Annex-B byte stream.  *off = bytes consumed.  Per-NAL errors are swallowed
 * and 0 is returned (ref src/h264_reader.c:117-143). */
H264_API int h264_reader_parse(struct h264_reader *reader,
			       uint32_t flags,
			       const uint8_t *buf,
			       size_t len,
			       size_t *off);
/* One NAL unit without start code, escape bytes still in place */
H264_API int h264_reader_parse_nalu(struct h264_reader *reader,
				    uint32_t flags,
				    const uint8_t *buf,
				    size_t len);

/* Context-free helpers (host only; ref src/h264_reader.c:165-255) */
H264_API int h264_parse_nalu_header(const uint8_t *buf,
				    size_t len,
				    struct h264_nalu_header *nh);
H264_API int h264_parse_sps(const uint8_t *buf, size_t len, struct h264_sps *sps);
H264_API int h264_parse_pps(const uint8_t *buf,
			    size_t len,
			    const struct h264_sps *sps,
			    struct h264_pps *pps);

#endif /* !_H264_READER_H_ */
