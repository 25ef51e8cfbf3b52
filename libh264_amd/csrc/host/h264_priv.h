/**
 * h264_priv.h -- private declarations of the drop-in host library (libh264.so).
 *
 * The host library is the control plane of the engine: it owns the libh264
 * object model (struct h264_ctx, struct h264_reader -- counterparts of the
 * reference's src/h264_priv.h:67-140 and src/h264_reader.c:30-36), parses and
 * serialises the small non-VCL NAL units (parameter sets, SEI, AUD, filler),
 * performs access-unit detection and replays callbacks in bitstream order.
 * Slice NAL units -- header and data, reading and writing -- are handed to
 * the MI355X kernels through <h264_amd.h>; the one exception is the header of a
 * single slice whose data nobody asked for (slicehdr.c), and that path too
 * refuses to run without the engine: there is no CPU fallback here.
 */
#ifndef H264_AMD_HOST_PRIV_H
#define H264_AMD_HOST_PRIV_H

#include <h264/h264.h>
#include <h264_amd.h>

#define H264_ARRAY_SIZE(a) (sizeof(a) / sizeof((a)[0]))

struct h264_ctx {
	struct {
		enum h264_nalu_type type;
		struct h264_nalu_header hdr;
		int unknown;
		int is_first_vcl;
		int is_prev_vcl;
		int is_prev_filler;
	} nalu;
	struct h264_aud aud;
	struct h264_sps *sps; /* active */
	struct h264_pps *pps; /* active */
	struct h264_sps *sps_table[32];
	struct h264_pps *pps_table[256];
	struct h264_sei *sei_table;
	uint32_t sei_count;
	size_t filler_len;
	struct {
		enum h264_slice_type type;
		struct h264_slice_header hdr;
		size_t hdr_len; /* raw bits */
		struct h264_slice_header saved_hdr;
		struct {
			uint8_t partial;
			uint8_t partialbits;
			const uint8_t *buf;
			size_t len;
		} rawdata;
		struct h264_nalu_header prev_slice_nalu_hdr;
		struct h264_slice_header prev_slice_hdr;
	} slice;
	struct h264_sps_derived sps_derived;
	struct {
		uint32_t SliceGroupChangeRate;
		int MbaffFrameFlag;
		uint32_t PicHeightInMbs;
		uint32_t PicSizeInMbs;
		uint32_t MaxPicNum, CurrPicNum;
		int32_t SliceQPLuma, QSLuma;
		uint32_t FilterOffsetA, FilterOffsetB;
		uint32_t MapUnitsInSliceGroup0;
	} derived;
	/* bumped whenever an SPS/PPS table entry changes: lets the writer
	 * cache its device-side parameter snapshot */
	uint64_t ps_generation;
	struct h264_amd_params *wr_params;
	uint64_t wr_params_generation;
};

struct h264_reader {
	struct h264_ctx_cbs cbs;
	void *userdata;
	int stop;
	struct h264_ctx *ctx;
	uint32_t flags;
	/* engine side */
	struct h264_amd_params *params; /* every SPS/PPS version seen */
	uint64_t ps_gen_seen; /* ctx->ps_generation `params` is in step with */
	void *d_buf; /* device copy of the caller's buffer */
	size_t d_buf_cap;
	void *d_nals, *d_descs, *d_results, *d_headers, *d_mb;
	size_t cap_nals, cap_descs, cap_results, cap_headers, cap_mb;
	/* one slice decoded again with more room (retry_big_slice): buffers of
	 * its own, the launch's results stay on the device for later pages */
	void *d_retry, *d_retry_mb;
	size_t cap_retry, cap_retry_mb;
	/* host images of one decoded chunk of slices: the page the replay reads
	 * (h_results / h_headers / h_mb point into page[cur_page]) and the page
	 * that is on its way from the device meanwhile; page-locked memory
	 * where the runtime grants it */
	struct rd_host_page {
		struct h264_amd_slice_result *results;
		struct h264_slice_header *headers;
		uint32_t *mb;
		size_t cap_results, cap_headers, cap_mb;
		int pinned_results, pinned_headers, pinned_mb;
	} page[2];
	int cur_page;
	double trace_wait_ms; /* H264_AMD_READER_TRACE */
	struct h264_amd_slice_result *h_results;
	struct h264_slice_header *h_headers;
	uint32_t *h_mb;
	uint32_t *h_big; /* one slice that outgrew the batch's mb_capacity */
	size_t hcap_big;
	/* slice group maps of the batch and of that one slice, fmo_pitch bytes
	 * each (0: the launch built none) */
	uint8_t *h_fmo, *h_fmo_big;
	size_t hcap_fmo, hcap_fmo_big, fmo_pitch;
};

/* direction-agnostic syntax cursor with an error latch (paramsets.c) */
struct h264_syn {
	struct h264_bitstream *bs;
	int wr;
	int err; /* first negative errno, 0 while fine */
};

/* ---- ctx.c ---- */
int h264_ctx_set_active_sps(struct h264_ctx *ctx, uint32_t sps_id);
int h264_ctx_set_active_pps(struct h264_ctx *ctx, uint32_t pps_id);
int h264_ctx_clear_sei_table(struct h264_ctx *ctx);
int h264_ctx_add_sei_internal(struct h264_ctx *ctx, struct h264_sei **ret);
int h264_ctx_clear_slice(struct h264_ctx *ctx);
int h264_get_info_from_ps(const struct h264_sps *sps,
			  const struct h264_sps_derived *d,
			  struct h264_info *info);
int h264_sei_update_internal_buf(struct h264_sei *sei);
uint32_t h264_intlog2(uint32_t x);

/* ---- paramsets.c ---- */
int h264_syn_nalu_header(struct h264_syn *s, struct h264_nalu_header *nh);
int h264_syn_sps(struct h264_syn *s, struct h264_sps *sps);
int h264_syn_pps_ids(struct h264_syn *s, struct h264_pps *pps);
int h264_syn_pps_body(struct h264_syn *s, const struct h264_sps *sps,
		      struct h264_pps *pps);
int h264_syn_aud(struct h264_syn *s, struct h264_aud *aud);
int h264_syn_filler(struct h264_syn *s, size_t *len);
int h264_syn_sei_read(struct h264_bitstream *bs, struct h264_ctx *ctx,
		      const struct h264_ctx_cbs *cbs, void *userdata);
int h264_syn_sei_write(struct h264_bitstream *bs, struct h264_ctx *ctx);
int h264_write_one_sei(struct h264_bitstream *bs, struct h264_ctx *ctx,
		       const struct h264_sei *sei);

/* ---- slicehdr.c: slice header on the host for header-only calls ---- */
int h264_host_slice_is_header_only(const struct h264_ctx *ctx, uint32_t flags,
				   const uint8_t *nal, size_t len);
int h264_host_slice_header(const struct h264_ctx *ctx, const uint8_t *nal,
			   size_t len, struct h264_amd_slice_result *r,
			   struct h264_slice_header *sh);

/* ---- engine glue (gpu.c) ---- */
/* process-wide engine on device $H264_AMD_DEVICE (default 0); -ENODEV when
 * there is no usable GPU */
int h264_host_engine(struct h264_amd_engine **ret);
struct h264_amd_group *h264_host_group(void);
int h264_host_grow(void **d_ptr, size_t *cap, size_t want);
void h264_host_lock(void);
void h264_host_unlock(void);
int h264_host_params_from_ctx(struct h264_amd_params *p,
			      const struct h264_ctx *ctx);
/* snapshot of ctx's CURRENT parameter-set tables for a writer call */
int h264_host_ctx_params(struct h264_ctx *ctx, struct h264_amd_params **ret,
			 uint32_t *epoch);

#endif /* H264_AMD_HOST_PRIV_H */
