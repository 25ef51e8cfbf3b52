/**
 * engine.hip -- C-ABI shim of the batched MI355X engine (<h264_amd.h>).
 *
 * Thin host layer: owns the HIP stream, the lane-table image, the compact
 * parameter-set snapshots and the scratch the kernels need; every entry point
 * validates its arguments, makes sure the device-side tables are current and
 * launches one hand-written gfx950 kernel.  No parsing happens here and there
 * is no CPU fallback: if HIP cannot give us a gfx950 device the calls return
 * -ENODEV.
 *
 * The kernels are compiled into this translation unit (no -fgpu-rdc needed).
 */
#include <hip/hip_runtime.h>

#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <type_traits>
#include <vector>

#define H264_API_EXPORTS 1
#include <h264_amd.h>

#include "kernels/dev_params.h"
#include "kernels/scan.hip"
#include "kernels/slice_decode.hip"
#include "kernels/slice_write.hip"
#include "kernels/cabac_slice.hip"

/* ---- helpers --------------------------------------------------------------- */

static int hip_err(hipError_t e)
{
	switch (e) {
	case hipSuccess:
		return 0;
	case hipErrorOutOfMemory:
		return -ENOMEM;
	case hipErrorInvalidValue:
	case hipErrorInvalidDevicePointer:
		return -EINVAL;
	case hipErrorNoDevice:
	case hipErrorInvalidDevice:
	case hipErrorNoBinaryForGpu:
	case hipErrorInvalidDeviceFunction:
		return -ENODEV;
	default:
		return -EIO;
	}
}

#define HIP_TRY(x)                                                             \
	do {                                                                   \
		hipError_t _e = (x);                                           \
		if (_e != hipSuccess) {                                        \
			fprintf(stderr, "libh264_amd: %s failed: %s\n", #x,    \
				hipGetErrorString(_e));                        \
			return hip_err(_e);                                    \
		}                                                              \
	} while (0)

enum { FAM_SCAN, FAM_DECODE, FAM_WRITE, FAM_SYNTH, FAM_SCAN_ALL, FAM_CABAC, FAM_AVCC, FAM_COUNT };

/* Serial numbers are process-wide so that a params object allocated at the
 * address of a freed one can never look "unchanged" to an engine's cache. */
static uint64_t next_serial(void)
{
	static uint64_t g_serial;
	return __atomic_add_fetch(&g_serial, 1, __ATOMIC_RELAXED);
}

struct h264_amd_params {
	std::vector<h264_sps> sps_full;
	std::vector<h264_pps> pps_full;
	std::vector<dev_sps> sps;
	std::vector<dev_pps> pps;
	std::vector<dev_fmo> fmo; /* parallel to pps */
	uint32_t n_fmo; /* PPS versions with slice groups */
	std::vector<dev_epoch> epochs;
	uint64_t serial; /* bumped on every change */
	uint32_t max_width_mbs, max_pic_mbs;
	bool any_mbaff; /* an SPS with mb_adaptive_frame_field_flag was seen */
};

struct h264_amd_engine {
	int device;
	uint32_t cu_count;
	hipStream_t stream;
	bool own_stream;
	/* h264_amd_download_async(): copies that run behind the engine's work
	 * without holding its stream */
	hipStream_t copy_stream;
	hipEvent_t copy_after;
	dev_vlc_tables *d_vlc;
	cabac_tables *d_cabac;
	cabac_dec_tables *d_cabac_dec;
	/* device copy of the last params object used */
	const h264_amd_params *params_src;
	uint64_t params_serial;
	dev_sps *d_sps;
	dev_pps *d_pps;
	dev_fmo *d_fmo;
	dev_epoch *d_epochs;
	size_t cap_sps, cap_pps, cap_fmo, cap_epochs;
	uint32_t *d_order; /* launch position -> slice, longest first */
	size_t cap_order;
	int use_order; /* H264_AMD_ORDER=0: positions in descriptor order (A/B runs) */
	uint8_t *d_fmo_map; /* per-slice macroblock-to-slice-group maps */
	size_t cap_fmo_map;
	uint32_t fmo_stride, fmo_n; /* of the last decode launch */
	hipEvent_t ev[FAM_COUNT][2];
	bool timed[FAM_COUNT];
	/* scan scratch */
	scan_scratch scan;
	uint32_t scan_seg_override; /* H264_AMD_SCAN_SEG (experiments, tests) */
	struct h264_amd_nal *d_avcc_nals; /* h264_amd_byte_stream_to_avcc */
	size_t cap_avcc_nals;
};

/* ---- lane tables from the standard's bit strings -------------------------- */

struct code_word {
	uint32_t mask, code, len;
};

static code_word parse_bits(const char *s)
{
	code_word c = {0, 0, 0};
	uint32_t v = 0;
	for (; *s; s++) {
		if (*s == '0' || *s == '1') {
			v = (v << 1) | (uint32_t)(*s - '0');
			c.len++;
		}
	}
	c.mask = c.len ? 0xffffffffu << (32 - c.len) : 0;
	c.code = c.len ? v << (32 - c.len) : 0;
	return c;
}

static void put_entry(dev_vlc_entry *e, const char *bits, uint32_t payload)
{
	code_word c = parse_bits(bits);
	e->mask = c.mask;
	e->cw = c.code | payload | c.len; /* len in bits 0-4 of every payload */
}

/* Layout documented in kernels/vlc_lanes.hip.h */
static void vlc_build_tables(dev_vlc_tables *t)
{
	uint32_t next[DEV_VLC_CT_CLASSES] = {0, 0, 0, 0, 0, 0};
	uint32_t tz_n[DEV_VLC_TZ_TABLES];
	memset(t, 0, sizeof(*t));
	memset(tz_n, 0, sizeof(tz_n));
	/* the .inc numbers the columns 0,1,2 (nC ranges), 3 (6-bit FLC, nC >= 8),
	 * 4 and 5 (chroma DC): the same order as the classes here */
#define H264_CT(col, bits, t1, tc)                                             \
	put_entry(&t->ct[col][next[col]++], bits,                              \
		  ((uint32_t)(t1) << 5) | ((uint32_t)(tc) << 7));
#define H264_TZ(idx, bits, v)                                                  \
	do {                                                                   \
		put_entry(&t->tz[(idx)-1][v], bits, 0);                        \
		tz_n[(idx)-1]++;                                               \
	} while (0);
#define H264_TZ_C420(idx, bits, v)                                             \
	do {                                                                   \
		put_entry(&t->tz[14 + (idx)][v], bits, 0);                     \
		tz_n[14 + (idx)]++;                                            \
	} while (0);
#define H264_TZ_C422(idx, bits, v)                                             \
	do {                                                                   \
		put_entry(&t->tz[17 + (idx)][v], bits, 0);                     \
		tz_n[17 + (idx)]++;                                            \
	} while (0);
#define H264_CBP(cls, code, intra, inter)                                      \
	t->cbp[code] |= ((uint32_t)(intra) << (16 * (cls))) |                  \
			((uint32_t)(inter) << (16 * (cls) + 8));
#include "common/h264_cavlc_tables.inc"
#undef H264_CT
#undef H264_TZ
#undef H264_TZ_C420
#undef H264_TZ_C422
#undef H264_CBP
	/* coeff_token rows end in catch-all entries (mask 0: every bit string
	 * matches; the decoder takes the LOWEST matching entry, so they only win
	 * when no code word does): length 0, no coefficients, DEV_VLC_BAD set --
	 * a malformed token needs no "nothing matched" case in the kernel.  Every
	 * class leaves room: 62 code words with nC, 14 / 30 for chroma DC */
	for (int c = 0; c < DEV_VLC_CT_CLASSES; c++) {
		if (next[c] > 63)
			abort();
		for (uint32_t l = next[c]; l < 64; l++) {
			t->ct[c][l].mask = 0;
			t->ct[c][l].cw = DEV_VLC_BAD;
		}
	}
	/* total_zeros tables can be full (16 values): entries without a code
	 * word repeat entry 0, a duplicate never wins */
	for (int k = 0; k < DEV_VLC_TZ_TABLES; k++)
		for (uint32_t l = tz_n[k]; l < 16; l++)
			t->tz[k][l] = t->tz[k][0];
}

/* ---- engine ----------------------------------------------------------------- */

extern "C" int h264_amd_available(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n > 0;
}

extern "C" int h264_amd_engine_new(int device, void *hip_stream,
				   struct h264_amd_engine **ret)
{
	if (ret == NULL)
		return -EINVAL;
	*ret = NULL;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n)
		return -ENODEV;
	HIP_TRY(hipSetDevice(device));
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
		fprintf(stderr,
			"libh264_amd: device %d is %s, kernels are built for gfx950\n",
			device, prop.gcnArchName);
		return -ENODEV;
	}
	h264_amd_engine *e = new (std::nothrow) h264_amd_engine();
	if (e == NULL)
		return -ENOMEM;
	memset(e, 0, sizeof(*e));
	/* a failure from here on releases what exists so far */
#define NEW_TRY(x)                                                             \
	do {                                                                   \
		hipError_t err_ = (x);                                         \
		if (err_ != hipSuccess) {                                      \
			int rc_ = hip_err(err_);                               \
			h264_amd_engine_destroy(e);                            \
			return rc_;                                            \
		}                                                              \
	} while (0)
	e->device = device;
	{
		/* workgroups are dispatched in layers of one per compute unit:
		 * the decode kernel derives a wave's age rank from it */
		int cus = 0;
		NEW_TRY(hipDeviceGetAttribute(&cus,
					      hipDeviceAttributeMultiprocessorCount,
					      device));
		e->cu_count = cus > 0 ? (uint32_t)cus : 1u;
	}
	if (hip_stream != NULL) {
		e->stream = (hipStream_t)hip_stream;
	} else {
		NEW_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
		e->own_stream = true;
	}
	for (int f = 0; f < FAM_COUNT; f++)
		for (int k = 0; k < 2; k++)
			NEW_TRY(hipEventCreate(&e->ev[f][k]));
	{
		const char *v = getenv("H264_AMD_ORDER");
		e->use_order = !(v != NULL && v[0] == '0');
	}
	{
		const char *v = getenv("H264_AMD_SCAN_SEG");
		unsigned long seg = v ? strtoul(v, NULL, 0) : 0;
		if (seg >= SCAN_BLOCK_BYTES && seg <= (1u << 30) &&
		    seg % SCAN_BLOCK_BYTES == 0)
			e->scan_seg_override = (uint32_t)seg;
	}
	{
		dev_vlc_tables *h = new dev_vlc_tables;
		vlc_build_tables(h);
		NEW_TRY(hipMalloc((void **)&e->d_vlc, sizeof(*h)));
		NEW_TRY(hipMemcpy(e->d_vlc, h, sizeof(*h), hipMemcpyHostToDevice));
		delete h;
	}
	{
		cabac_tables *h = new cabac_tables;
		cabac_build_tables(h);
		NEW_TRY(hipMalloc((void **)&e->d_cabac, sizeof(*h)));
		NEW_TRY(hipMemcpy(e->d_cabac, h, sizeof(*h), hipMemcpyHostToDevice));
		delete h;
	}
	{
		cabac_dec_tables *h = new cabac_dec_tables;
		cabac_dec_build_tables(h);
		NEW_TRY(hipMalloc((void **)&e->d_cabac_dec, sizeof(*h)));
		NEW_TRY(hipMemcpy(e->d_cabac_dec, h, sizeof(*h), hipMemcpyHostToDevice));
		delete h;
	}
#undef NEW_TRY
	*ret = e;
	return 0;
}

extern "C" int h264_amd_engine_destroy(struct h264_amd_engine *e)
{
	if (e == NULL)
		return 0;
	hipSetDevice(e->device);
	if (e->stream != NULL)
		hipStreamSynchronize(e->stream);
	for (int f = 0; f < FAM_COUNT; f++)
		for (int k = 0; k < 2; k++)
			if (e->ev[f][k] != NULL)
				hipEventDestroy(e->ev[f][k]);
	hipFree(e->d_vlc);
	hipFree(e->d_cabac);
	hipFree(e->d_cabac_dec);
	hipFree(e->d_sps);
	hipFree(e->d_pps);
	hipFree(e->d_fmo);
	hipFree(e->d_fmo_map);
	hipFree(e->d_order);
	hipFree(e->d_epochs);
	scan_scratch_free(&e->scan);
	hipFree(e->d_avcc_nals);
	if (e->copy_stream != NULL) {
		hipStreamDestroy(e->copy_stream);
		hipEventDestroy(e->copy_after);
	}
	if (e->own_stream)
		hipStreamDestroy(e->stream);
	delete e;
	return 0;
}

extern "C" int h264_amd_engine_sync(struct h264_amd_engine *e)
{
	if (e == NULL)
		return -EINVAL;
	HIP_TRY(hipStreamSynchronize(e->stream));
	return 0;
}

extern "C" void *h264_amd_engine_stream(struct h264_amd_engine *e)
{
	return e ? (void *)e->stream : NULL;
}

extern "C" int h264_amd_malloc(struct h264_amd_engine *e, size_t size,
			       void **d_ptr)
{
	if (e == NULL || d_ptr == NULL)
		return -EINVAL;
	HIP_TRY(hipSetDevice(e->device));
	/* slack so that 16-byte vector accesses at the tail stay in bounds */
	HIP_TRY(hipMalloc(d_ptr, size + 64));
	return 0;
}

extern "C" int h264_amd_free(struct h264_amd_engine *e, void *d_ptr)
{
	if (e == NULL)
		return -EINVAL;
	HIP_TRY(hipFree(d_ptr));
	return 0;
}

extern "C" int h264_amd_upload(struct h264_amd_engine *e, void *d_dst,
			       const void *src, size_t size)
{
	if (e == NULL)
		return -EINVAL;
	HIP_TRY(hipMemcpyAsync(d_dst, src, size, hipMemcpyHostToDevice, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	return 0;
}

extern "C" int h264_amd_download(struct h264_amd_engine *e, void *dst,
				 const void *d_src, size_t size)
{
	if (e == NULL)
		return -EINVAL;
	HIP_TRY(hipMemcpyAsync(dst, d_src, size, hipMemcpyDeviceToHost, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	return 0;
}

extern "C" int h264_amd_host_alloc(struct h264_amd_engine *e, size_t size, void **ptr)
{
	if (e == NULL || ptr == NULL)
		return -EINVAL;
	HIP_TRY(hipSetDevice(e->device));
	*ptr = NULL;
	if (hipHostMalloc(ptr, size ? size : 1, hipHostMallocDefault) != hipSuccess) {
		(void)hipGetLastError();
		*ptr = NULL;
		return -ENOMEM;
	}
	return 0;
}

extern "C" int h264_amd_host_free(struct h264_amd_engine *e, void *ptr)
{
	if (e == NULL)
		return -EINVAL;
	if (ptr != NULL)
		HIP_TRY(hipHostFree(ptr));
	return 0;
}

extern "C" int h264_amd_download_async(struct h264_amd_engine *e, void *dst,
				       const void *d_src, size_t size)
{
	if (e == NULL)
		return -EINVAL;
	HIP_TRY(hipSetDevice(e->device));
	if (e->copy_stream == NULL) {
		HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
		HIP_TRY(hipEventCreateWithFlags(&e->copy_after, hipEventDisableTiming));
	}
	/* behind everything queued on the engine's stream so far */
	HIP_TRY(hipEventRecord(e->copy_after, e->stream));
	HIP_TRY(hipStreamWaitEvent(e->copy_stream, e->copy_after, 0));
	if (size)
		HIP_TRY(hipMemcpyAsync(dst, d_src, size, hipMemcpyDeviceToHost,
				       e->copy_stream));
	return 0;
}

extern "C" int h264_amd_download_wait(struct h264_amd_engine *e)
{
	if (e == NULL)
		return -EINVAL;
	if (e->copy_stream != NULL)
		HIP_TRY(hipStreamSynchronize(e->copy_stream));
	return 0;
}

extern "C" int h264_amd_copy(struct h264_amd_engine *e, void *d_dst,
			     const void *d_src, size_t size)
{
	if (e == NULL || (size && (d_dst == NULL || d_src == NULL)))
		return -EINVAL;
	HIP_TRY(hipSetDevice(e->device));
	if (size)
		HIP_TRY(hipMemcpyAsync(d_dst, d_src, size, hipMemcpyDeviceToDevice,
				       e->stream));
	return 0;
}

extern "C" int h264_amd_memset(struct h264_amd_engine *e, void *d_dst, int value,
			       size_t size)
{
	if (e == NULL)
		return -EINVAL;
	HIP_TRY(hipMemsetAsync(d_dst, value, size, e->stream));
	return 0;
}

extern "C" int h264_amd_last_kernel_ms(struct h264_amd_engine *e,
				       const char *family, float *ms)
{
	static const char *const names[FAM_COUNT] = {"scan", "decode", "write",
						     "synth", "scan_all", "cabac",
						     "avcc"};
	if (e == NULL || family == NULL || ms == NULL)
		return -EINVAL;
	for (int f = 0; f < FAM_COUNT; f++) {
		if (strcmp(family, names[f]) != 0)
			continue;
		if (!e->timed[f])
			return -ENOENT;
		HIP_TRY(hipEventSynchronize(e->ev[f][1]));
		HIP_TRY(hipEventElapsedTime(ms, e->ev[f][0], e->ev[f][1]));
		return 0;
	}
	return -EINVAL;
}

/* ---- parameter-set snapshots ---------------------------------------------- */

extern "C" int h264_amd_params_new(struct h264_amd_params **ret)
{
	if (ret == NULL)
		return -EINVAL;
	h264_amd_params *p = new (std::nothrow) h264_amd_params();
	if (p == NULL)
		return -ENOMEM;
	*ret = p;
	return h264_amd_params_reset(p);
}

extern "C" int h264_amd_params_destroy(struct h264_amd_params *p)
{
	delete p;
	return 0;
}

extern "C" int h264_amd_params_reset(struct h264_amd_params *p)
{
	if (p == NULL)
		return -EINVAL;
	p->sps_full.clear();
	p->pps_full.clear();
	p->sps.clear();
	p->pps.clear();
	p->fmo.clear();
	p->n_fmo = 0;
	p->epochs.clear();
	dev_epoch e0;
	memset(&e0, 0xff, sizeof(e0)); /* every id: DEV_NO_VERSION */
	p->epochs.push_back(e0);
	p->serial = next_serial();
	p->max_width_mbs = 1;
	p->max_pic_mbs = 1;
	p->any_mbaff = false;
	return 0;
}

extern "C" int h264_amd_params_has_slice_groups(const struct h264_amd_params *p)
{
	return p != NULL && p->n_fmo > 0;
}

extern "C" int h264_amd_params_epoch(const struct h264_amd_params *p)
{
	return p ? (int)p->epochs.size() - 1 : -EINVAL;
}

extern "C" int h264_amd_params_put_sps(struct h264_amd_params *p,
				       const struct h264_sps *sps)
{
	if (p == NULL || sps == NULL || sps->seq_parameter_set_id >= 32)
		return -EINVAL;
	const dev_epoch &cur = p->epochs.back();
	uint16_t v = cur.sps_ver[sps->seq_parameter_set_id];
	if (v != DEV_NO_VERSION && memcmp(&p->sps_full[v], sps, sizeof(*sps)) == 0)
		return h264_amd_params_epoch(p);
	if (p->sps.size() >= DEV_NO_VERSION)
		return -ENOSPC;
	dev_sps d;
	memset(&d, 0, sizeof(d));
	d.chroma_format_idc = DEV_SAT8(sps->chroma_format_idc);
	d.separate_colour_plane_flag = (uint8_t)(sps->separate_colour_plane_flag != 0);
	/* the SPS parser refuses minus8 > 6; a hand-built SPS beyond that gets
	 * memory-safe garbage samples */
	d.bit_depth_luma = DEV_SAT8(8u + sps->bit_depth_luma_minus8);
	d.bit_depth_chroma = DEV_SAT8(8u + sps->bit_depth_chroma_minus8);
	d.log2_max_frame_num = 4u + sps->log2_max_frame_num_minus4;
	d.pic_order_cnt_type = DEV_SAT8(sps->pic_order_cnt_type);
	d.log2_max_pic_order_cnt_lsb = 4u + sps->log2_max_pic_order_cnt_lsb_minus4;
	d.delta_pic_order_always_zero_flag =
		(uint8_t)(sps->delta_pic_order_always_zero_flag != 0);
	d.frame_mbs_only_flag = (uint8_t)(sps->frame_mbs_only_flag != 0);
	d.mb_adaptive_frame_field_flag = (uint8_t)(sps->mb_adaptive_frame_field_flag != 0);
	d.direct_8x8_inference_flag = (uint8_t)(sps->direct_8x8_inference_flag != 0);
	d.pic_width_in_mbs = sps->pic_width_in_mbs_minus1 + 1;
	d.pic_height_in_map_units = sps->pic_height_in_map_units_minus1 + 1;
	p->sps_full.push_back(*sps);
	p->sps.push_back(d);
	dev_epoch ne = cur;
	ne.sps_ver[sps->seq_parameter_set_id] = (uint16_t)(p->sps.size() - 1);
	p->epochs.push_back(ne);
	p->serial = next_serial();
	if (d.pic_width_in_mbs > p->max_width_mbs)
		p->max_width_mbs = d.pic_width_in_mbs;
	if (d.mb_adaptive_frame_field_flag && !d.frame_mbs_only_flag)
		p->any_mbaff = true;
	uint64_t mbs = (uint64_t)d.pic_width_in_mbs * d.pic_height_in_map_units *
		       (2 - (sps->frame_mbs_only_flag ? 1 : 0));
	if (mbs > p->max_pic_mbs && mbs < 0x7fffffffu)
		p->max_pic_mbs = (uint32_t)mbs;
	return h264_amd_params_epoch(p);
}

extern "C" int h264_amd_params_put_pps(struct h264_amd_params *p,
				       const struct h264_pps *pps)
{
	if (p == NULL || pps == NULL || pps->pic_parameter_set_id >= 256)
		return -EINVAL;
	const dev_epoch &cur = p->epochs.back();
	uint16_t v = cur.pps_ver[pps->pic_parameter_set_id];
	if (v != DEV_NO_VERSION && memcmp(&p->pps_full[v], pps, sizeof(*pps)) == 0)
		return h264_amd_params_epoch(p);
	if (p->pps.size() >= DEV_NO_VERSION)
		return -ENOSPC;
	dev_pps d;
	memset(&d, 0, sizeof(d));
	d.seq_parameter_set_id = DEV_SAT8(pps->seq_parameter_set_id);
	d.entropy_coding_mode_flag = (uint8_t)(pps->entropy_coding_mode_flag != 0);
	d.bottom_field_pic_order_in_frame_present_flag =
		(uint8_t)(pps->bottom_field_pic_order_in_frame_present_flag != 0);
	d.num_slice_groups_minus1 = DEV_SAT8(pps->num_slice_groups_minus1);
	d.slice_group_map_type = DEV_SAT8(pps->slice_group_map_type);
	d.weighted_pred_flag = (uint8_t)(pps->weighted_pred_flag != 0);
	d.weighted_bipred_idc = DEV_SAT8(pps->weighted_bipred_idc);
	d.deblocking_filter_control_present_flag =
		(uint8_t)(pps->deblocking_filter_control_present_flag != 0);
	d.redundant_pic_cnt_present_flag =
		(uint8_t)(pps->redundant_pic_cnt_present_flag != 0);
	d.transform_8x8_mode_flag = (uint8_t)(pps->transform_8x8_mode_flag != 0);
	d.constrained_intra_pred_flag = (uint8_t)(pps->constrained_intra_pred_flag != 0);
	d.pic_init_qp_minus26 = pps->pic_init_qp_minus26;
	d.pic_init_qs_minus26 = pps->pic_init_qs_minus26;
	d.num_ref_idx_l0_default_active_minus1 =
		pps->num_ref_idx_l0_default_active_minus1;
	d.num_ref_idx_l1_default_active_minus1 =
		pps->num_ref_idx_l1_default_active_minus1;
	d.slice_group_change_rate_minus1 = pps->slice_group_change_rate_minus1;
	dev_fmo f;
	memset(&f, 0, sizeof(f));
	if (pps->num_slice_groups_minus1 > 0) {
		f.slice_group_change_direction_flag =
			pps->slice_group_change_direction_flag != 0;
		f.pic_size_in_map_units_minus1 = pps->pic_size_in_map_units_minus1;
		for (int i = 0; i < 8; i++) {
			f.run_length_minus1[i] = pps->run_length_minus1[i];
			f.top_left[i] = pps->top_left[i];
			f.bottom_right[i] = pps->bottom_right[i];
		}
		for (int i = 0; i < 256; i++)
			f.slice_group_id[i] = DEV_SAT8(pps->slice_group_id[i]);
		p->n_fmo++;
	}
	p->pps_full.push_back(*pps);
	p->pps.push_back(d);
	p->fmo.push_back(f);
	dev_epoch ne = cur;
	ne.pps_ver[pps->pic_parameter_set_id] = (uint16_t)(p->pps.size() - 1);
	p->epochs.push_back(ne);
	p->serial = next_serial();
	return h264_amd_params_epoch(p);
}

template <typename T>
static int upload_vec(T **d, size_t *cap, const std::vector<T> &v,
		      hipStream_t s)
{
	size_t n = v.size() ? v.size() : 1;
	if (n > *cap) {
		hipFree(*d);
		*d = NULL;
		size_t nc = n * 2;
		HIP_TRY(hipMalloc((void **)d, nc * sizeof(T)));
		*cap = nc;
	}
	if (v.size())
		HIP_TRY(hipMemcpyAsync(*d, v.data(), v.size() * sizeof(T),
				       hipMemcpyHostToDevice, s));
	return 0;
}

static int sync_params(h264_amd_engine *e, const h264_amd_params *p)
{
	if (e->params_src == p && e->params_serial == p->serial)
		return 0;
	/* the source vectors may be reallocated by the caller afterwards */
	int r;
	if ((r = upload_vec(&e->d_sps, &e->cap_sps, p->sps, e->stream)) < 0)
		return r;
	if ((r = upload_vec(&e->d_pps, &e->cap_pps, p->pps, e->stream)) < 0)
		return r;
	if ((r = upload_vec(&e->d_fmo, &e->cap_fmo, p->fmo, e->stream)) < 0)
		return r;
	if ((r = upload_vec(&e->d_epochs, &e->cap_epochs, p->epochs, e->stream)) < 0)
		return r;
	HIP_TRY(hipStreamSynchronize(e->stream));
	e->params_src = p;
	e->params_serial = p->serial;
	return 0;
}

/* the macroblock-to-slice-group map the last h264_amd_decode_slices() launch
 * built for slice `index` of that launch */
extern "C" int h264_amd_slice_group_map(struct h264_amd_engine *e, uint32_t index,
					uint8_t *dst, size_t size)
{
	if (e == NULL || dst == NULL)
		return -EINVAL;
	if (index >= e->fmo_n || size > e->fmo_stride)
		return -ERANGE;
	HIP_TRY(hipSetDevice(e->device));
	HIP_TRY(hipMemcpyAsync(dst, e->d_fmo_map + (size_t)index * e->fmo_stride,
			       size, hipMemcpyDeviceToHost, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	return 0;
}

/* the same for `count` slices from `first` on, `size` bytes of each into rows
 * of `dst_pitch` bytes: one copy */
extern "C" int h264_amd_slice_group_maps(struct h264_amd_engine *e, uint32_t first,
					 uint32_t count, uint8_t *dst,
					 size_t dst_pitch, size_t size)
{
	if (e == NULL || dst == NULL || dst_pitch < size)
		return -EINVAL;
	if (first > e->fmo_n || count > e->fmo_n - first || size > e->fmo_stride)
		return -ERANGE;
	if (count == 0 || size == 0)
		return 0;
	HIP_TRY(hipSetDevice(e->device));
	HIP_TRY(hipMemcpy2DAsync(dst, dst_pitch,
				 e->d_fmo_map + (size_t)first * e->fmo_stride,
				 e->fmo_stride, size, count, hipMemcpyDeviceToHost,
				 e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	return 0;
}

extern "C" size_t h264_amd_slice_group_map_pitch(const struct h264_amd_engine *e)
{
	return e != NULL && e->fmo_n > 0 ? e->fmo_stride : 0;
}

/* ---- slice decode ------------------------------------------------------------ */

extern "C" int h264_amd_decode_slices(struct h264_amd_engine *e,
				      const struct h264_amd_params *params,
				      const struct h264_amd_decode_args *args)
{
	if (e == NULL || params == NULL || args == NULL)
		return -EINVAL;
	if (args->n == 0)
		return 0;
	if (args->d_buf == NULL || args->d_descs == NULL || args->d_results == NULL)
		return -EINVAL;
	/* the kernel uses 16-byte vector stores for header images and T2 records
	 * and dword accesses elsewhere: refuse what would fault on the device */
	if (((uintptr_t)args->d_descs & 7u) || ((uintptr_t)args->d_results & 15u) ||
	    ((uintptr_t)args->d_headers & 15u))
		return -EINVAL;
	if ((args->reader_flags & H264_AMD_READER_FLAGS_CABAC) &&
	    args->d_headers == NULL)
		return -EINVAL; /* the CABAC pass reads the header images */
	uint32_t stride = args->mb_stride;
	if (args->d_mb_out != NULL) {
		if ((uintptr_t)args->d_mb_out & 15u)
			return -EINVAL;
		if (args->tier != H264_AMD_TIER_NONE && args->mb_capacity == 0)
			return -EINVAL;
		if ((uint64_t)args->n * args->mb_capacity * stride >
		    (uint64_t)1 << 46)
			return -E2BIG;
		if (stride & 3u)
			return -EINVAL;
		if (args->tier == H264_AMD_TIER_T0 && stride < 4)
			return -EINVAL;
		if (args->tier == H264_AMD_TIER_T1 && stride < 64)
			return -EINVAL;
		if (args->tier == H264_AMD_TIER_T2 &&
		    (stride < h264_amd_t2_stride(3) || (stride & 15)))
			return -EINVAL;
		if (args->tier > H264_AMD_TIER_T2)
			return -EINVAL;
	}
	HIP_TRY(hipSetDevice(e->device));
	int r = sync_params(e, params);
	if (r < 0)
		return r;

	sd_args a;
	memset(&a, 0, sizeof(a));
	a.buf = args->d_buf;
	a.descs = args->d_descs;
	a.n = args->n;
	a.spsv = e->d_sps;
	a.ppsv = e->d_pps;
	a.epochs = e->d_epochs;
	a.n_sps = (uint32_t)params->sps.size();
	a.n_pps = (uint32_t)params->pps.size();
	a.n_epochs = (uint32_t)params->epochs.size();
	a.vlc = e->d_vlc;
	a.reader_flags = args->reader_flags;
	a.tier = args->d_mb_out ? args->tier : (uint32_t)H264_AMD_TIER_NONE;
	a.mb_capacity = args->mb_capacity;
	if (a.mb_capacity == 0 && a.tier == H264_AMD_TIER_NONE)
		a.mb_capacity = H264_AMD_MAX_SLICE_MBS; /* nothing to overflow */
	a.mb_stride = stride;
	a.results = args->d_results;
	a.headers = args->d_headers;
	a.mb_out = args->d_mb_out;
	a.buf_len = args->d_buf_len;
	/* row context only where slice data is walked, and never for more
	 * columns than H264_AMD_MAX_PIC_WIDTH_MBS: one damaged SPS must not make
	 * the whole batch unlaunchable (it used to: -E2BIG for every slice) */
	a.row_pitch = 1;
	if (args->reader_flags & H264_READER_FLAGS_SLICE_DATA)
		a.row_pitch = params->max_width_mbs < H264_AMD_MAX_PIC_WIDTH_MBS
				      ? params->max_width_mbs
				      : H264_AMD_MAX_PIC_WIDTH_MBS;
	/* slice groups: a macroblock-to-slice-group map per slice, built by the
	 * kernel behind the slice header (types 3..5 depend on it) */
	a.fmov = e->d_fmo;
	if ((args->reader_flags & H264_READER_FLAGS_SLICE_DATA) &&
	    params->n_fmo > 0) {
		uint32_t stride = params->max_pic_mbs < DEV_FMO_MAX_MBS
					  ? params->max_pic_mbs
					  : DEV_FMO_MAX_MBS;
		stride = (stride + 63u) & ~63u;
		size_t want = (size_t)a.n * stride;
		if (want > e->cap_fmo_map) {
			hipFree(e->d_fmo_map);
			e->d_fmo_map = NULL;
			e->cap_fmo_map = 0;
			HIP_TRY(hipMalloc((void **)&e->d_fmo_map, want));
			e->cap_fmo_map = want;
		}
		a.fmo_map = e->d_fmo_map;
		a.fmo_stride = stride;
	}
	e->fmo_stride = a.fmo_stride;
	e->fmo_n = a.fmo_map ? a.n : 0;

	a.row_mbaff = params->any_mbaff ? 1u : 0u;
	{
		uint32_t tail = SD_REC_BYTES +
				(a.row_mbaff ? SD_ROW_STRIDE_MBAFF : 16u) * a.row_pitch;
		if (tail < SD_STAGE_BYTES)
			tail = SD_STAGE_BYTES;
		a.wave_lds_bytes = (SD_STAGE_OFF + tail + 255u) & ~255u;
	}
	size_t lds = SD_TABS_BYTES + (size_t)SD_WAVES_PER_BLOCK * a.wave_lds_bytes;
	if (lds > 160u * 1024u)
		return -E2BIG;
	/* more than one round of 16 waves per CU: the 5-waves-per-SIMD build
	 * (H264_AMD_DECODE_W5=0/1 forces one or the other); a parameter set
	 * that allows MBAFF: the pair-aware build */
	bool w5 = a.n > 4096u;
	{
		const char *v = getenv("H264_AMD_DECODE_W5");
		if (v != NULL && (v[0] == '0' || v[0] == '1'))
			w5 = v[0] == '1';
	}
	void (*kern)(sd_args) = params->any_mbaff ? h264_amd_k_decode_slices_mbaff :
				w5 ? h264_amd_k_decode_slices_w5 :
				     h264_amd_k_decode_slices;
	const uint32_t per_simd = (kern == h264_amd_k_decode_slices_w5) ? 5u : 4u;
	const uint32_t wgs = (a.n + SD_WAVES_PER_BLOCK - 1) / SD_WAVES_PER_BLOCK;
	/* Slice data walked and more slices than a handful: launch positions
	 * stand for the slices longest first (h264_amd_k_order_slices on the
	 * device) -- the workgroup dispatcher then hands the long slices out
	 * first and fills the end of the launch with the short ones, whatever
	 * order the caller's descriptors are in. */
	const bool ordered = e->use_order && a.n > 64u &&
			     (args->reader_flags & H264_READER_FLAGS_SLICE_DATA);
	if (ordered) {
		if ((size_t)a.n > e->cap_order) {
			hipFree(e->d_order);
			e->d_order = NULL;
			e->cap_order = 0;
			HIP_TRY(hipMalloc((void **)&e->d_order,
					  (size_t)a.n * sizeof(uint32_t)));
			e->cap_order = (size_t)a.n;
		}
		a.order = e->d_order;
	}
	{
		/* Priority rotation (slice_decode.hip, "fair shares"): pays when
		 * the waves of the last round all end together -- a launch of
		 * more than one dispatch layer that fits the chip, or one of
		 * whole rounds (10240 slices: 133.9 -> 121.4 ms).  With a partly
		 * filled last round the oldest-first order is the better one: the
		 * early finishers hand their slots to it sooner (5632 slices:
		 * 96.2 against 100.6 ms). */
		const uint32_t cap = e->cu_count * 4u * per_simd; /* resident waves */
		const bool fits = a.n <= cap, whole = a.n % cap == 0;
		a.prio_div = wgs > e->cu_count && (fits || whole) ? e->cu_count : 0u;
	}
	if (lds > 48u * 1024u)
		HIP_TRY(hipFuncSetAttribute(
			(const void *)kern,
			hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	HIP_TRY(hipEventRecord(e->ev[FAM_DECODE][0], e->stream));
	if (ordered) {
		hipLaunchKernelGGL(h264_amd_k_order_slices, dim3(1), dim3(1024), 0,
				   e->stream, a.descs, a.n, e->d_order);
		HIP_TRY(hipGetLastError());
	}
	hipLaunchKernelGGL(kern, dim3(wgs), dim3(64 * SD_WAVES_PER_BLOCK), lds,
			   e->stream, a);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(e->ev[FAM_DECODE][1], e->stream));
	e->timed[FAM_DECODE] = true;

	if ((args->reader_flags & H264_AMD_READER_FLAGS_CABAC) &&
	    (args->reader_flags & H264_READER_FLAGS_SLICE_DATA)) {
		/* second pass over the slices the first one flagged as CABAC */
		cs_args c;
		memset(&c, 0, sizeof(c));
		c.buf = a.buf;
		c.descs = a.descs;
		c.n = a.n;
		c.spsv = a.spsv;
		c.ppsv = a.ppsv;
		c.epochs = a.epochs;
		c.n_sps = a.n_sps;
		c.n_pps = a.n_pps;
		c.n_epochs = a.n_epochs;
		c.tab = e->d_cabac;
		c.dtab = e->d_cabac_dec;
		c.tier = a.tier;
		c.mb_capacity = a.mb_capacity;
		c.mb_stride = a.mb_stride;
		c.results = a.results;
		c.headers = a.headers;
		c.mb_out = a.mb_out;
		c.row_pitch = a.row_pitch < H264_AMD_MAX_CABAC_WIDTH_MBS
				      ? a.row_pitch
				      : H264_AMD_MAX_CABAC_WIDTH_MBS;
		c.order = a.order;
		c.quirks = ((args->reader_flags & H264_AMD_READER_FLAGS_CABAC_X264_PRE151) ? 1u : 0u) |
			   ((args->reader_flags & H264_AMD_READER_FLAGS_CABAC_STD_QP) ? 2u : 0u);
		c.wave_lds_bytes =
			(CS_ROW_OFF + CS_ROW_STRIDE * c.row_pitch + 255u) & ~255u;
		{
			const uint32_t wgs = (c.n + CS_WAVES_PER_BLOCK - 1) / CS_WAVES_PER_BLOCK;
			const uint32_t cap = e->cu_count * 16u;
			c.prio_div = wgs > e->cu_count && (c.n <= cap || c.n % cap == 0)
					     ? e->cu_count
					     : 0u;
		}
		size_t clds = CS_MN_BYTES + (size_t)CS_WAVES_PER_BLOCK * c.wave_lds_bytes;
		if (clds > 160u * 1024u)
			return -E2BIG;
		if (clds > 48u * 1024u)
			HIP_TRY(hipFuncSetAttribute(
				(const void *)h264_amd_k_decode_cabac,
				hipFuncAttributeMaxDynamicSharedMemorySize, (int)clds));
		HIP_TRY(hipEventRecord(e->ev[FAM_CABAC][0], e->stream));
		hipLaunchKernelGGL(h264_amd_k_decode_cabac,
				   dim3((c.n + CS_WAVES_PER_BLOCK - 1) / CS_WAVES_PER_BLOCK),
				   dim3(64 * CS_WAVES_PER_BLOCK), clds, e->stream, c);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipEventRecord(e->ev[FAM_CABAC][1], e->stream));
		e->timed[FAM_CABAC] = true;
	}
	return 0;
}

#include "engine_more.inc"
#include "group.inc"
