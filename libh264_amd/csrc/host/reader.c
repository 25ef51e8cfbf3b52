/**
 * reader.c -- struct h264_reader, drop-in for the reference's
 * src/h264_reader.c:58-162 and the NAL dispatch of src/h264_syntax.h:1446-1604,
 * re-organised around the GPU:
 *
 *   1. the caller's buffer goes to HBM once; h264_amd_scan() builds the NAL
 *      table there (h264_reader_parse only);
 *   2. a host pre-pass walks the table, parses the parameter sets (a few bytes
 *      each) into the reader's versioned snapshot and tags every slice NAL
 *      with the epoch current at its position;
 *   3. ALL slice NAL units of the buffer are decoded by ONE
 *      h264_amd_decode_slices() launch -- one slice per wavefront -- into
 *      device buffers (up to RD_DEVICE_BYTES of records per launch: HBM is
 *      large, nothing ties the launch to the host's page); the records come
 *      to the host a page of RD_CHUNK_BYTES at a time as the replay reaches
 *      them;
 *   4. the callbacks are replayed on the host in bitstream order from the
 *      kernel's records, together with the per-NAL context updates and the
 *      access-unit detection of 7.4.1.2.4, so that a callback observes the
 *      same struct h264_ctx state as with the reference.
 *
 * Slices are never parsed on the CPU.  Without a GPU h264_reader_parse() and
 * h264_reader_parse_nalu() on a slice return -ENODEV.
 */
#include "h264_priv.h"
#include <stdio.h>
#include <time.h>

/* H264_AMD_READER_TRACE=1: stage times of h264_reader_parse on stderr */
static double rd_now(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static int rd_trace(void)
{
	static int on = -1;
	if (on < 0)
		on = getenv("H264_AMD_READER_TRACE") != NULL;
	return on;
}


#define RD_CHUNK_BYTES ((size_t)64 << 20) /* host image of one page of records */
#define RD_DEVICE_BYTES ((size_t)8 << 30) /* records of one launch, on the device */
#define RD_MAX_EPOCHS 1024 /* compact the snapshot beyond this */

/* what the pre-pass decided for one NAL unit */
struct rd_plan {
	int64_t slice; /* index into the slice batch or -1 */
};

/* the slice batch of one buffer: launches and host pages */
struct rd_batch {
	struct h264_amd_engine *eng;
	size_t n; /* slices in the buffer */
	size_t launch_n; /* slices per launch (a multiple of chunk_n, or n) */
	size_t llo, lhi; /* slices whose records are on the device: [llo, lhi) */
	size_t chunk_n; /* slices per host page */
	size_t lo, hi; /* slices currently on the host: [lo, hi) */
	size_t plo, phi; /* slices of the page in flight to the other host page */
	uint32_t tier, mb_cap, flags;
};

/* one buffer of a host page: page-locked if the runtime gives it (the copy
 * from the device then runs at link speed and truly beside the replay),
 * plain memory otherwise */
static int page_buf_grow(struct h264_amd_engine *eng, void **p, size_t *cap,
			 int *pinned, size_t want)
{
	void *np = NULL;
	if (want <= *cap && *p != NULL)
		return 0;
	if (*p != NULL) {
		if (*pinned)
			h264_amd_host_free(eng, *p);
		else
			free(*p);
	}
	*p = NULL;
	*cap = 0;
	*pinned = 0;
	if (h264_amd_host_alloc(eng, want ? want : 1, &np) == 0 && np != NULL) {
		*pinned = 1;
	} else {
		np = malloc(want ? want : 1);
		if (np == NULL)
			return -ENOMEM;
	}
	*p = np;
	*cap = want;
	return 0;
}

static int page_grow(struct h264_amd_engine *eng, struct rd_host_page *pg, size_t cn,
		     size_t mb_bytes)
{
	int res = page_buf_grow(eng, (void **)&pg->results, &pg->cap_results,
				&pg->pinned_results, cn * sizeof(*pg->results));
	if (res == 0)
		res = page_buf_grow(eng, (void **)&pg->headers, &pg->cap_headers,
				    &pg->pinned_headers, cn * sizeof(*pg->headers));
	if (res == 0 && mb_bytes)
		res = page_buf_grow(eng, (void **)&pg->mb, &pg->cap_mb,
				    &pg->pinned_mb, mb_bytes);
	return res;
}

static void page_release(struct h264_reader *rd)
{
	struct h264_amd_engine *eng = NULL;
	size_t i;
	int have = 0;
	for (i = 0; i < 2; i++)
		have |= rd->page[i].pinned_results | rd->page[i].pinned_headers |
			rd->page[i].pinned_mb;
	if (have && h264_host_engine(&eng) < 0)
		eng = NULL; /* (no device any more: the runtime went first) */
	if (eng != NULL)
		h264_amd_download_wait(eng);
	for (i = 0; i < 2; i++) {
		struct rd_host_page *pg = &rd->page[i];
		void *ptr[3] = {pg->results, pg->headers, pg->mb};
		int pin[3] = {pg->pinned_results, pg->pinned_headers, pg->pinned_mb};
		size_t k;
		for (k = 0; k < 3; k++) {
			if (ptr[k] == NULL)
				continue;
			if (!pin[k])
				free(ptr[k]);
			else if (eng != NULL)
				h264_amd_host_free(eng, ptr[k]);
		}
		memset(pg, 0, sizeof(*pg));
	}
	rd->h_results = NULL;
	rd->h_headers = NULL;
	rd->h_mb = NULL;
}

/* queue the copies of page [lo, lo + cn) of the launch into host page `pg` */
static int page_start(struct h264_reader *rd, struct rd_batch *b,
		      struct rd_host_page *pg, size_t lo, size_t cn)
{
	size_t mb_bytes = b->tier == H264_AMD_TIER_T0 ? cn * b->mb_cap * 4 : 0;
	size_t off = lo - b->llo; /* first slice of the page within the launch */
	int res = page_grow(b->eng, pg, cn, mb_bytes);
	if (res < 0)
		return res;
	res = h264_amd_download_async(
		b->eng, pg->results,
		(const struct h264_amd_slice_result *)rd->d_results + off,
		cn * sizeof(*pg->results));
	if (res == 0)
		res = h264_amd_download_async(
			b->eng, pg->headers,
			(const struct h264_slice_header *)rd->d_headers + off,
			cn * sizeof(*pg->headers));
	if (res == 0 && mb_bytes)
		res = h264_amd_download_async(b->eng, pg->mb,
					      (const uint8_t *)rd->d_mb +
						      off * b->mb_cap * 4,
					      mb_bytes);
	return res;
}

int h264_reader_new(const struct h264_ctx_cbs *cbs, void *userdata,
		    struct h264_reader **ret_obj)
{
	struct h264_reader *rd;
	int res;
	if (ret_obj == NULL)
		return -EINVAL;
	*ret_obj = NULL;
	if (cbs == NULL)
		return -EINVAL;
	rd = calloc(1, sizeof(*rd));
	if (rd == NULL)
		return -ENOMEM;
	rd->cbs = *cbs;
	rd->userdata = userdata;
	res = h264_ctx_new(&rd->ctx);
	if (res < 0) {
		free(rd);
		return res;
	}
	*ret_obj = rd;
	return 0;
}

int h264_reader_destroy(struct h264_reader *rd)
{
	struct h264_amd_engine *eng = NULL;
	size_t i;
	if (rd == NULL)
		return 0;
	if (rd->d_buf != NULL && h264_host_engine(&eng) == 0) {
		void *d[] = {rd->d_buf,	    rd->d_nals,	   rd->d_descs,
			     rd->d_results, rd->d_headers, rd->d_mb,
			     rd->d_retry,   rd->d_retry_mb};
		h264_host_lock();
		for (i = 0; i < H264_ARRAY_SIZE(d); i++) {
			if (d[i] != NULL)
				h264_amd_free(eng, d[i]);
		}
		h264_host_unlock();
	}
	if (rd->params != NULL)
		h264_amd_params_destroy(rd->params);
	page_release(rd);
	free(rd->h_big);
	free(rd->h_fmo);
	free(rd->h_fmo_big);
	h264_ctx_destroy(rd->ctx);
	free(rd);
	return 0;
}

struct h264_ctx *h264_reader_get_ctx(struct h264_reader *rd)
{
	return rd != NULL ? rd->ctx : NULL;
}

int h264_reader_stop(struct h264_reader *rd)
{
	if (rd == NULL)
		return -EINVAL;
	rd->stop = 1;
	return 0;
}

/* ---- parameter sets: shared by the pre-pass and the replay --------------- */

typedef const struct h264_sps *(*sps_lookup)(void *arg, uint32_t sps_id);

static const struct h264_sps *sps_from_params(void *arg, uint32_t sps_id)
{
	return h264_amd_params_sps(arg, sps_id);
}

/* the replay's lookup ACTIVATES the SPS, as the reference does while reading
 * a PPS (src/h264_syntax.h:505-512) */
static const struct h264_sps *sps_from_ctx(void *arg, uint32_t sps_id)
{
	struct h264_ctx *ctx = arg;
	if (h264_ctx_set_active_sps(ctx, sps_id) < 0)
		return NULL;
	return ctx->sps;
}

/* bs sits behind the NAL header byte */
static int read_pps(struct h264_bitstream *bs, sps_lookup find, void *arg,
		    struct h264_pps *pps)
{
	struct h264_syn s = {bs, 0, 0};
	const struct h264_sps *sps;
	memset(pps, 0, sizeof(*pps));
	if (h264_syn_pps_ids(&s, pps) < 0)
		return s.err;
	sps = find(arg, pps->seq_parameter_set_id);
	if (sps == NULL)
		return -EINVAL;
	return h264_syn_pps_body(&s, sps, pps);
}

static int read_sps(struct h264_bitstream *bs, struct h264_sps *sps)
{
	struct h264_syn s = {bs, 0, 0};
	memset(sps, 0, sizeof(*sps));
	sps->chroma_format_idc = 1; /* inferred when absent, 7.4.2.1.1 */
	return h264_syn_sps(&s, sps);
}

/* ---- step 2: pre-pass ------------------------------------------------------ */

static int sync_params(struct h264_reader *rd)
{
	int res;
	if (rd->params == NULL) {
		res = h264_amd_params_new(&rd->params);
		if (res < 0)
			return res;
		rd->ps_gen_seen = ~(uint64_t)0;
	}
	/* the caller may have fed parameter sets through h264_ctx_set_sps /
	 * h264_ctx_set_pps on our ctx since the last call */
	if (rd->ps_gen_seen != rd->ctx->ps_generation ||
	    h264_amd_params_epoch(rd->params) > RD_MAX_EPOCHS) {
		res = h264_host_params_from_ctx(rd->params, rd->ctx);
		if (res < 0)
			return res;
		rd->ps_gen_seen = rd->ctx->ps_generation;
	}
	return 0;
}

static int is_slice_nal(const uint8_t *p, size_t len)
{
	uint32_t t;
	if (len == 0 || (p[0] & 0x80))
		return 0;
	t = p[0] & 0x1f;
	return t == H264_NALU_TYPE_SLICE || t == H264_NALU_TYPE_SLICE_IDR;
}

static int prepass(struct h264_reader *rd, const uint8_t *buf,
		   const struct h264_amd_nal *nals, size_t n,
		   struct rd_plan *plan, struct h264_amd_slice_desc *descs,
		   size_t *n_slices)
{
	size_t i, ns = 0;
	int res;
	for (i = 0; i < n; i++) {
		const uint8_t *p = buf + nals[i].start;
		size_t len = (size_t)(nals[i].end - nals[i].start);
		struct h264_bitstream bs;
		uint32_t type, ref;
		plan[i].slice = -1;
		if (len == 0 || (p[0] & 0x80))
			continue;
		type = p[0] & 0x1f;
		ref = (p[0] >> 5) & 3;
		if (type == H264_NALU_TYPE_SLICE ||
		    type == H264_NALU_TYPE_SLICE_IDR) {
			res = h264_amd_params_epoch(rd->params);
			if (res < 0)
				return res;
			if (len > UINT32_MAX)
				return -E2BIG;
			descs[ns].offset = nals[i].start;
			descs[ns].size = (uint32_t)len;
			descs[ns].epoch = (uint32_t)res;
			plan[i].slice = (int64_t)ns++;
		} else if (type == H264_NALU_TYPE_SPS && ref != 0) {
			struct h264_sps sps;
			h264_bs_cinit(&bs, p + 1, len - 1, 1);
			if (read_sps(&bs, &sps) == 0 &&
			    sps.seq_parameter_set_id < 32) {
				res = h264_amd_params_put_sps(rd->params, &sps);
				if (res < 0)
					return res;
			}
		} else if (type == H264_NALU_TYPE_PPS && ref != 0) {
			struct h264_pps pps;
			h264_bs_cinit(&bs, p + 1, len - 1, 1);
			if (read_pps(&bs, sps_from_params, rd->params, &pps) == 0 &&
			    pps.pic_parameter_set_id < 256) {
				res = h264_amd_params_put_pps(rd->params, &pps);
				if (res < 0)
					return res;
			}
		}
	}
	*n_slices = ns;
	return 0;
}

/* ---- step 3: batched slice decode ------------------------------------------ */

static int host_grow(void **p, size_t *cap, size_t want)
{
	void *np;
	if (want <= *cap && *p != NULL)
		return 0;
	free(*p);
	*p = NULL;
	*cap = 0;
	np = malloc(want ? want : 1);
	if (np == NULL)
		return -ENOMEM;
	*p = np;
	*cap = want;
	return 0;
}

static int batch_setup(struct h264_reader *rd, struct rd_batch *b,
		       const struct h264_amd_slice_desc *descs, size_t ns,
		       uint32_t flags)
{
	size_t per_slice;
	const char *env;
	int res;
	memset(b, 0, sizeof(*b));
	res = h264_host_engine(&b->eng);
	if (res < 0)
		return res;
	/* (a parse that ended early may have left a page on its way) */
	res = h264_amd_download_wait(b->eng);
	if (res < 0)
		return res;
	b->n = ns;
	b->flags = flags;
	/* macroblock records are only materialised for a caller that looks at
	 * them; the parse itself (status, mb_count) always runs */
	b->tier = ((flags & H264_READER_FLAGS_SLICE_DATA) &&
		   rd->cbs.slice_data_mb != NULL)
			  ? H264_AMD_TIER_T0
			  : H264_AMD_TIER_NONE;
	b->mb_cap = h264_amd_params_max_mbs(rd->params);
	if (b->mb_cap == 0)
		b->mb_cap = 1;
	per_slice = sizeof(struct h264_slice_header) +
		    sizeof(struct h264_amd_slice_result) +
		    (b->tier == H264_AMD_TIER_T0 ? (size_t)4 * b->mb_cap : 0);
	env = getenv("H264_AMD_READER_CHUNK_BYTES");
	b->chunk_n = (env != NULL && atol(env) > 0 ? (size_t)atol(env)
						    : RD_CHUNK_BYTES) /
		     per_slice;
	if (b->chunk_n == 0)
		b->chunk_n = 1;
	if (b->chunk_n > ns)
		b->chunk_n = ns;
	/* the launch: everything at once where the device has the room.  Slice
	 * group maps stay with the launch that built them and a callback may
	 * launch again (a dump of the slice does): such streams are launched a
	 * page at a time, maps fetched right behind the launch */
	env = getenv("H264_AMD_READER_DEVICE_BYTES");
	b->launch_n = (env != NULL && atol(env) > 0 ? (size_t)atol(env)
						     : RD_DEVICE_BYTES) /
		      per_slice;
	if (h264_amd_params_has_slice_groups(rd->params) || b->launch_n < b->chunk_n)
		b->launch_n = b->chunk_n;
	b->launch_n -= b->launch_n % b->chunk_n;
	if (b->launch_n > ns)
		b->launch_n = ns;
	res = h264_host_grow(&rd->d_descs, &rd->cap_descs, ns * sizeof(*descs));
	if (res == 0)
		res = h264_amd_upload(b->eng, rd->d_descs, descs,
				      ns * sizeof(*descs));
	return res;
}

/* The slice group maps of the launch just made, to the host: the engine keeps
 * them only until its next launch, and a callback may well cause one (a dump
 * of the slice does). */
static int fetch_maps(struct h264_reader *rd, struct h264_amd_engine *eng,
		      size_t count, uint8_t **h_maps, size_t *hcap)
{
	size_t pitch = h264_amd_slice_group_map_pitch(eng);
	int res;
	rd->fmo_pitch = pitch;
	if (pitch == 0)
		return 0;
	res = host_grow((void **)h_maps, hcap, count * pitch);
	if (res < 0)
		return res;
	return h264_amd_slice_group_maps(eng, 0, (uint32_t)count, *h_maps, pitch,
					 pitch);
}

/* make slice `idx` available in the host images: launch the part of the batch
 * it belongs to unless its records are on the device already, then bring its
 * page over */
static int batch_fetch_do(struct h264_reader *rd, struct rd_batch *b, size_t idx);
static int batch_fetch(struct h264_reader *rd, struct rd_batch *b, size_t idx)
{
	if (idx >= b->lo && idx < b->hi)
		return 0;
	if (rd_trace()) {
		double t = rd_now();
		int res = batch_fetch_do(rd, b, idx);
		rd->trace_wait_ms += rd_now() - t;
		return res;
	}
	return batch_fetch_do(rd, b, idx);
}

static int batch_fetch_do(struct h264_reader *rd, struct rd_batch *b, size_t idx)
{
	struct h264_amd_decode_args a;
	size_t lo, cn;
	int res;
	if (idx < b->llo || idx >= b->lhi) {
		size_t llo = idx - idx % b->launch_n;
		if (b->phi > b->plo) {
			/* (the device buffers are about to be overwritten) */
			res = h264_amd_download_wait(b->eng);
			if (res < 0)
				return res;
			b->plo = b->phi = 0;
		}
		size_t ln = b->n - llo < b->launch_n ? b->n - llo : b->launch_n;
		size_t lmb = b->tier == H264_AMD_TIER_T0 ? ln * b->mb_cap * 4 : 0;
		res = h264_host_grow(&rd->d_results, &rd->cap_results,
				     ln * sizeof(*rd->h_results));
		if (res == 0)
			res = h264_host_grow(&rd->d_headers, &rd->cap_headers,
					     ln * sizeof(*rd->h_headers));
		if (res == 0 && lmb)
			res = h264_host_grow(&rd->d_mb, &rd->cap_mb, lmb);
		if (res < 0)
			return res;
		memset(&a, 0, sizeof(a));
		a.d_buf = rd->d_buf;
		a.d_descs = (const struct h264_amd_slice_desc *)rd->d_descs + llo;
		a.n = (uint32_t)ln;
		a.reader_flags = b->flags;
		a.tier = b->tier;
		a.mb_capacity = b->mb_cap;
		a.d_results = rd->d_results;
		a.d_headers = rd->d_headers;
		a.d_mb_out = lmb ? rd->d_mb : NULL;
		a.mb_stride = lmb ? 4 : 0;
		{
			/* H264_AMD_DEVICES names several devices: the launch is
			 * spread over the engine group (a small one stays on the
			 * first device, see h264_amd_group_plan).  Slice group maps
			 * live with the engine that built them: streams with slice
			 * groups stay on one device. */
			struct h264_amd_group *grp = h264_host_group();
			if (grp != NULL &&
			    !h264_amd_params_has_slice_groups(rd->params))
				res = h264_amd_group_decode_slices(grp, rd->params, &a,
								   NULL, NULL);
			else
				res = h264_amd_decode_slices(b->eng, rd->params, &a);
		}
		if (res < 0)
			return res;
		b->llo = llo;
		b->lhi = llo + ln;
		b->lo = b->hi = 0;
	}
	lo = idx - idx % b->chunk_n;
	cn = b->lhi - lo < b->chunk_n ? b->lhi - lo : b->chunk_n;
	if (b->phi > b->plo) {
		/* a page is on its way: wait for it, it is the one asked for
		 * unless the caller jumped */
		res = h264_amd_download_wait(b->eng);
		if (res < 0)
			return res;
		if (b->plo == lo)
			rd->cur_page ^= 1;
		else
			b->plo = b->phi = 0;
	}
	if (b->plo != lo || b->phi == b->plo) {
		res = page_start(rd, b, &rd->page[rd->cur_page], lo, cn);
		if (res == 0)
			res = h264_amd_download_wait(b->eng);
		if (res < 0)
			return res;
	}
	b->plo = b->phi = 0;
	rd->h_results = rd->page[rd->cur_page].results;
	rd->h_headers = rd->page[rd->cur_page].headers;
	rd->h_mb = rd->page[rd->cur_page].mb;
	/* maps: only built for streams with slice groups, which launch a page
	 * at a time (off = 0) */
	res = fetch_maps(rd, b->eng, cn, &rd->h_fmo, &rd->hcap_fmo);
	if (res < 0)
		return res;
	/* the next page of this launch starts its way to the host now and
	 * travels while the callbacks of this one are replayed */
	if (lo + cn < b->lhi) {
		size_t ncn = b->lhi - (lo + cn) < b->chunk_n ? b->lhi - (lo + cn)
							     : b->chunk_n;
		res = page_start(rd, b, &rd->page[rd->cur_page ^ 1], lo + cn, ncn);
		if (res < 0)
			return res;
		b->plo = lo + cn;
		b->phi = lo + cn + ncn;
	}
	b->lo = lo;
	b->hi = lo + cn;
	return 0;
}

/* A damaged slice can run past the picture: the reference's macroblock table
 * simply grows (src/h264_macroblock.c:285-316).  The batch is sized for sane
 * pictures; a slice that overflowed it is decoded again on its own with 8x
 * the room until it fits.  RD_MAX_MB_CAP bounds what a corrupt mb_skip_run can
 * ask for (the reference would call back 2^32 times); beyond it the slice
 * fails with -EOVERFLOW. */
#define RD_MAX_MB_CAP H264_AMD_MAX_SLICE_MBS

static int retry_big_slice(struct h264_reader *rd, struct rd_batch *b,
			   size_t slice, struct h264_amd_slice_result *r,
			   const uint32_t **t0)
{
	struct h264_amd_decode_args a;
	uint32_t cap = b->mb_cap;
	int res;
	do {
		size_t mb_bytes;
		cap = cap > RD_MAX_MB_CAP / 8 ? RD_MAX_MB_CAP : cap * 8;
		mb_bytes = b->tier == H264_AMD_TIER_T0 ? (size_t)cap * 4 : 0;
		res = h264_host_grow(&rd->d_retry, &rd->cap_retry,
				     4096 + sizeof(struct h264_slice_header));
		if (res == 0 && mb_bytes)
			res = h264_host_grow(&rd->d_retry_mb, &rd->cap_retry_mb,
					     mb_bytes);
		if (res < 0)
			return res;
		memset(&a, 0, sizeof(a));
		a.d_buf = rd->d_buf;
		a.d_descs = (const struct h264_amd_slice_desc *)rd->d_descs + slice;
		a.n = 1;
		a.reader_flags = b->flags;
		a.tier = b->tier;
		a.mb_capacity = cap;
		a.d_results = rd->d_retry;
		a.d_headers = (struct h264_slice_header *)((uint8_t *)rd->d_retry + 4096);
		a.d_mb_out = mb_bytes ? rd->d_retry_mb : NULL;
		a.mb_stride = mb_bytes ? 4 : 0;
		res = h264_amd_decode_slices(b->eng, rd->params, &a);
		if (res == 0)
			res = h264_amd_download(b->eng, r, rd->d_retry, sizeof(*r));
		if (res < 0)
			return res;
	} while (r->status == H264_AMD_E_MB_OVERFLOW && cap < RD_MAX_MB_CAP);
	/* this slice's map is the only one of that launch; the batch's are on
	 * the host already */
	res = fetch_maps(rd, b->eng, 1, &rd->h_fmo_big, &rd->hcap_fmo_big);
	if (res < 0)
		return res;
	*t0 = NULL;
	if (b->tier == H264_AMD_TIER_T0 && r->mb_count > 0) {
		size_t bytes = (size_t)r->mb_count * 4;
		res = host_grow((void **)&rd->h_big, &rd->hcap_big, bytes);
		if (res == 0)
			res = h264_amd_download(b->eng, rd->h_big, rd->d_retry_mb,
						bytes);
		if (res < 0)
			return res;
		*t0 = rd->h_big;
	}
	return 0;
}

/* ---- step 4: replay -------------------------------------------------------- */

/* Addresses of a slice's macroblocks under slice groups: the reference walks
 * its map with h264_next_mb_addr (src/h264_fmo.c:307-319); the map is the one
 * the kernel built for this slice, fetched behind the launch (fetch_maps). */
static int slice_addresses(struct h264_reader *rd, const uint8_t *map,
			   const struct h264_amd_slice_result *r, uint32_t **ret)
{
	struct h264_ctx *ctx = rd->ctx;
	uint32_t pic = ctx->derived.PicSizeInMbs, m;
	uint32_t cur = r->first_mb_in_slice * (1u + (uint32_t)ctx->derived.MbaffFrameFlag);
	uint32_t *addrs;
	if (map == NULL || pic > rd->fmo_pitch)
		return -EPROTO;
	addrs = malloc((r->mb_count ? r->mb_count : 1) * sizeof(*addrs));
	if (addrs == NULL)
		return -ENOMEM;
	for (m = 0; m < r->mb_count; m++) {
		uint32_t i = cur + 1;
		addrs[m] = cur;
		if (i < pic && cur < pic) {
			uint8_t g = map[cur];
			while (i < pic && map[i] != g)
				i++;
		}
		cur = i;
	}
	*ret = addrs;
	return 0;
}

static int replay_slice(struct h264_reader *rd, const uint8_t *nal, size_t len,
			const struct h264_amd_slice_result *r,
			const struct h264_slice_header *sh, const uint32_t *t0,
			const uint8_t *group_map)
{
	struct h264_ctx *ctx = rd->ctx;
	const struct h264_ctx_cbs *cbs = &rd->cbs;
	uint32_t m;

	if (!(r->flags & H264_AMD_SR_HEADER_OK)) {
		/* the reference activates the PPS the moment its id has been
		 * read, before the rest of the header gets a chance to fail
		 * (src/h264_syntax.h:1256-1262): keep that side effect */
		struct h264_bitstream bs;
		uint32_t v[3] = {0, 0, 0};
		h264_bs_cinit(&bs, nal + 1, len - 1, 1);
		if (h264_bs_read_bits_ue(&bs, &v[0]) >= 0 &&
		    h264_bs_read_bits_ue(&bs, &v[1]) >= 0 &&
		    h264_bs_read_bits_ue(&bs, &v[2]) >= 0)
			h264_ctx_set_active_pps(ctx, v[2]);
		ctx->slice.hdr_len = 0;
		return r->status < 0 ? r->status : -EIO;
	}

	if (h264_ctx_set_active_pps(ctx, sh->pic_parameter_set_id) < 0)
		return -EINVAL;
	h264_ctx_set_slice_header(ctx, sh);
	ctx->slice.hdr_len = r->hdr_len_bits;
	ctx->slice.rawdata.partial = r->partial;
	ctx->slice.rawdata.partialbits = r->partialbits;
	ctx->slice.rawdata.buf = nal + r->rawdata_off;
	ctx->slice.rawdata.len = r->rawdata_len;

	if (r->flags & H264_AMD_SR_DATA_PARSED) {
		if (cbs->slice_data_begin)
			cbs->slice_data_begin(ctx, &ctx->slice.hdr, rd->userdata);
		if (cbs->slice_data_mb) {
			uint32_t *addrs = NULL;
			/* macroblock pairs are numbered 2n, 2n + 1
			 * (ref src/h264_cavlc.c:725) */
			uint32_t first = r->first_mb_in_slice *
					 (1u + (uint32_t)ctx->derived.MbaffFrameFlag);
			if (ctx->pps->num_slice_groups_minus1 > 0) {
				int res = slice_addresses(rd, group_map, r, &addrs);
				if (res < 0)
					return res;
			}
			for (m = 0; m < r->mb_count; m++)
				cbs->slice_data_mb(ctx, &ctx->slice.hdr,
						   addrs ? addrs[m] : first + m,
						   t0[m] & 0xffu, rd->userdata);
			free(addrs);
		}
		if (r->status < 0)
			return r->status;
		if (cbs->slice_data_end)
			cbs->slice_data_end(ctx, &ctx->slice.hdr, r->mb_count,
					    rd->userdata);
	} else if (r->status < 0) {
		return r->status;
	}
	return 0;
}

/* One NAL unit against the context: the READ instantiation of the NAL
 * dispatch (ref src/h264_syntax.h:1446-1604), slices fed from the kernel. */
struct rd_host_slice {
	struct h264_amd_slice_result r;
	struct h264_slice_header sh;
};

static int replay_nal(struct h264_reader *rd, struct rd_batch *b,
		      const uint8_t *nal, size_t len, int64_t slice,
		      const struct rd_host_slice *hs)
{
	struct h264_ctx *ctx = rd->ctx;
	const struct h264_ctx_cbs *cbs = &rd->cbs;
	void *ud = rd->userdata;
	struct h264_bitstream bs;
	struct h264_syn s = {&bs, 0, 0};
	uint32_t t;
	int res;

	h264_ctx_clear_nalu(ctx);
	h264_bs_cinit(&bs, nal, len, 1);
	if (h264_syn_nalu_header(&s, &ctx->nalu.hdr) < 0)
		return s.err;
	ctx->nalu.type = ctx->nalu.hdr.nal_unit_type;
	t = ctx->nalu.type;
	if (cbs->nalu_begin)
		cbs->nalu_begin(ctx, t, nal, len, &ctx->nalu.hdr, ud);

	switch (t) {
	case H264_NALU_TYPE_SLICE:
	case H264_NALU_TYPE_SLICE_IDR: {
		const struct h264_amd_slice_result *r;
		struct h264_amd_slice_result big;
		const uint32_t *t0;
		const uint8_t *group_map;
		size_t k;
		if (hs != NULL) {
			/* header parsed on the host (slicehdr.c) */
			res = replay_slice(rd, nal, len, &hs->r, &hs->sh, NULL, NULL);
			if (res < 0)
				return res;
			if (cbs->slice)
				cbs->slice(ctx, nal, len, &ctx->slice.hdr, ud);
			break;
		}
		if (slice < 0 || b == NULL)
			return -EPROTO;
		res = batch_fetch(rd, b, (size_t)slice);
		if (res < 0)
			return res;
		k = (size_t)slice - b->lo;
		r = &rd->h_results[k];
		t0 = b->tier == H264_AMD_TIER_T0 ? rd->h_mb + k * b->mb_cap
						 : NULL;
		group_map = rd->fmo_pitch ? rd->h_fmo + k * rd->fmo_pitch : NULL;
		if (r->status == H264_AMD_E_MB_OVERFLOW) {
			res = retry_big_slice(rd, b, (size_t)slice, &big, &t0);
			if (res < 0)
				return res;
			r = &big;
			group_map = rd->fmo_pitch ? rd->h_fmo_big : NULL;
		}
		res = replay_slice(rd, nal, len, r, &rd->h_headers[k], t0,
				   group_map);
		if (res < 0)
			return res;
		if (cbs->slice)
			cbs->slice(ctx, nal, len, &ctx->slice.hdr, ud);
		break;
	}
	case H264_NALU_TYPE_SEI:
		res = h264_syn_sei_read(&bs, ctx, cbs, ud);
		if (res < 0)
			return res;
		break;
	case H264_NALU_TYPE_SPS: {
		struct h264_sps sps;
		if (ctx->nalu.hdr.nal_ref_idc == 0)
			return -EIO;
		res = read_sps(&bs, &sps);
		if (res < 0)
			return res;
		res = h264_ctx_set_sps(ctx, &sps);
		if (res < 0)
			return res;
		if (cbs->sps)
			cbs->sps(ctx, nal, len, ctx->sps, ud);
		break;
	}
	case H264_NALU_TYPE_PPS: {
		struct h264_pps pps;
		if (ctx->nalu.hdr.nal_ref_idc == 0)
			return -EIO;
		res = read_pps(&bs, sps_from_ctx, ctx, &pps);
		if (res < 0)
			return res;
		res = h264_ctx_set_pps(ctx, &pps);
		if (res < 0)
			return res;
		if (cbs->pps)
			cbs->pps(ctx, nal, len, ctx->pps, ud);
		break;
	}
	case H264_NALU_TYPE_AUD:
		if (ctx->nalu.hdr.nal_ref_idc != 0)
			return -EIO;
		if (h264_syn_aud(&s, &ctx->aud) < 0)
			return s.err;
		if (cbs->aud)
			cbs->aud(ctx, nal, len, &ctx->aud, ud);
		break;
	case H264_NALU_TYPE_FILLER:
		if (ctx->nalu.hdr.nal_ref_idc != 0)
			return -EIO;
		if (h264_syn_filler(&s, &ctx->filler_len) < 0)
			return s.err;
		break;
	default:
		/* data partitions and everything else: header only */
		ctx->nalu.unknown = 1;
		break;
	}

	/* 7.4.1.2.4: does this NAL unit close the previous access unit?
	 * (ref src/h264_syntax.h:1572-1592) */
	if ((ctx->nalu.is_prev_vcl || ctx->nalu.is_prev_filler) &&
	    (t == H264_NALU_TYPE_AUD || t == H264_NALU_TYPE_SPS ||
	     t == H264_NALU_TYPE_PPS || t == H264_NALU_TYPE_SEI ||
	     (t >= 14 && t <= 18) || ctx->nalu.is_first_vcl)) {
		if (cbs->au_end)
			cbs->au_end(ctx, ud);
	}
	ctx->nalu.is_prev_vcl =
		(t == H264_NALU_TYPE_SLICE || t == H264_NALU_TYPE_SLICE_IDR);
	ctx->nalu.is_prev_filler = (t == H264_NALU_TYPE_FILLER);

	if (cbs->nalu_end)
		cbs->nalu_end(ctx, t, nal, len, &ctx->nalu.hdr, ud);
	return 0;
}

/* ---- the whole pipeline over a NAL table ----------------------------------- */

/* `stream`: honour h264_reader_stop() between NAL units and report the
 * consumed offset.  Returns the status of the LAST NAL unit replayed. */
static int run(struct h264_reader *rd, uint32_t flags, const uint8_t *buf,
	       const struct h264_amd_nal *nals, size_t n, int stream,
	       size_t *off, int *fatal)
{
	struct rd_plan *plan;
	struct h264_amd_slice_desc *descs;
	struct rd_batch batch, *b = NULL;
	size_t ns = 0, i;
	int res, last = 0;

	*fatal = 0;
	plan = malloc((n ? n : 1) * sizeof(*plan));
	descs = malloc((n ? n : 1) * sizeof(*descs));
	double t0 = rd_trace() ? rd_now() : 0, t1 = 0, t2 = 0;
	res = (plan == NULL || descs == NULL) ? -ENOMEM : sync_params(rd);
	if (res == 0)
		res = prepass(rd, buf, nals, n, plan, descs, &ns);
	if (rd_trace())
		t1 = rd_now();
	if (res == 0 && ns > 0) {
		res = batch_setup(rd, &batch, descs, ns, flags);
		b = &batch;
	}
	if (rd_trace())
		t2 = rd_now();
	if (res < 0) {
		*fatal = res;
		goto out;
	}

	for (i = 0; i < n && !(stream && rd->stop); i++) {
		last = replay_nal(rd, b, buf + nals[i].start,
				  (size_t)(nals[i].end - nals[i].start),
				  plan[i].slice, NULL);
		if (last == -ENODEV || last == -ENOMEM) {
			*fatal = last;
			break;
		}
		if (off != NULL)
			*off = (size_t)nals[i].end;
	}
	/* stopped early: the snapshot ran ahead of the context */
	rd->ps_gen_seen = i == n ? rd->ctx->ps_generation : ~(uint64_t)0;
	if (rd_trace())
		fprintf(stderr,
			"h264_reader: %zu NAL units, %zu slices: pre-pass %.2f ms, "
			"batch set-up %.2f ms, launch + replay %.2f ms (of which "
			"waiting for the device %.2f ms)\n",
			n, ns, t1 - t0, t2 - t1, rd_now() - t2, rd->trace_wait_ms);
out:
	free(plan);
	free(descs);
	return last;
}

int h264_reader_parse(struct h264_reader *rd, uint32_t flags, const uint8_t *buf,
		      size_t len, size_t *off)
{
	struct h264_amd_engine *eng;
	struct h264_amd_nal *nals = NULL;
	size_t count = 0, consumed = 0, cap;
	int res, fatal = 0;

	if (rd == NULL || buf == NULL || off == NULL)
		return -EINVAL;
	rd->stop = 0;
	rd->flags = flags;
	*off = 0;
	if (len == 0)
		return 0;
	res = h264_host_engine(&eng);
	if (res < 0)
		return res;

	h264_host_lock();
	double t0 = rd_trace() ? rd_now() : 0, t1 = 0;
	rd->trace_wait_ms = 0;
	res = h264_host_grow(&rd->d_buf, &rd->d_buf_cap, len + 64);
	if (res == 0)
		res = h264_amd_upload(eng, rd->d_buf, buf, len);
	if (rd_trace())
		t1 = rd_now();
	/* NAL table: sized by a guess, rescanned when the guess was short */
	cap = rd->cap_nals / sizeof(*nals);
	if (cap < 4096)
		cap = 4096;
	while (res == 0) {
		res = h264_host_grow(&rd->d_nals, &rd->cap_nals,
				     cap * sizeof(*nals));
		if (res == 0)
			res = h264_amd_scan(eng, rd->d_buf, len, rd->d_nals, cap,
					    &count, &consumed);
		if (res < 0 || count <= cap)
			break;
		cap = count;
	}
	if (res == 0 && count > 0) {
		nals = malloc(count * sizeof(*nals));
		res = nals == NULL ? -ENOMEM
				   : h264_amd_download(eng, nals, rd->d_nals,
						       count * sizeof(*nals));
	}
	if (rd_trace())
		fprintf(stderr, "h264_reader: %zu bytes: upload %.2f ms, scan + NAL "
				"table %.2f ms\n", len, t1 - t0, rd_now() - t1);
	if (res == 0) {
		run(rd, flags, buf, nals, count, 1, off, &fatal);
		res = fatal;
		/* everything replayed: the offset the reference's loop ends on
		 * (a trailing NAL without end marker runs to len) */
		if (res == 0 && !rd->stop)
			*off = consumed;
	}
	h264_host_unlock();
	free(nals);
	return res;
}

int h264_reader_parse_nalu(struct h264_reader *rd, uint32_t flags,
			   const uint8_t *buf, size_t len)
{
	struct h264_amd_engine *eng;
	struct h264_amd_nal nal;
	int res, fatal = 0;

	if (rd == NULL || buf == NULL)
		return -EINVAL;
	rd->stop = 0;
	rd->flags = flags;
	nal.start = 0;
	nal.end = len;
	if (!is_slice_nal(buf, len)) {
		/* control-plane NAL units never leave the host */
		res = run(rd, flags, buf, &nal, 1, 0, NULL, &fatal);
		return fatal < 0 ? fatal : res;
	}
	const int header_only =
		h264_host_slice_is_header_only(rd->ctx, flags, buf, len) &&
		getenv("H264_AMD_NO_HOST_HEADERS") == NULL;
	/* no engine, no slice path: not even the header-only one -- except for
	 * the sanitizer run of the host parser, which has no GPU to offer
	 * (tests/helpers/fuzz_ctl.c sets the variable; nothing else does) */
	if (!(header_only &&
	      getenv("H264_AMD_TEST_HEADERS_WITHOUT_ENGINE") != NULL)) {
		res = h264_host_engine(&eng);
		if (res < 0)
			return res;
	}
	if (header_only) {
		/* one slice whose data nobody looks at: a few dozen bytes of
		 * header are read here instead of crossing PCIe twice around a
		 * kernel launch (slicehdr.c) */
		struct rd_host_slice *hs = malloc(sizeof(*hs));
		if (hs == NULL)
			return -ENOMEM;
		res = h264_host_slice_header(rd->ctx, buf, len, &hs->r, &hs->sh);
		if (res == 0)
			res = replay_nal(rd, NULL, buf, len, -1, hs);
		free(hs);
		return res;
	}
	h264_host_lock();
	res = h264_host_grow(&rd->d_buf, &rd->d_buf_cap, len + 64);
	if (res == 0)
		res = h264_amd_upload(eng, rd->d_buf, buf, len);
	if (res == 0) {
		res = run(rd, flags, buf, &nal, 1, 0, NULL, &fatal);
		if (fatal < 0)
			res = fatal;
	}
	h264_host_unlock();
	return res;
}
