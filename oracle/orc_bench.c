/**
 * oracle/orc_bench.c -- TEST INFRASTRUCTURE (see orc.h).
 * CPU baseline for bench.py ("cpu_baseline", kind "port"): the restatement's
 * reader run the way an application drives the reference --
 * h264_reader_parse_nalu(flags = SLICE_DATA) per NAL with a slice_data_end
 * callback -- over a bounded sample of the same slice batch the GPU decodes,
 * one reader per thread (readers share no state, src/h264_reader.c:30-36).
 */
#define _GNU_SOURCE
#include "orc.h"
#include <pthread.h>
#include <time.h>

struct bench_job {
	const uint8_t *buf;
	const uint64_t *offs;
	const uint32_t *sizes;
	uint32_t n, first, step;
	const uint8_t *sps, *pps;
	size_t sps_len, pps_len;
	uint32_t flags;
	uint64_t mbs, bytes;
	int errors;
};

static void count_mbs(struct orc_ctx *ctx, const struct h264_slice_header *sh,
		      uint32_t mb_count, void *ud)
{
	struct bench_job *j = ud;
	(void)ctx;
	(void)sh;
	j->mbs += mb_count;
}

static void *bench_worker(void *arg)
{
	struct bench_job *j = arg;
	struct orc_cbs cbs;
	struct orc_reader *rd = NULL;
	uint32_t i;
	memset(&cbs, 0, sizeof(cbs));
	cbs.slice_data_end = count_mbs;
	if (orc_reader_new(&cbs, j, &rd) < 0) {
		j->errors++;
		return NULL;
	}
	if (orc_reader_parse_nalu(rd, 0, j->sps, j->sps_len) < 0 ||
	    orc_reader_parse_nalu(rd, 0, j->pps, j->pps_len) < 0)
		j->errors++;
	for (i = j->first; i < j->n; i += j->step) {
		if (orc_reader_parse_nalu(rd, j->flags, j->buf + j->offs[i],
					  j->sizes[i]) < 0)
			j->errors++;
		j->bytes += j->sizes[i];
	}
	orc_reader_destroy(rd);
	return NULL;
}

int orc_bench_slices(const uint8_t *buf, const uint64_t *offs,
		     const uint32_t *sizes, uint32_t n, const uint8_t *sps,
		     size_t sps_len, const uint8_t *pps, size_t pps_len,
		     uint32_t flags, int threads, double *seconds,
		     uint64_t *mbs, uint64_t *bytes)
{
	struct bench_job *jobs;
	pthread_t *tids;
	struct timespec t0, t1;
	int t, errors = 0;
	if (threads < 1)
		threads = 1;
	jobs = calloc((size_t)threads, sizeof(*jobs));
	tids = calloc((size_t)threads, sizeof(*tids));
	if (jobs == NULL || tids == NULL)
		return -ENOMEM;
	for (t = 0; t < threads; t++) {
		jobs[t].buf = buf;
		jobs[t].offs = offs;
		jobs[t].sizes = sizes;
		jobs[t].n = n;
		jobs[t].first = (uint32_t)t;
		jobs[t].step = (uint32_t)threads;
		jobs[t].sps = sps;
		jobs[t].sps_len = sps_len;
		jobs[t].pps = pps;
		jobs[t].pps_len = pps_len;
		jobs[t].flags = flags;
	}
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (t = 0; t < threads; t++)
		pthread_create(&tids[t], NULL, bench_worker, &jobs[t]);
	*mbs = 0;
	*bytes = 0;
	for (t = 0; t < threads; t++) {
		pthread_join(tids[t], NULL);
		*mbs += jobs[t].mbs;
		*bytes += jobs[t].bytes;
		errors += jobs[t].errors;
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	*seconds = (double)(t1.tv_sec - t0.tv_sec) +
		   1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
	free(jobs);
	free(tids);
	return errors ? -EIO : 0;
}
