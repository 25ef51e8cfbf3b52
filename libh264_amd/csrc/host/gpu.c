/**
 * gpu.c -- engine glue of the drop-in host library.
 *
 * One process-wide engine on HIP device $H264_AMD_DEVICE (default 0, or the
 * rank's LOCAL_RANK when that is set and H264_AMD_DEVICE is not), created on
 * first use.  Without a usable MI355X every slice entry point of libh264.so
 * fails with -ENODEV: there is no CPU slice path to fall back to.
 */
#include "h264_priv.h"

#include <pthread.h>

static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static struct h264_amd_engine *g_engine;
static struct h264_amd_group *g_group; /* H264_AMD_DEVICES=0,1,...: several */
static int g_engine_err = -ENODEV;

static void engine_init(void)
{
	const char *dev = getenv("H264_AMD_DEVICE");
	const char *devs = getenv("H264_AMD_DEVICES");
	int ordinal = 0;
	if (devs != NULL && strchr(devs, ',') != NULL && h264_amd_available()) {
		/* a list of HIP ordinals: an engine group; the first one holds
		 * the reader's and writer's buffers */
		int list[64];
		uint32_t n = 0;
		const char *p = devs;
		while (*p != '\0' && n < 64) {
			char *end = NULL;
			long v = strtol(p, &end, 10);
			if (end == p)
				break;
			list[n++] = (int)v;
			p = *end == ',' ? end + 1 : end;
		}
		if (n > 1) {
			g_engine_err = h264_amd_group_new(list, n, &g_group);
			if (g_engine_err == 0) {
				g_engine = h264_amd_group_engine(g_group, 0);
				return;
			}
			g_group = NULL;
			return;
		}
	}
	if (dev == NULL || *dev == '\0')
		dev = getenv("LOCAL_RANK");
	if (dev != NULL && *dev != '\0')
		ordinal = atoi(dev);
	if (!h264_amd_available()) {
		g_engine_err = -ENODEV;
		return;
	}
	g_engine_err = h264_amd_engine_new(ordinal, NULL, &g_engine);
	if (g_engine_err < 0)
		g_engine = NULL;
}

int h264_host_engine(struct h264_amd_engine **ret)
{
	pthread_once(&g_once, engine_init);
	*ret = g_engine;
	return g_engine != NULL ? 0 : (g_engine_err < 0 ? g_engine_err : -ENODEV);
}

/* the engine group of H264_AMD_DEVICES, or NULL: one device */
struct h264_amd_group *h264_host_group(void)
{
	pthread_once(&g_once, engine_init);
	return g_group;
}

/* the engine's stream, events and timing slots are shared by every reader and
 * writer call of the process: one GPU sequence at a time.  Recursive, because
 * a reader callback may call the writer (the reference's dump tool re-writes
 * every NAL unit from nalu_end). */
static pthread_mutex_t g_lock = PTHREAD_RECURSIVE_MUTEX_INITIALIZER_NP;

void h264_host_lock(void)
{
	pthread_mutex_lock(&g_lock);
}

void h264_host_unlock(void)
{
	pthread_mutex_unlock(&g_lock);
}

/* device scratch that only ever grows; contents are NOT preserved */
int h264_host_grow(void **d_ptr, size_t *cap, size_t want)
{
	struct h264_amd_engine *eng;
	size_t ncap;
	int res = h264_host_engine(&eng);
	if (res < 0)
		return res;
	if (want <= *cap && *d_ptr != NULL)
		return 0;
	ncap = *cap + *cap / 2;
	if (ncap < want)
		ncap = want;
	ncap = (ncap + 4095) & ~(size_t)4095;
	if (*d_ptr != NULL)
		h264_amd_free(eng, *d_ptr);
	*d_ptr = NULL;
	*cap = 0;
	res = h264_amd_malloc(eng, ncap, d_ptr);
	if (res < 0)
		return res;
	*cap = ncap;
	return 0;
}

/* Mirror every SPS/PPS of ctx's tables into `p` as a fresh single epoch.
 * SPS first: a PPS names its SPS by id. */
int h264_host_params_from_ctx(struct h264_amd_params *p, const struct h264_ctx *ctx)
{
	size_t i;
	int res = h264_amd_params_reset(p);
	for (i = 0; res >= 0 && i < H264_ARRAY_SIZE(ctx->sps_table); i++) {
		if (ctx->sps_table[i] != NULL)
			res = h264_amd_params_put_sps(p, ctx->sps_table[i]);
	}
	for (i = 0; res >= 0 && i < H264_ARRAY_SIZE(ctx->pps_table); i++) {
		if (ctx->pps_table[i] != NULL)
			res = h264_amd_params_put_pps(p, ctx->pps_table[i]);
	}
	return res < 0 ? res : 0;
}

int h264_host_ctx_params(struct h264_ctx *ctx, struct h264_amd_params **ret,
			 uint32_t *epoch)
{
	int res;
	if (ctx->wr_params == NULL) {
		res = h264_amd_params_new(&ctx->wr_params);
		if (res < 0)
			return res;
		ctx->wr_params_generation = ~(uint64_t)0;
	}
	if (ctx->wr_params_generation != ctx->ps_generation) {
		res = h264_host_params_from_ctx(ctx->wr_params, ctx);
		if (res < 0)
			return res;
		ctx->wr_params_generation = ctx->ps_generation;
	}
	res = h264_amd_params_epoch(ctx->wr_params);
	if (res < 0)
		return res;
	*ret = ctx->wr_params;
	*epoch = (uint32_t)res;
	return 0;
}

/* Identity of the sources this library was built from (libh264_amd/build.py),
 * the drop-in library's counterpart of h264_amd_build_id(). */
#ifndef H264_AMD_BUILD_ID
#define H264_AMD_BUILD_ID "unidentified"
#endif
H264_API const char *h264_amd_dropin_build_id(void)
{
	static const char id[] = "h264_amd_build_id=" H264_AMD_BUILD_ID;
	return id + sizeof("h264_amd_build_id=") - 1;
}
