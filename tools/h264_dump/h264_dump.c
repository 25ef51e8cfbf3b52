/**
 * h264-dump -- per-NAL JSON dump of an Annex-B file through the drop-in
 * libh264 API (libh264_amd/lib/libh264.so), BASELINE config 1.
 *
 * Same behaviour as the reference's tool (tools/h264_dump.c:195-263 the
 * per-NAL callback, :309-435 the command line): the file is mapped, parsed with
 * h264_reader_parse(); behind every NAL unit its JSON description
 * (h264_dump_nalu with H264_DUMP_FLAGS_SLICE_DATA) is printed on one line,
 * then the NAL unit is written back with h264_write_nalu() and compared with
 * the input -- "OK", or both byte strings in hex.  Options -o / --output and
 * --pretty as there.  Own code; the JSON text comes from this library's own
 * emitter (csrc/host/dump.c), --pretty re-indents it here.
 *
 * Not in the reference's tool:
 *   --reader-slice-data   the variant its source documents under "#if 0"
 *                         (tools/h264_dump.c:67-77): slice data walked by the
 *                         READER (a slice_data_mb callback per macroblock), no
 *                         slice data in the dump
 *   --bench N             N silent passes over the file, then one line of JSON
 *                         with the time per pass on stderr
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <h264/h264.h>

struct app {
	const uint8_t *data;
	size_t size;
	struct h264_reader *reader;
	struct h264_dump *dump;
	FILE *fout; /* NULL: silent pass */
	int pretty;
	uint32_t dump_flags;
	uint64_t nalus, ok, mismatches, mbs;
};

static void hex(FILE *f, const uint8_t *p, size_t n)
{
	size_t i;
	for (i = 0; i < n; i++)
		fprintf(f, "%02x%s", p[i], (i + 1) % 32 == 0 || i + 1 == n ? "\n" : " ");
}

/* json-c's "pretty" form of a string in its "spaced" form: two spaces per
 * level, one member per line */
static void print_pretty(FILE *f, const char *s)
{
	int depth = 0, in_str = 0, i;
	for (; *s; s++) {
		char c = *s;
		if (in_str) {
			fputc(c, f);
			if (c == '\\' && s[1])
				fputc(*++s, f);
			else if (c == '"')
				in_str = 0;
			continue;
		}
		switch (c) {
		case '"':
			in_str = 1;
			fputc(c, f);
			break;
		case '{':
		case '[':
			fputc(c, f);
			depth++;
			fputc('\n', f);
			for (i = 0; i < depth; i++)
				fputs("  ", f);
			break;
		case '}':
		case ']':
			depth--;
			fputc('\n', f);
			for (i = 0; i < depth; i++)
				fputs("  ", f);
			fputc(c, f);
			break;
		case ',':
			fputc(c, f);
			fputc('\n', f);
			for (i = 0; i < depth; i++)
				fputs("  ", f);
			break;
		case ' ':
			if (s[-1] == ':')
				fputc(c, f);
			break;
		default:
			fputc(c, f);
			break;
		}
	}
	fputc('\n', f);
}

static void nalu_end_cb(struct h264_ctx *ctx, enum h264_nalu_type type,
			const uint8_t *buf, size_t len,
			const struct h264_nalu_header *nh, void *userdata)
{
	struct app *app = userdata;
	struct h264_bitstream bs;
	const char *str = NULL;
	int res, ok = 1;
	(void)type;
	(void)nh;

	app->nalus++;
	res = h264_dump_nalu(app->dump, ctx, app->dump_flags);
	if (res < 0)
		fprintf(stderr, "h264_dump_nalu: %s\n", strerror(-res));
	res = h264_dump_get_json_str(app->dump, &str);
	if (res < 0 || str == NULL)
		fprintf(stderr, "h264_dump_get_json_str: %s\n", strerror(-res));
	else if (app->fout && app->pretty)
		print_pretty(app->fout, str);
	else if (app->fout)
		fprintf(app->fout, "%s\n", str);

	if (!h264_ctx_is_nalu_unknown(ctx)) {
		h264_bs_init(&bs, NULL, 0, 1);
		res = h264_write_nalu(&bs, ctx);
		if (res < 0) {
			fprintf(stderr, "h264_write_nalu: %s\n", strerror(-res));
			ok = 0;
		} else if (bs.off == len) {
			ok = memcmp(bs.data, buf, len) == 0;
		} else if (bs.off < len) {
			/* trailing_zero_8bits belong to the NAL unit as found */
			size_t i;
			ok = memcmp(bs.data, buf, bs.off) == 0;
			for (i = bs.off; ok && i < len; i++)
				ok = buf[i] == 0x00;
		} else {
			ok = 0;
		}
		if (!ok && app->fout) {
			fprintf(stderr, "write mismatch\n");
			hex(app->fout, buf, len);
			hex(app->fout, bs.data, bs.off);
		}
		h264_bs_clear(&bs);
	}
	if (ok) {
		app->ok++;
		if (app->fout) {
			fprintf(app->fout, "OK\n");
			fflush(app->fout);
		}
	} else {
		app->mismatches++;
	}
}

static void slice_data_mb_cb(struct h264_ctx *ctx, const struct h264_slice_header *sh,
			     uint32_t mb_addr, enum h264_mb_type mb_type,
			     void *userdata)
{
	struct app *app = userdata;
	(void)ctx;
	(void)sh;
	(void)mb_addr;
	(void)mb_type;
	app->mbs++;
}

static double now_ms(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
}

static void usage(const char *prog)
{
	printf("Usage: %s [options] <input file>\n"
	       "\n"
	       "Options:\n"
	       "-h | --help                        Print this message\n"
	       "-o | --output <file>               Output file\n"
	       "     --pretty                      Pretty output for JSON file\n"
	       "     --reader-slice-data           Slice data by the reader's "
	       "callbacks, not in the dump\n"
	       "     --bench <passes>              Time silent passes\n"
	       "\n",
	       prog);
}

int main(int argc, char *argv[])
{
	enum { ARG_PRETTY = 256, ARG_READER_SD, ARG_BENCH };
	static const struct option long_options[] = {
		{"help", no_argument, NULL, 'h'},
		{"output", required_argument, NULL, 'o'},
		{"pretty", no_argument, NULL, ARG_PRETTY},
		{"reader-slice-data", no_argument, NULL, ARG_READER_SD},
		{"bench", required_argument, NULL, ARG_BENCH},
		{0, 0, 0, 0},
	};
	struct h264_ctx_cbs cbs;
	struct h264_dump_cfg dump_cfg;
	struct app app;
	const char *outpath = NULL, *inpath;
	uint32_t reader_flags = 0;
	int c, idx, res = 0, fd = -1, bench = 0, pass;
	struct stat st;
	size_t off = 0;
	void *map = MAP_FAILED;

	memset(&app, 0, sizeof(app));
	app.dump_flags = H264_DUMP_FLAGS_SLICE_DATA;
	while ((c = getopt_long(argc, argv, "ho:", long_options, &idx)) != -1) {
		switch (c) {
		case 'h':
			usage(argv[0]);
			return EXIT_SUCCESS;
		case 'o':
			outpath = optarg;
			break;
		case ARG_PRETTY:
			app.pretty = 1;
			break;
		case ARG_READER_SD:
			reader_flags = H264_READER_FLAGS_SLICE_DATA;
			app.dump_flags = 0;
			break;
		case ARG_BENCH:
			bench = atoi(optarg);
			break;
		default:
			usage(argv[0]);
			return EXIT_FAILURE;
		}
	}
	if (argc - optind < 1) {
		usage(argv[0]);
		return EXIT_FAILURE;
	}
	inpath = argv[optind];

	fd = open(inpath, O_RDONLY);
	if (fd < 0 || fstat(fd, &st) < 0) {
		res = -errno;
		fprintf(stderr, "open('%s'): %s\n", inpath, strerror(-res));
		goto out;
	}
	app.size = (size_t)st.st_size;
	if (app.size > 0) {
		map = mmap(NULL, app.size, PROT_READ, MAP_PRIVATE, fd, 0);
		if (map == MAP_FAILED) {
			res = -errno;
			fprintf(stderr, "mmap('%s'): %s\n", inpath, strerror(-res));
			goto out;
		}
		app.data = map;
	}

	memset(&cbs, 0, sizeof(cbs));
	cbs.nalu_end = nalu_end_cb;
	if (reader_flags)
		cbs.slice_data_mb = slice_data_mb_cb;
	res = h264_reader_new(&cbs, &app, &app.reader);
	if (res < 0) {
		fprintf(stderr, "h264_reader_new: %s\n", strerror(-res));
		goto out;
	}
	memset(&dump_cfg, 0, sizeof(dump_cfg));
	dump_cfg.type = H264_DUMP_TYPE_JSON;
	res = h264_dump_new(&dump_cfg, &app.dump);
	if (res < 0) {
		fprintf(stderr, "h264_dump_new: %s\n", strerror(-res));
		goto out;
	}

	if (bench > 0) {
		double best = 0, sum = 0, t0, t1;
		/* one pass to warm the engine up, then the timed ones */
		for (pass = -1; pass < bench; pass++) {
			off = 0;
			h264_ctx_clear(h264_reader_get_ctx(app.reader));
			app.nalus = app.ok = app.mismatches = app.mbs = 0;
			t0 = now_ms();
			res = h264_reader_parse(app.reader, reader_flags, app.data,
						app.size, &off);
			t1 = now_ms();
			if (res < 0)
				break;
			if (pass >= 0) {
				sum += t1 - t0;
				if (pass == 0 || t1 - t0 < best)
					best = t1 - t0;
			}
		}
		fprintf(stderr,
			"{\"passes\": %d, \"ms_per_pass\": %.4f, \"best_ms\": %.4f, "
			"\"bytes\": %zu, \"nal_units\": %llu, \"ok\": %llu, "
			"\"mismatches\": %llu, \"slice_data_mb_callbacks\": %llu, "
			"\"mode\": \"%s\"}\n",
			bench, sum / bench, best, app.size,
			(unsigned long long)app.nalus, (unsigned long long)app.ok,
			(unsigned long long)app.mismatches,
			(unsigned long long)app.mbs,
			reader_flags ? "reader-slice-data" : "dump-slice-data");
		goto out;
	}

	if (outpath != NULL) {
		app.fout = fopen(outpath, "w");
		if (app.fout == NULL) {
			res = -errno;
			fprintf(stderr, "fopen('%s'): %s\n", outpath, strerror(-res));
			goto out;
		}
	} else {
		app.fout = stdout;
	}
	res = h264_reader_parse(app.reader, reader_flags, app.data, app.size, &off);
	if (res < 0)
		fprintf(stderr, "h264_reader_parse: %s\n", strerror(-res));
	if (res >= 0 && app.mismatches)
		res = -EPROTO;

out:
	if (app.reader != NULL)
		h264_reader_destroy(app.reader);
	if (app.dump != NULL)
		h264_dump_destroy(app.dump);
	if (app.fout != NULL && app.fout != stdout)
		fclose(app.fout);
	if (map != MAP_FAILED)
		munmap(map, app.size);
	if (fd >= 0)
		close(fd);
	return res >= 0 ? EXIT_SUCCESS : EXIT_FAILURE;
}
