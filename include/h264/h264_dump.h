/**
 * h264_dump.h -- per-NAL JSON dump, drop-in for the reference's
 * include/h264/h264_dump.h (:31-73): same types, same six entry points.
 *
 * The reference builds the dump with json-c and hands the object tree out
 * through h264_dump_get_json_object(); struct json_object is only forward
 * declared there, as it is here.  This build carries its own small JSON tree
 * (libh264_amd/csrc/host/dump.c) with json-c's object semantics and its default
 * text form: h264_dump_get_json_str() is the way out, and
 * h264_dump_get_json_object() answers -ENOSYS (there is no json-c object to
 * give).  Slice data (H264_DUMP_FLAGS_SLICE_DATA, CAVLC only like the
 * reference) comes from the slice-decode kernel's full-syntax records.
 */
#ifndef _H264_DUMP_H_
#define _H264_DUMP_H_

struct json_object;
struct h264_dump;

/* Dump slice data (CAVLC only) */
#define H264_DUMP_FLAGS_SLICE_DATA 0x01

enum h264_dump_type {
	H264_DUMP_TYPE_JSON,
};

struct h264_dump_cfg {
	enum h264_dump_type type;
};

H264_API
int h264_dump_new(const struct h264_dump_cfg *cfg, struct h264_dump **ret_obj);

H264_API
int h264_dump_destroy(struct h264_dump *dump);

H264_API
int h264_dump_clear(struct h264_dump *dump);

/* -ENOSYS in this build (no json-c), see above */
H264_API
int h264_dump_get_json_object(struct h264_dump *dump,
			      struct json_object **jobj);

/* borrowed, valid until the next h264_dump_nalu / clear / destroy */
H264_API
int h264_dump_get_json_str(struct h264_dump *dump, const char **str);

/* describe the NAL unit the context currently holds */
H264_API
int h264_dump_nalu(struct h264_dump *dump,
		   struct h264_ctx *ctx,
		   uint32_t flags);

#endif /* !_H264_DUMP_H_ */
