/**
 * libh264_amd -- MI355X-native H.264 bitstream parse/serialise engine behind
 * the libh264 C API.  Umbrella header, drop-in for the reference's
 * include/h264/h264.h (include order ref :50-58).
 */
#ifndef _H264_H_
#define _H264_H_

#include <errno.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifdef H264_API_EXPORTS
#	define H264_API __attribute__((visibility("default")))
#else
#	define H264_API
#endif

#include "h264/h264_types.h"
#include "h264/h264_bitstream.h"
#include "h264/h264_ctx.h"
#include "h264/h264_dump.h"
#include "h264/h264_reader.h"
#include "h264/h264_writer.h"

/* Variables of 7.4.2.1.1 computed from an SPS (ref src/h264.c:36-117) */
H264_API int h264_get_sps_derived(const struct h264_sps *sps,
				  struct h264_sps_derived *sps_derived);

/* Parse one SPS NAL + one PPS NAL (no start codes) into a stream summary
 * (ref src/h264.c:120-181) */
H264_API int h264_get_info(const uint8_t *sps,
			   size_t sps_len,
			   const uint8_t *pps,
			   size_t pps_len,
			   struct h264_info *info);

H264_API int h264_sar_to_aspect_ratio_idc(unsigned int sar_width,
					  unsigned int sar_height);

/* In-place container conversion; only 4-byte start codes are recognised
 * (ref src/h264.c:184-272) */
H264_API int h264_byte_stream_to_avcc(uint8_t *data, size_t len);
H264_API int h264_avcc_to_byte_stream(uint8_t *data, size_t len);

#ifdef __cplusplus
}
#endif

#endif /* !_H264_H_ */
