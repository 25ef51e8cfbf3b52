/**
 * types.c -- enum-to-string helpers and macroblock class predicates (host).
 * Drop-in for src/h264_types.c:36-308: same strings, same fallbacks.
 */
#include "h264_priv.h"
#include <stdio.h>

struct name_entry {
	int value;
	const char *name;
};

static const char *lookup(const struct name_entry *t, size_t n, int v,
			  const char *fallback)
{
	size_t i;
	for (i = 0; i < n; i++)
		if (t[i].value == v)
			return t[i].name;
	return fallback;
}

#define N(t) (sizeof(t) / sizeof((t)[0]))

const char *h264_nalu_type_str(enum h264_nalu_type val)
{
	static const struct name_entry t[] = {
		{1, "SLICE"},	  {2, "SLICE_DPA"},	{3, "SLICE_DPB"},
		{4, "SLICE_DPC"}, {5, "SLICE_IDR"},	{6, "SEI"},
		{7, "SPS"},	  {8, "PPS"},		{9, "AUD"},
		{10, "END_OF_SEQ"}, {11, "END_OF_STREAM"}, {12, "FILLER"}};
	return lookup(t, N(t), val, "UNKNOWN");
}

const char *h264_slice_type_str(enum h264_slice_type val)
{
	static const struct name_entry t[] = {
		{0, "P"}, {1, "B"}, {2, "I"}, {3, "SP"}, {4, "SI"}};
	return lookup(t, N(t), val, "UNKNOWN");
}

const char *h264_mb_type_str(enum h264_mb_type val)
{
	static const char *const names[] = {
		"UNKNOWN",  "I_NxN",  "I_16x16", "I_PCM",   "SI",
		"P_16x16",  "P_16x8", "P_8x16",	 "P_8x8",   "P_8x8ref0",
		"P_SKIP",   "B_Direct_16x16",	 "B_16x16", "B_16x8",
		"B_8x16",   "B_8x8",  "B_SKIP"};
	if ((unsigned)val < N(names))
		return names[val];
	return "UNKNOWN";
}

int h264_mb_type_is_intra(enum h264_mb_type val)
{
	return val >= H264_MB_TYPE_I_NxN && val <= H264_MB_TYPE_SI;
}

int h264_mb_type_is_inter(enum h264_mb_type val)
{
	return val >= H264_MB_TYPE_P_16x16 && val <= H264_MB_TYPE_B_SKIP;
}

const char *h264_profile_str(enum h264_profile val)
{
	static const struct name_entry t[] = {
		{44, "CAVLC_444"}, {66, "BASELINE"},  {77, "MAIN"},
		{88, "EXTENDED"},  {100, "HIGH"},     {110, "HIGH_10"},
		{122, "HIGH_422"}, {244, "HIGH_444"}};
	return lookup(t, N(t), val, "UNKNOWN");
}

const char *h264_color_format_str(enum h264_color_format val)
{
	static const struct name_entry t[] = {
		{0, "MONO"}, {1, "YUV420"}, {2, "YUV422"}, {3, "YUV444"}};
	return lookup(t, N(t), val, "UNKNOWN");
}

char *h264_aspect_ratio_str_alloc(enum h264_aspect_ratio val, uint32_t sar_width,
				  uint32_t sar_height)
{
	static const char *const ratios[17] = {
		"UNSPECIFIED", "1:1",	"12:11", "10:11", "16:11", "40:33",
		"24:11",       "20:11", "32:11", "80:33", "18:11", "15:11",
		"64:33",       "160:99", "4:3",	 "3:2",	  "2:1"};
	char *str = NULL;
	if ((unsigned)val < 17)
		return strdup(ratios[val]);
	if (val == H264_ASPECT_RATIO_EXTENDED_SAR) {
		if (asprintf(&str, "EXTENDED_SAR_%u:%u", sar_width, sar_height) < 0)
			return NULL;
		return str;
	}
	return strdup("UNKNOWN");
}

const char *h264_sei_type_str(enum h264_sei_type val)
{
	/* enumerators 0..54 in the order of enum h264_sei_type */
	static const char *const names[] = {
		"BUFFERING_PERIOD",
		"PIC_TIMING",
		"PAN_SCAN_RECT",
		"FILLER_PAYLOAD",
		"USER_DATA_REGISTERED",
		"USER_DATA_UNREGISTERED",
		"RECOVERY_POINT",
		"DEC_REF_PIC_MARKING_REPETITION",
		"SPARE_PIC",
		"SCENE_INFO",
		"SUB_SEQ_INFO",
		"SUB_SEQ_LAYER_CHARACTERISTICS",
		"SUB_SEQ_CHARACTERISTICS",
		"FULL_FRAME_FREEZE",
		"FULL_FRAME_FREEZE_RELEASE",
		"FULL_FRAME_SNAPSHOT",
		"PROGRESSIVE_REFINEMENT_SEGMENT_START",
		"PROGRESSIVE_REFINEMENT_SEGMENT_END",
		"MOTION_CONSTRAINED_SLICE_GROUP_SET",
		"FILM_GRAIN_CHARACTERISTICS",
		"DEBLOCKING_FILTER_DISPLAY_PREFERENCE",
		"STEREO_VIDEO_INFO",
		"POST_FILTER_HINT",
		"TONE_MAPPING_INFO",
		"SCALABILITY_INFO",
		"SUB_PIC_SCALABLE_LAYER",
		"NON_REQUIRED_LAYER_REP",
		"PRIORITY_LAYER_INFO",
		"LAYERS_NOT_PRESENT",
		"LAYER_DEPENDENCY_CHANGE",
		"SCALABLE_NESTING",
		"BASE_LAYER_TEMPORAL_HRD",
		"QUALITY_LAYER_INTEGRITY_CHECK",
		"REDUNDANT_PIC_PROPERTY",
		"TL0_DEP_REP_INDEX",
		"TL_SWITCHING_POINT",
		"PARALLEL_DECODING_INFO",
		"MVC_SCALABLE_NESTING",
		"VIEW_SCALABILITY_INFO",
		"MULTIVIEW_SCENE_INFO",
		"MULTIVIEW_ACQUISITION_INFO",
		"NON_REQUIRED_VIEW_COMPONENT",
		"VIEW_DEPENDENCY_CHANGE",
		"OPERATION_POINTS_NOT_PRESENT",
		"BASE_VIEW_TEMPORAL_HRD",
		"FRAME_PACKING_ARRANGEMENT",
		"MULTIVIEW_VIEW_POSITION",
		"DISPLAY_ORIENTATION",
		"MVCD_SCALABLE_NESTING",
		"MVCD_VIEW_SCALABILITY_INFO",
		"DEPTH_REPRESENTATION_INFO",
		"THREE_DIMENSIONAL_REFERENCE_DISPLAYS_INFO",
		"DEPTH_TIMING",
		"DEPTH_SAMPLING_INFO",
		"CONSTRAINED_DEPTH_PARAMETER_SET_IDENTIFIER"};
	if ((unsigned)val < N(names))
		return names[val];
	return "UNKNONW"; /* sic: the reference's spelling */
}
