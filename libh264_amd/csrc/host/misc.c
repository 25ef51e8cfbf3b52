/**
 * misc.c -- SPS-derived variables, stream summary from raw parameter sets,
 * AVCC <-> Annex-B conversion and the context-free parse helpers (host).
 * Drop-in for src/h264.c:36-272 and src/h264_reader.c:165-255.
 */
#include "h264_priv.h"
#include <arpa/inet.h>

/* 6.2 / 7.4.2.1.1 (ref src/h264.c:36-117) */
int h264_get_sps_derived(const struct h264_sps *sps, struct h264_sps_derived *d)
{
	/* SubWidthC, SubHeightC, MbWidthC, MbHeightC by ChromaArrayType */
	static const uint8_t chroma[4][4] = {
		{0, 0, 0, 0}, {2, 2, 8, 8}, {2, 1, 8, 16}, {1, 1, 16, 16}};
	uint32_t cat, frame_factor;
	if (sps == NULL || d == NULL)
		return -EINVAL;
	memset(d, 0, sizeof(*d));
	cat = sps->separate_colour_plane_flag ? 0 : sps->chroma_format_idc;
	frame_factor = 2 - (sps->frame_mbs_only_flag ? 1 : 0);
	d->ChromaArrayType = cat;
	if (cat < 4) {
		d->SubWidthC = chroma[cat][0];
		d->SubHeightC = chroma[cat][1];
		d->MbWidthC = chroma[cat][2];
		d->MbHeightC = chroma[cat][3];
	}
	d->BitDepthLuma = sps->bit_depth_luma_minus8 + 8;
	d->BitDepthChroma = sps->bit_depth_chroma_minus8 + 8;
	d->QpBdOffsetLuma = 6 * sps->bit_depth_luma_minus8;
	d->QpBdOffsetChroma = 6 * sps->bit_depth_chroma_minus8;
	d->RawMbBits = 256 * d->BitDepthLuma +
		       2 * d->MbWidthC * d->MbHeightC * d->BitDepthChroma;
	/* out-of-range exponents: count mod 32, as the x86 code of the reference
	 * behaves, minus the undefined behaviour */
	d->MaxFrameNum = 1u << ((sps->log2_max_frame_num_minus4 + 4) & 31u);
	d->MaxPicOrderCntLsb =
		1u << ((sps->log2_max_pic_order_cnt_lsb_minus4 + 4) & 31u);
	d->PicWidthInMbs = sps->pic_width_in_mbs_minus1 + 1;
	d->PicHeightInMapUnits = sps->pic_height_in_map_units_minus1 + 1;
	d->PicWidthInSamplesLuma = 16 * d->PicWidthInMbs;
	d->PicWidthInSamplesChroma = d->MbWidthC * d->PicWidthInMbs;
	d->PicSizeInMapUnits = d->PicWidthInMbs * d->PicHeightInMapUnits;
	d->FrameHeightInMbs = frame_factor * d->PicHeightInMapUnits;
	d->CropUnitX = cat == 0 ? 1 : d->SubWidthC;
	d->CropUnitY = (cat == 0 ? 1 : d->SubHeightC) * frame_factor;
	d->Width = d->PicWidthInSamplesLuma -
		   d->CropUnitX * (sps->frame_crop_left_offset +
				   sps->frame_crop_right_offset);
	d->Height = 16 * d->FrameHeightInMbs -
		    d->CropUnitY * (sps->frame_crop_top_offset +
				    sps->frame_crop_bottom_offset);
	return 0;
}

/* ---- context-free parsing (ref src/h264_reader.c:165-255) ---------------- */

int h264_parse_nalu_header(const uint8_t *buf, size_t len,
			   struct h264_nalu_header *nh)
{
	struct h264_bitstream bs;
	struct h264_syn s = {&bs, 0, 0};
	if (buf == NULL || nh == NULL)
		return -EINVAL;
	memset(nh, 0, sizeof(*nh));
	h264_bs_cinit(&bs, buf, len, 1);
	return h264_syn_nalu_header(&s, nh);
}

int h264_parse_sps(const uint8_t *buf, size_t len, struct h264_sps *sps)
{
	struct h264_bitstream bs;
	struct h264_syn s = {&bs, 0, 0};
	struct h264_nalu_header nh;
	if (buf == NULL || sps == NULL)
		return -EINVAL;
	memset(sps, 0, sizeof(*sps));
	memset(&nh, 0, sizeof(nh));
	sps->chroma_format_idc = 1; /* inferred when absent, 7.4.2.1.1 */
	h264_bs_cinit(&bs, buf, len, 1);
	if (h264_syn_nalu_header(&s, &nh) < 0)
		return s.err;
	if (nh.nal_unit_type != H264_NALU_TYPE_SPS)
		return -EIO;
	return h264_syn_sps(&s, sps);
}

int h264_parse_pps(const uint8_t *buf, size_t len, const struct h264_sps *sps,
		   struct h264_pps *pps)
{
	struct h264_bitstream bs;
	struct h264_syn s = {&bs, 0, 0};
	struct h264_nalu_header nh;
	if (buf == NULL || sps == NULL || pps == NULL)
		return -EINVAL;
	memset(pps, 0, sizeof(*pps));
	memset(&nh, 0, sizeof(nh));
	h264_bs_cinit(&bs, buf, len, 1);
	if (h264_syn_nalu_header(&s, &nh) < 0)
		return s.err;
	if (nh.nal_unit_type != H264_NALU_TYPE_PPS)
		return -EIO;
	if (h264_syn_pps_ids(&s, pps) < 0)
		return s.err;
	if (sps->seq_parameter_set_id != pps->seq_parameter_set_id)
		return -EIO;
	return h264_syn_pps_body(&s, sps, pps);
}

/* ref src/h264.c:120-181 */
int h264_get_info(const uint8_t *sps, size_t sps_len, const uint8_t *pps,
		  size_t pps_len, struct h264_info *info)
{
	struct h264_sps *s;
	struct h264_pps *p;
	struct h264_sps_derived d;
	int res;
	if (sps == NULL || sps_len == 0 || pps == NULL || pps_len == 0 ||
	    info == NULL)
		return -EINVAL;
	memset(info, 0, sizeof(*info));
	s = calloc(1, sizeof(*s));
	p = calloc(1, sizeof(*p));
	res = (s == NULL || p == NULL) ? -ENOMEM : 0;
	if (res == 0)
		res = h264_parse_sps(sps, sps_len, s);
	if (res == 0)
		res = h264_parse_pps(pps, pps_len, s, p);
	if (res == 0)
		res = h264_get_sps_derived(s, &d);
	if (res == 0)
		res = h264_get_info_from_ps(s, &d, info);
	free(s);
	free(p);
	return res;
}

/* ---- AVCC <-> Annex-B, in place, 4-byte start codes (ref :184-272) ------- */

/* offset of the next 00 00 00 01 at or after `from`, or len */
static size_t next_start_code4(const uint8_t *b, size_t len, size_t from)
{
	size_t i;
	for (i = from; i + 4 <= len; i++) {
		if (b[i] == 0 && b[i + 1] == 0 && b[i + 2] == 0 && b[i + 3] == 1)
			return i;
	}
	return len;
}

int h264_byte_stream_to_avcc(uint8_t *data, size_t len)
{
	size_t pos, next, nalu_len;
	uint32_t be;
	if (data == NULL || len == 0)
		return -EINVAL;
	pos = next_start_code4(data, len, 0);
	if (pos == len)
		return 0; /* no start code: nothing to convert */
	while (len - pos > 4) {
		next = next_start_code4(data, len, pos + 4);
		nalu_len = next - (pos + 4);
		be = htonl((uint32_t)nalu_len);
		memcpy(data + pos, &be, 4);
		pos = next;
	}
	return 0;
}

int h264_avcc_to_byte_stream(uint8_t *data, size_t len)
{
	static const uint8_t start_code[4] = {0, 0, 0, 1};
	size_t pos = 0;
	uint32_t be, nalu_len;
	if (data == NULL || len == 0)
		return -EINVAL;
	while (pos < len) {
		/* deviation: the reference reads the four length bytes even when
		 * fewer are left (src/h264.c:259-260) */
		if (len - pos < 4)
			return -EPROTO;
		memcpy(&be, data + pos, 4);
		nalu_len = ntohl(be);
		if (nalu_len == 0)
			return -EPROTO;
		memcpy(data + pos, start_code, 4);
		pos += 4 + (size_t)nalu_len;
	}
	return 0;
}
