/**
 * cabac_decode.hip -- CABAC slice data on gfx950, as far as the CABAC SUBSET
 * goes (H264_AMD_READER_FLAGS_CABAC_SUBSET, include/h264_amd.h).
 *
 * NOT reference behaviour: the reference returns before any CABAC slice data
 * (src/h264_syntax_slice_data.h:715-717) and so does h264_amd_k_decode_slices.
 * This second kernel runs after it, on the slices it flagged
 * H264_AMD_SR_CABAC, and is the inverse -- by the standard's decoding process
 * 9.3.1.2 / 9.3.3.2.1-3 -- of the reference's ENCODER (src/h264_bac.c:253-358,
 * src/h264_cabac.c) for the syntax the reference's writers emit
 * (src/h264_writer.c:49-309):
 *   P / SP / B slices   mb_skip_flag = 1 macroblocks
 *   I slices            I_16x16 macroblocks without residual (mb_type 1..4,
 *                       intra_chroma_pred_mode, mb_qp_delta,
 *                       coded_block_flag(Intra16x16DCLevel) = 0)
 *   end_of_slice_flag
 * The first bin that leaves the subset ends the slice with
 * H264_AMD_E_UNSUPPORTED; macroblocks decoded until then are delivered.
 * CPU twin, statement for statement: oracle/orc_cabac.c
 * (orc_cabac_subset_slice_data).
 *
 * One slice per wavefront.  Per wave in LDS: the bit reader's ring, the T2
 * record being staged, all 1024 context variables (pStateIdx | valMPS << 8,
 * initialised 16 per lane from the (m, n) tables), one byte of neighbour
 * context per macroblock column.  Range table and state transitions sit in a
 * VGPR each, one state per lane, read with v_readlane.  The arithmetic decoder
 * state (codIRange, codIOffset) is wave-uniform.
 */
#ifndef H264_AMD_CABAC_DECODE_HIP
#define H264_AMD_CABAC_DECODE_HIP

/* engine.hip includes slice_decode.hip and slice_write.hip before this file:
 * OutCtx / sd_emit_* / WaveBits and BacTabs / sw_activate / cabac_tables */

#define CB_RING_OFF 0u
#define CB_STATE_OFF 1024u
#define CB_STAGE_OFF (1024u + WB_STATE_BYTES)
#define CB_STAGE_BYTES 2048u /* largest T2 record: 1792 B */
#define CB_CTX_OFF (CB_STAGE_OFF + CB_STAGE_BYTES)
#define CB_ROW_OFF (CB_CTX_OFF + 2048u)

/* neighbour context byte of one macroblock */
#define CB_NB_VALID 0x80u
#define CB_NB_SKIPPED 0x01u
#define CB_NB_CHROMA_NZ 0x02u /* intra_chroma_pred_mode != 0 */

struct cb_args {
	const uint8_t *buf;
	const struct h264_amd_slice_desc *descs;
	uint32_t n;
	const dev_sps *spsv;
	const dev_pps *ppsv;
	const dev_epoch *epochs;
	uint32_t n_sps, n_pps, n_epochs;
	const cabac_tables *tab;
	uint32_t tier, mb_capacity, mb_stride;
	struct h264_amd_slice_result *results;
	const struct h264_slice_header *headers;
	uint8_t *mb_out;
	uint32_t row_pitch; /* macroblock columns of neighbour context in LDS */
	/* waves per dispatch layer (4 per compute unit; one-wave workgroups):
	 * blockIdx.x / prio_div is a wave's age rank on its SIMD; 0: the launch
	 * fits one layer, no priority rotation (see slice_decode.hip) */
	uint32_t prio_div;
};

struct BacDec {
	uint32_t range, offset;
	uint32_t eof; /* a read went past the end of the NAL: -EIO */
};

/* n bits (1..9), failing exactly where the reference's byte reader would:
 * when fewer than n bits are left in the RBSP */
__device__ __forceinline__ uint32_t cb_bits(WaveBits &wb, BacDec &d, uint32_t lane,
					    uint32_t n)
{
	wb_sync(wb, lane);
	if (wb_all_in(wb) && wb_pos(wb) + n > wb_total_bits(wb)) {
		d.eof = 1;
		return 0;
	}
	return wb_u(wb, lane, n);
}

/* 9.3.3.2.2 RenormD, all shifts of one call at once */
__device__ __forceinline__ void cb_renorm(WaveBits &wb, BacDec &d, uint32_t lane)
{
	if (d.range < 256u) {
		uint32_t n = (uint32_t)__builtin_clz(d.range) - 23u;
		d.range <<= n;
		d.offset = (d.offset << n) | cb_bits(wb, d, lane, n);
	}
}

/* 9.3.3.2.1 DecodeDecision */
__device__ __forceinline__ uint32_t cb_decision(WaveBits &wb, BacDec &d,
						const BacTabs &t, uint16_t *ctx,
						uint32_t ci, uint32_t lane)
{
	uint32_t st = wb_uniform(ctx[ci]);
	uint32_t idx = st & 0xffu, mps = st >> 8;
	uint32_t q = (d.range >> 6) & 3u;
	uint32_t r4 = (uint32_t)__builtin_amdgcn_readlane((int)t.lps4, (int)idx);
	uint32_t r_lps = (r4 >> (8u * q)) & 0xffu;
	uint32_t tr = (uint32_t)__builtin_amdgcn_readlane((int)t.trans, (int)idx);
	uint32_t bin;
	d.range -= r_lps;
	if (d.offset >= d.range) {
		bin = mps ^ 1u;
		d.offset -= d.range;
		d.range = r_lps;
		if (idx == 0)
			mps ^= 1u;
		idx = tr & 0xffu;
	} else {
		bin = mps;
		idx = tr >> 8;
	}
	ctx[ci] = (uint16_t)(idx | (mps << 8)); /* every lane, same value */
	cb_renorm(wb, d, lane);
	return bin;
}

/* 9.3.3.2.3 DecodeTerminate */
__device__ __forceinline__ uint32_t cb_terminate(WaveBits &wb, BacDec &d,
						 uint32_t lane)
{
	d.range -= 2u;
	if (d.offset >= d.range)
		return 1;
	cb_renorm(wb, d, lane);
	return 0;
}

__device__ __forceinline__ uint32_t cb_hdr(const struct h264_slice_header *h,
					   uint32_t off)
{
	return wb_uniform(*reinterpret_cast<const uint32_t *>(
		reinterpret_cast<const uint8_t *>(h) + off));
}

/* the slice data of one CABAC slice; returns the status */
__device__ __forceinline__ int
cb_slice_data(WaveBits &wb, const BacTabs &bt, uint16_t *ctx, uint8_t *stage,
	      uint8_t *row, OutCtx &o, uint32_t lane, uint32_t type, uint32_t cat,
	      uint32_t W, uint32_t first_mb, uint32_t field_pic,
	      uint32_t mb_capacity, uint32_t &mb_count, uint32_t prio_slot,
	      uint32_t prio_mask)
{
	BacDec d;
	d.eof = 0;
	/* cabac_alignment_one_bit */
	{
		uint32_t p = wb_pos(wb) & 7u;
		if (p) {
			uint32_t v = cb_bits(wb, d, lane, 8u - p);
			if (d.eof || v != (1u << (8u - p)) - 1u)
				return -EIO;
		}
	}
	d.range = 510;
	d.offset = cb_bits(wb, d, lane, 9);
	if (d.eof)
		return -EIO;

	const bool is_i = type == H264_SLICE_TYPE_I, is_si = type == H264_SLICE_TYPE_SI;
	const uint32_t skip_off = type == H264_SLICE_TYPE_B ? 24u : 11u;
	/* field pictures: every macroblock carries the picture's flag */
	const uint32_t field_flag = field_pic ? (uint32_t)H264_AMD_MB_FIELD : 0u;
	const uint32_t skip_word0 =
		(type == H264_SLICE_TYPE_B ? (uint32_t)H264_MB_TYPE_B_SKIP
					   : (uint32_t)H264_MB_TYPE_P_SKIP) |
		(((uint32_t)H264_AMD_MB_SKIPPED | field_flag) << 16);
	uint32_t mbx = first_mb % W, mbidx = 0;
	uint32_t left = 0; /* neighbour byte of the previous macroblock */
	uint32_t prev_qp_nonzero = 0;

	for (;;) {
		if (mbidx >= mb_capacity)
			return H264_AMD_E_MB_OVERFLOW;
		/* fair shares of the SIMD (the arbiter serves the oldest wave
		 * first, see the slice-data loop of slice_decode.hip): a step
		 * round the four priorities every 256 macroblocks */
		if ((mbidx & prio_mask) == 0u) {
			switch (((mbidx >> 8) + prio_slot) & 3u) {
			case 0:
				__builtin_amdgcn_s_setprio(0);
				break;
			case 1:
				__builtin_amdgcn_s_setprio(1);
				break;
			case 2:
				__builtin_amdgcn_s_setprio(2);
				break;
			default:
				__builtin_amdgcn_s_setprio(3);
				break;
			}
		}
		/* neighbours inside the slice (ref src/h264_macroblock.c:306-351) */
		const uint32_t nbA = (mbx != 0 && mbidx >= 1) ? left : 0u;
		const uint32_t nbB = mbidx >= W ? wb_uniform(row[mbx]) : 0u;
		const uint32_t availA = nbA >> 7, availB = nbB >> 7;
		uint32_t me = CB_NB_VALID;

		if (!is_i && !is_si) {
			/* ctxIdxInc: neighbour available and not skipped */
			uint32_t inc = (availA & ~nbA & 1u) + (availB & ~nbB & 1u);
			uint32_t skip = cb_decision(wb, d, bt, ctx, skip_off + inc, lane);
			if (d.eof)
				return -EIO;
			if (!skip)
				return H264_AMD_E_UNSUPPORTED; /* a coded P/B MB */
			sd_emit_skipped(o, skip_word0, mbidx, lane);
			me |= CB_NB_SKIPPED;
			prev_qp_nonzero = 0;
		} else {
			if (is_si)
				return H264_AMD_E_UNSUPPORTED; /* mb_type prefix */
			/* mb_type (Table 9-36): every coded neighbour is I_16x16 in
			 * the subset, so condTermFlag = available */
			uint32_t b = cb_decision(wb, d, bt, ctx, 3u + availA + availB, lane);
			if (d.eof)
				return -EIO;
			if (!b)
				return H264_AMD_E_UNSUPPORTED; /* I_NxN */
			b = cb_terminate(wb, d, lane);
			if (d.eof)
				return -EIO;
			if (b)
				return H264_AMD_E_UNSUPPORTED; /* I_PCM */
			b = cb_decision(wb, d, bt, ctx, 3u + 3u, lane);
			if (d.eof)
				return -EIO;
			if (b)
				return H264_AMD_E_UNSUPPORTED; /* luma AC coded */
			b = cb_decision(wb, d, bt, ctx, 3u + 4u, lane);
			if (d.eof)
				return -EIO;
			if (b)
				return H264_AMD_E_UNSUPPORTED; /* chroma coded */
			uint32_t pred = cb_decision(wb, d, bt, ctx, 3u + 6u, lane) << 1;
			if (d.eof)
				return -EIO;
			pred |= cb_decision(wb, d, bt, ctx, 3u + 7u, lane);
			if (d.eof)
				return -EIO;
			if (cat == 3)
				return H264_AMD_E_UNSUPPORTED; /* Cb/Cr coded as luma */
			uint32_t chroma = 0;
			if (cat != 0) {
				/* intra_chroma_pred_mode, TU with cMax 3 */
				uint32_t inc = ((nbA >> 1) & 1u) + ((nbB >> 1) & 1u);
				chroma = cb_decision(wb, d, bt, ctx, 64u + inc, lane);
				if (d.eof)
					return -EIO;
				if (chroma == 1) {
					chroma += cb_decision(wb, d, bt, ctx, 64u + 3u, lane);
					if (d.eof)
						return -EIO;
				}
				if (chroma == 2) {
					chroma += cb_decision(wb, d, bt, ctx, 64u + 3u, lane);
					if (d.eof)
						return -EIO;
				}
			}
			/* mb_qp_delta: unary, mapped (Table 9-3) */
			uint32_t k = 0;
			b = cb_decision(wb, d, bt, ctx, 60u + prev_qp_nonzero, lane);
			if (d.eof)
				return -EIO;
			if (b) {
				k = 1;
				b = cb_decision(wb, d, bt, ctx, 60u + 2u, lane);
				if (d.eof)
					return -EIO;
				while (b) {
					if (++k > 255)
						return -EIO;
					b = cb_decision(wb, d, bt, ctx, 60u + 3u, lane);
					if (d.eof)
						return -EIO;
				}
			}
			int32_t qpd = (k & 1u) ? (int32_t)((k + 1u) / 2u) : -(int32_t)(k / 2u);
			prev_qp_nonzero = qpd != 0 ? 1u : 0u;
			/* coded_block_flag(Intra16x16DCLevel): an unavailable
			 * neighbour of an intra macroblock counts as coded */
			b = cb_decision(wb, d, bt, ctx,
					85u + (1u - availA) + 2u * (1u - availB), lane);
			if (d.eof)
				return -EIO;
			if (b)
				return H264_AMD_E_UNSUPPORTED; /* residual */

			/* record: 8 bytes of summary, zeros behind them */
			uint32_t word0 = (uint32_t)H264_MB_TYPE_I_16x16 | (field_flag << 16) |
					 ((1u + pred) << 24);
			uint32_t word1 = ((uint32_t)qpd & 0xffffu) | (chroma << 16) |
					 (pred << 24);
			*reinterpret_cast<uint32_t *>(stage) = word0;
			*reinterpret_cast<uint32_t *>(stage + 4) = word1;
			sd_emit_staged(o, stage, mbidx, lane);
			me |= chroma ? CB_NB_CHROMA_NZ : 0u;
		}
		row[mbx] = (uint8_t)me;
		left = me;
		mbx++;
		if (mbx == W)
			mbx = 0;
		mbidx++;
		mb_count = mbidx;
		uint32_t end = cb_terminate(wb, d, lane); /* end_of_slice_flag */
		if (d.eof)
			return -EIO;
		if (end)
			break;
	}
	return 0;
}

extern "C" __global__ void __launch_bounds__(64)
h264_amd_k_decode_cabac_subset(cb_args a)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	uint8_t *ring = lds + CB_RING_OFF;
	uint8_t *stage = lds + CB_STAGE_OFF;
	uint16_t *ctx = reinterpret_cast<uint16_t *>(lds + CB_CTX_OFF);
	uint8_t *row = lds + CB_ROW_OFF;
	const uint32_t lane = wb_lane();
	const uint32_t slice = blockIdx.x;
	if (slice >= a.n)
		return;

	struct h264_amd_slice_result r;
	sd_load_uniform(&r, a.results + slice);
	/* only what the first kernel left as "CABAC, header parsed, data raw" */
	if (r.status != 0 || !(r.flags & H264_AMD_SR_HEADER_OK) ||
	    !(r.flags & H264_AMD_SR_CABAC) || (r.flags & H264_AMD_SR_DATA_PARSED))
		return;
	struct h264_amd_slice_desc d;
	sd_load_uniform(&d, a.descs + slice);
	const struct h264_slice_header *hdr = a.headers + slice;

	int status = 0;
	uint32_t mb_count = 0;
	OutCtx o;
	o.tier = a.tier;
	o.mb_stride = a.mb_stride;
	o.rec_bytes = sd_t2_stride(r.chroma_array_type);
	o.mb_out = a.mb_out ? a.mb_out + (size_t)slice * a.mb_capacity * a.mb_stride
			    : nullptr;
	if (o.mb_out == nullptr)
		o.tier = H264_AMD_TIER_NONE;
	o.t0acc = 0;
	o.lo = o.hi = 0;

	SwParams P;
	{
		uint32_t e = d.epoch < a.n_epochs ? d.epoch : 0u;
		status = sw_activate(P, cb_hdr(hdr, WOFF(pic_parameter_set_id)), a.spsv,
				     a.ppsv, a.epochs + e, a.n_sps, a.n_pps);
	}
	const uint32_t W = r.pic_width_in_mbs;
	const uint32_t field_pic = cb_hdr(hdr, WOFF(field_pic_flag));
	if (status == 0) {
		if ((P.sps.mb_adaptive_frame_field_flag && !field_pic) ||
		    P.pps.num_slice_groups_minus1 > 0 || W == 0 || W > a.row_pitch)
			status = H264_AMD_E_UNSUPPORTED;
	}

	WaveBits wb;
	wb_open(wb, a.buf + d.offset, d.size, ring,
		reinterpret_cast<uint32_t *>(lds + CB_STATE_OFF), lane);
	if (status == 0) {
		/* to the end of the slice header, where the first kernel stopped */
		uint32_t left = r.end_bit;
		while (left) {
			uint32_t k = left < 512u ? left : 512u;
			wb_skip(wb, k);
			wb_sync(wb, lane);
			left -= k;
		}
		/* staging area: zeros (only the first 8 bytes ever change) */
		for (uint32_t part = 0; part < CB_STAGE_BYTES / 1024u; part++)
			*reinterpret_cast<uint4 *>(stage + part * 1024u + 16u * lane) =
				make_uint4(0, 0, 0, 0);
		/* 9.3.1.1: all 1024 context variables */
		const uint32_t type = (uint32_t)r.slice_type;
		{
			const uint32_t idc = cb_hdr(hdr, WOFF(cabac_init_idc));
			const bool intra = type == H264_SLICE_TYPE_I ||
					   type == H264_SLICE_TYPE_SI;
			const bool valid = intra || idc <= 2;
			const uint32_t col = intra ? 0u : 1u + (valid ? idc : 0u);
			for (uint32_t c = lane; c < 1024; c += 64) {
				uint32_t s = cabac_state_init(r.slice_qp,
							      a.tab->mn[c][col][0],
							      a.tab->mn[c][col][1]);
				ctx[c] = (uint16_t)(valid ? s : 0u);
			}
		}
		BacTabs bt;
		{
			const uint8_t *q = &a.tab->range_lps[lane][0];
			bt.lps4 = (uint32_t)q[0] | ((uint32_t)q[1] << 8) |
				  ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
			bt.trans = (uint32_t)a.tab->next_lps[lane] |
				   ((uint32_t)a.tab->next_mps[lane] << 8);
		}
		status = cb_slice_data(wb, bt, ctx, stage, row, o, lane, type,
				       r.chroma_array_type, W, r.first_mb_in_slice,
				       field_pic, a.mb_capacity, mb_count,
				       a.prio_div ? (blockIdx.x / a.prio_div) & 3u : 0u,
				       a.prio_div ? 255u : 0xffffffffu);
	}

	/* T0: flush the partial last group of 64 words */
	if (o.tier == H264_AMD_TIER_T0 && (mb_count & 63u) != 0) {
		uint32_t base = mb_count & ~63u;
		uint32_t l = lane < (mb_count & 63u) ? lane : (mb_count & 63u) - 1u;
		uint32_t v = (uint32_t)__shfl((int)o.t0acc, (int)l);
		*reinterpret_cast<uint32_t *>(o.mb_out + (size_t)(base + l) * o.mb_stride) = v;
	}
	uint32_t lo = o.lo, hi = o.hi;
#pragma unroll
	for (int sft = 32; sft >= 1; sft >>= 1) {
		lo += (uint32_t)__shfl_xor((int)lo, sft);
		hi += (uint32_t)__shfl_xor((int)hi, sft);
	}
	r.status = status;
	if (status != H264_AMD_E_UNSUPPORTED)
		r.flags |= H264_AMD_SR_DATA_PARSED;
	r.mb_count = mb_count;
	r.end_bit = status == 0 ? wb_pos(wb) : 0u;
	r.checksum = ((uint64_t)hi << 32) | lo;
	{
		const uint32_t *src = reinterpret_cast<const uint32_t *>(&r);
		uint32_t *dst = reinterpret_cast<uint32_t *>(a.results + slice);
		uint32_t k = lane & 15u, v = src[0];
#pragma unroll
		for (uint32_t j = 1; j < 16; j++)
			v = k == j ? src[j] : v;
		dst[k] = v;
	}
}

#endif /* H264_AMD_CABAC_DECODE_HIP */
