/**
 * cabac_slice.hip -- CABAC slice data on gfx950 (H264_AMD_READER_FLAGS_CABAC,
 * include/h264_amd.h): the whole macroblock layer of I / SI / P / SP / B slices,
 * frame and field pictures, all chroma formats, 8x8 transform, I_PCM.
 *
 * NOT reference behaviour: the reference returns before any CABAC slice data
 * (src/h264_syntax_slice_data.h:715-717) and so does h264_amd_k_decode_slices.
 * This second kernel runs after it, on the slices it flagged
 * H264_AMD_SR_CABAC.  Written from the standard (7.3.4 / 7.3.5 syntax, 9.3
 * binarisations, context index rules, decoding engine); spec-only, "parity
 * unpinned", except that it is the exact inverse of the reference's ENCODER on
 * what that encoder emits (src/h264_bac.c:253-358, src/h264_cabac.c:61-975,
 * src/h264_writer.c:79-129,177-219) and that the context variables are
 * initialised the reference's way (src/h264_bac.c:216-230,
 * src/h264_cabac_ctx_tables.c:4441-4468).  CPU twin with the same conventions
 * and the same behaviour on damaged input: oracle/orc_cabac_dec.c.
 *
 * One slice per wavefront, four per workgroup.  How the 64 lanes are used:
 *   * the (m, n) tables of all 1024 contexts sit in LDS, shared by the
 *     workgroup ("h264_cabac_ctx_tables resident in LDS"); the context
 *     VARIABLES live in 14 vector registers, one context per lane, grouped by
 *     syntax element so that a call site names its register statically: a
 *     state is read with one v_readlane and written with one v_writelane
 *     instead of an LDS round trip (54+ cycles on the bin-serial chain);
 *   * rangeTabLPS, the state transitions and Table 9-43 are vector registers
 *     too, one table row per lane;
 *   * the arithmetic decoder (codIRange, codIOffset scaled by up to 23 bits of
 *     read-ahead, so that renormalisation is a subtraction from a counter and
 *     the stream is touched once per 16 bits) is wave-uniform;
 *   * what ctxIdxInc asks of the neighbouring blocks -- coded_block_flag,
 *     |mvd|, ref_idx > 0 -- is kept as 5x5 (3x5, 3x3) GRIDS across the lanes of
 *     a register: current macroblock in the middle, the left macroblock's right
 *     column and the upper macroblock's bottom row (from the LDS row context)
 *     as the border, "not available" already folded into the border values.
 *     A ctxIdxInc is two v_readlane, without a case for "block on the edge";
 *   * the 64 dwords of a T2 record's head are one register (v_writelane per
 *     field), the coefficients of the block being decoded another (lane =
 *     scanning position), stored with one ds_write per lane;
 *   * with ChromaArrayType 3 the residual code exists once per colour component,
 *     each copy naming that component's registers (a loop that rotated the
 *     sets cost moves and 16 spilled registers).
 */
#ifndef H264_AMD_CABAC_SLICE_HIP
#define H264_AMD_CABAC_SLICE_HIP

/* engine.hip includes slice_decode.hip and slice_write.hip before this file:
 * OutCtx / sd_emit_* / WaveBits / sw_activate / cabac_tables / cabac_state_init */

#define CS_WAVES_PER_BLOCK 4u
#define CS_MN_BYTES 8192u /* int8 (m, n)[1024][4] */
/* per wave */
#define CS_RING_OFF 0u
#define CS_STATE_OFF 1024u
#define CS_STAGE_OFF (1024u + WB_STATE_BYTES)
#define CS_STAGE_BYTES 2048u /* largest T2 record: 1792 B */
#define CS_ROW_OFF (CS_STAGE_OFF + CS_STAGE_BYTES)
#define CS_ROW_STRIDE 40u

/* row context of one macroblock column (what the macroblock below looks at):
 *   [0,4)   flags | cbp << 8 | DC coded_block_flags << 16
 *   [4,8)   coded_block_flag of the bottom row of luma 4x4 blocks
 *   [8,16)  ChromaArrayType 3: the same for Cb, Cr; else chroma AC bottom rows
 *           (Cb at 8, 9; Cr at 10, 11)
 *   [16,28) list 0: |mvd x| and |mvd y| of the bottom row (clamped to 255),
 *           ref_idx of the bottom 8x8 blocks;  [28,40) list 1 */
#define CS_NB_VALID 0x80u
#define CS_NB_SKIP 0x01u
#define CS_NB_ICPNZ 0x02u /* intra_chroma_pred_mode != 0 */
#define CS_NB_T8 0x04u
#define CS_NB_PCM 0x08u
#define CS_NB_INXN 0x10u /* mb_type I_NxN */
#define CS_NB_SI 0x20u
#define CS_NB_BDIR 0x40u /* B_Skip or B_Direct_16x16 */

/* registers holding context variables */
enum { CX_MB0, CX_MB1, CX_SIG, CX_LAST = CX_SIG + 3, CX_ABS = CX_LAST + 3,
       CX_M8 = CX_ABS + 3, CX_REGS = CX_M8 + 3 };
/* lanes inside the per-component registers */
#define CXL_SIG8 0u /* CX_M8: significant_coeff_flag of 8x8 blocks, 15 */
#define CXL_LAST8 15u /* 9 */
#define CXL_CBF8 24u /* coded_block_flag of 8x8 blocks (4:4:4), 4 */
#define CXL_CBF 28u /* coded_block_flag DC / AC / 4x4: + 0 / 4 / 8 + inc */
#define CXL_ABS8 49u /* CX_ABS: coeff_abs_level_minus1 of 8x8 blocks, 10 */
/* CX_MB1 holds ctxIdx 64 .. 104 at lane ctxIdx - 64 and transform_size_8x8_flag
 * (399 .. 401) at lanes 41 .. 43 */
#define CXL_T8 41u

struct cabac_dec_tables {
	uint16_t slot_ctx[2][CX_REGS][64]; /* [field][register][lane] -> ctxIdx */
	uint16_t pos8[2][64]; /* Table 9-43: sig8 | last8 << 8, [field][levelListIdx] */
};

static void cabac_dec_build_tables(cabac_dec_tables *t)
{
#define H264_CRT(type, name, dims, ...) static const type crt_##name dims = __VA_ARGS__;
#include "../common/h264_cabac_residual_tables.inc"
#undef H264_CRT
	memset(t, 0xff, sizeof(*t));
	for (int f = 0; f < 2; f++) {
		for (int l = 0; l < 64; l++) {
			t->slot_ctx[f][CX_MB0][l] = (uint16_t)l;
			if (l <= 40)
				t->slot_ctx[f][CX_MB1][l] = (uint16_t)(64 + l);
			else if (l <= 43)
				t->slot_ctx[f][CX_MB1][l] = (uint16_t)(399 + l - 41);
		}
		for (int c = 0; c < 3; c++) {
			/* ctxBlockCat of the DC block of the component and of its
			 * 8x8 block */
			const int cat0 = c == 0 ? 0 : c == 1 ? 6 : 10, cat8 = 5 + 4 * c;
			const int nsl = c == 0 ? 61 : 44; /* luma set also holds chroma DC / AC */
			for (int l = 0; l < nsl; l++) {
				t->slot_ctx[f][CX_SIG + c][l] =
					(uint16_t)(crt_sig_base[f][cat0] + l);
				t->slot_ctx[f][CX_LAST + c][l] =
					(uint16_t)(crt_last_base[f][cat0] + l);
			}
			for (int l = 0; l < (c == 0 ? 49 : 30); l++)
				t->slot_ctx[f][CX_ABS + c][l] =
					(uint16_t)(crt_abs_base[cat0] + l);
			for (int l = 0; l < 10; l++)
				t->slot_ctx[f][CX_ABS + c][CXL_ABS8 + l] =
					(uint16_t)(crt_abs_base[cat8] + l);
			for (int l = 0; l < 15; l++)
				t->slot_ctx[f][CX_M8 + c][CXL_SIG8 + l] =
					(uint16_t)(crt_sig_base[f][cat8] + l);
			for (int l = 0; l < 9; l++)
				t->slot_ctx[f][CX_M8 + c][CXL_LAST8 + l] =
					(uint16_t)(crt_last_base[f][cat8] + l);
			for (int l = 0; l < 4; l++)
				t->slot_ctx[f][CX_M8 + c][CXL_CBF8 + l] =
					(uint16_t)(crt_cbf_base[cat8] + l);
			for (int l = 0; l < 12; l++)
				t->slot_ctx[f][CX_M8 + c][CXL_CBF + l] =
					(uint16_t)(crt_cbf_base[cat0] + l);
		}
		for (int l = 0; l < 64; l++)
			t->pos8[f][l] = l < 63 ? (uint16_t)(crt_sig8[f][l] |
							    (crt_last8[l] << 8))
					       : 0;
	}
}

struct cs_args {
	const uint8_t *buf;
	const struct h264_amd_slice_desc *descs;
	uint32_t n;
	const dev_sps *spsv;
	const dev_pps *ppsv;
	const dev_epoch *epochs;
	uint32_t n_sps, n_pps, n_epochs;
	const cabac_tables *tab;
	const cabac_dec_tables *dtab;
	uint32_t tier, mb_capacity, mb_stride;
	struct h264_amd_slice_result *results;
	const struct h264_slice_header *headers;
	uint8_t *mb_out;
	uint32_t row_pitch; /* macroblock columns of row context in LDS */
	uint32_t wave_lds_bytes;
	/* workgroups per dispatch layer (see slice_decode.hip, "fair shares");
	 * 0: no priority rotation */
	uint32_t prio_div;
	const uint32_t *order; /* launch position -> slice, or NULL */
	uint32_t quirks; /* bit 0: 4:4:4 streams of x264 before build 151;
			  * bit 1: context variables at SliceQPY 0 per the standard */
};

__device__ __forceinline__ uint32_t cb_hdr(const struct h264_slice_header *h,
					   uint32_t off)
{
	return wb_uniform(*reinterpret_cast<const uint32_t *>(
		reinterpret_cast<const uint8_t *>(h) + off));
}

/* ---- decoding engine (9.3.1.2, 9.3.3.2), wave-uniform ------------------------ */
#define CS_AL 23u /* codIRange and codIOffset sit in bits 31..23 */
struct CbEng {
	/* O: codIOffset << 23 | up to 22 bits of the stream read ahead | a
	 * sentinel 1 behind the last of them (zeros below);  R: codIRange << 23.
	 * Both top-aligned: "codIOffset >= codIRange - rLPS" is one 32-bit
	 * subtraction without a shift by the read-ahead count, the
	 * renormalisation shift is s_flbit(R) itself, and there is no counter to
	 * keep: the sentinel's position says how many bits are left */
	uint32_t O, R;
	/* per lane (= pStateIdx): rangeTabLPS[lane][0..3]; what an LPS / an MPS
	 * changes in a state, as an XOR on pStateIdx | !valMPS << 6 (lane ^
	 * transIdxLPS[lane], with bit 6 in lane 0 where an LPS flips valMPS; lane ^
	 * transIdxMPS[lane]); Table 9-43 by levelListIdx */
	uint32_t lps4, xl, xm, pos8;
};

__device__ __forceinline__ uint32_t cs_rl(uint32_t v, uint32_t l)
{
	return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l);
}

/* v_writelane_b32 (this clang has no builtin for it).  gfx9 allows one scalar
 * register per vector instruction, so the lane select travels in M0 */
__device__ __forceinline__ uint32_t cs_wl(uint32_t vec, uint32_t val, uint32_t l)
{
	/* both scalar by construction; said again for the register allocator */
	val = (uint32_t)__builtin_amdgcn_readfirstlane((int)val);
	l = (uint32_t)__builtin_amdgcn_readfirstlane((int)l);
	asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0"
		     : "+v"(vec)
		     : "s"(val), "s"(l)
		     : "m0");
	return vec;
}

/* the decoding engine's registers are operands of the hand-written decision
 * (scalar registers): where the optimiser has computed one on the vector side
 * it comes back here, behind the rarer operations, not in front of every bin */
__device__ __forceinline__ void cs_pin(CbEng &c)
{
	c.O = (uint32_t)__builtin_amdgcn_readfirstlane((int)c.O);
	c.R = (uint32_t)__builtin_amdgcn_readfirstlane((int)c.R);
}

/* bits read ahead of the arithmetic decoder */
__device__ __forceinline__ uint32_t cs_ahead(const CbEng &c)
{
	return 22u - (uint32_t)__builtin_ctz(c.O);
}

/* 16 more bits of read-ahead when fewer than 6 are left (a decision shifts by
 * at most 6: rangeTabLPS is 6 or more in every state a context can reach);
 * the window slides 32 bits late so that the cursor can step back by the
 * read-ahead (I_PCM, end position) without leaving it */
__device__ __forceinline__ void cs_refill_now(CbEng &c, WaveBits &wb);
__device__ __forceinline__ void cs_refill(CbEng &c, WaveBits &wb)
{
	if (__builtin_expect((c.O & 0x1ffffu) == 0u, 0))
		cs_refill_now(c, wb);
}

__device__ __forceinline__ void cs_refill_now(CbEng &c, WaveBits &wb)
{
	{
		if (__builtin_expect(wb.bitpos >= WB_SLIDE_BITS + 32u, 0)) {
			wb.base += WB_SLIDE_BITS / 8u;
			wb.bitpos -= WB_SLIDE_BITS;
			wb_refill(wb);
		}
		uint32_t v = wb_peek(wb, 0);
		wb_skip(wb, 16);
		/* the sentinel at bit s >= 17 becomes the first of 16 more bits,
		 * the new sentinel sits at s - 16 */
		const uint32_t s16 = (uint32_t)__builtin_ctz(c.O) - 16u;
		c.O += (((v >> 15) | 1u) - 0x10000u) << s16;
		cs_pin(c);
	}
}

/* 9.3.1.2 at the cursor.  False: codIOffset 510 or 511, which the standard
 * forbids -- codIOffset < codIRange would never hold again and the subtraction
 * of the decision would wrap; -EIO in oracle and product */
__device__ __forceinline__ bool cs_start(CbEng &c, WaveBits &wb, uint32_t lane)
{
	wb_sync(wb, lane);
	const uint32_t v = wb_peek(wb, lane);
	wb_skip(wb, 31);
	c.O = v | 1u; /* 9 bits of offset, 22 read ahead, the sentinel */
	c.R = 510u << CS_AL;
	cs_pin(c);
	return (v >> CS_AL) < 510u;
}

/* RBSP bits the arithmetic decoder has consumed (what a bit-serial decoder
 * would have read) */
__device__ __forceinline__ uint32_t cs_pos(const CbEng &c, const WaveBits &wb)
{
	return wb_pos(wb) - cs_ahead(c);
}

/* a read went past the end of the NAL: the oracle's reader fails there */
__device__ __forceinline__ bool cs_eof(const CbEng &c, const WaveBits &wb)
{
	return wb_all_in(wb) && cs_pos(c, wb) > wb_total_bits(wb);
}

/* DecodeDecision behind "v_readlane st, reg, m0": 9 scalar + 12 vector
 * instructions.  The scalar unit is the one the kernel waits for, the vector
 * ports have room (profiles/r03_cabac_pmc.json), so what can be computed for
 * all 64 states at once is: the rangeTabLPS column of the current range
 * (selector 4 + qCodIRangeIdx = range bits 8..6, v_perm_b32), aligned, and
 * codIRange - rLPS for every state; the state then picks rLPS and the MPS range
 * with one v_readlane each.  !valMPS comes through the vector side too (the
 * whole register shifted, one lane read).  The four vector instructions in
 * front are also the wait states the v_readlane by `st` needs (its lane select
 * was written by the v_readlane before).  REG: the register operand holding the
 * context, written with the new state */
#ifdef CS_NM_SALU
#define CS_NM_A(REG) "s_lshr_b32 %[nm], %[st], 6\n\t" /* (also a wait state) */
#define CS_NM_B
#else
#define CS_NM_A(REG) "v_lshrrev_b32 %[vnm], 6, " REG "\n\t"
#define CS_NM_B "v_readlane_b32 %[nm], %[vnm], m0\n\t" /* !valMPS */
#endif
#ifdef CS_RM_SALU
#define CS_RM_A
#define CS_RM_B "s_sub_u32 %[rm], %[R], %[rl]\n\t"
#else
#define CS_RM_A "v_sub_u32 %[vrm], %[R], %[vcol]\n\t"
#define CS_RM_B "v_readlane_b32 %[rm], %[vrm], %[st]\n\t"
#endif
/* the state change of the path taken: both tables read and one picked on the
 * scalar side (CS_X_SCALAR), or picked in all lanes under a mask made from
 * SCC and read once */
#ifdef CS_X_SCALAR
#define CS_X_SEL                                                               \
	"v_readlane_b32 %[xa], %[xl], %[st]\n\t"                                \
	"v_readlane_b32 %[xb], %[xm], %[st]\n\t"                                \
	"s_cselect_b32 %[xa], %[xb], %[xa]\n\t"
#define CS_X_GET
#else
#define CS_X_SEL                                                               \
	"s_cselect_b64 %[m64], -1, 0\n\t"                                       \
	"v_cndmask_b32_e64 %[vnm], %[xl], %[xm], %[m64]\n\t"
#define CS_X_GET "v_readlane_b32 %[xa], %[vnm], %[st]\n\t"
#endif
#define CS_DEC_TAIL(REG)                                                        \
	"v_lshrrev_b32 %[vcol], 29, %[R]\n\t" /* 4 + qCodIRangeIdx */            \
	"v_perm_b32 %[vcol], %[lps4], 0, %[vcol]\n\t"                           \
	"v_lshlrev_b32 %[vcol], 23, %[vcol]\n\t"                                \
	CS_RM_A                                                                 \
	CS_NM_A(REG)                                                            \
	"v_readlane_b32 %[rl], %[vcol], %[st]\n\t"                              \
	CS_RM_B                                                                 \
	CS_NM_B                                                                 \
	"s_sub_u32 %[t0], %[O], %[rm]\n\t" /* SCC: the MPS path */               \
	CS_X_SEL                                                                \
	"s_cselect_b32 %[O], %[O], %[t0]\n\t"                                   \
	"s_cselect_b32 %[R], %[rm], %[rl]\n\t"                                  \
	"s_addc_u32 %[bin], %[nm], 0\n\t" /* bit 0: the bin */                   \
	CS_X_GET                                                                \
	"s_xor_b32 %[xa], %[xa], %[st]\n\t"                                     \
	"v_writelane_b32 " REG ", %[xa], m0\n\t"                                \
	"s_flbit_i32_b32 %[t0], %[R]\n\t"                                       \
	"s_lshl_b32 %[R], %[R], %[t0]\n\t"                                      \
	"s_lshl_b32 %[O], %[O], %[t0]"
#define CS_DEC_BODY CS_DEC_TAIL("%[reg]")

/* 9.3.3.2.1 DecodeDecision on the context in lane `slot` of `reg`; returns the
 * bin in bit 0 (the caller looks at nothing else).
 *
 * Written out instruction by instruction: the kernel is bound by the scalar
 * unit (profiles/r03_cabac_pmc*.json: the time per bin is the scalar
 * instructions per bin times the 4.6 cycles one costs with four waves per
 * SIMD), and the compiler spends eight of them on turning the decision into a
 * lane mask and back.  Here:
 *   * a state is pStateIdx | !valMPS << 6 and serves as a lane select as it is
 *     (the hardware looks at six bits, tools/probe/lane_probe.hip);
 *   * the rangeTabLPS column of the current codIRange is picked in all 64
 *     lanes at once (v_perm_b32 with codIRange's bits 8..6 as the selector: bit
 *     8 is always set, so the selector is 4 + qCodIRangeIdx = "byte q of the
 *     first source") and aligned (v_lshlrev), the state picks its row with
 *     one v_readlane;
 *   * "codIOffset >= codIRange - rLPS" is the BORROW of one s_sub_u32 on the
 *     top-aligned registers: the three selects (offset, range, state change)
 *     hang on SCC, and the bin is !valMPS + borrow from an s_addc_u32;
 *   * the next state is the old one XOR a table value (transIdx differences,
 *     the valMPS flip of state 0 included): nothing to carry over;
 *   * renormalisation: s_flbit of the range IS the shift, for both registers;
 *   * four instructions between the first v_readlane and the next are the
 *     wait states that one needs (its lane select was written by the first);
 *     M0 written by s_mov_b32 needs none in front of a v_readlane.
 * 10 scalar + 13 vector instructions with the lane select (round 3's first
 * assembly version: 18 + 6, the compiler's: 23 + 8). */
__device__ __forceinline__ uint32_t cs_dec(CbEng &c, WaveBits &wb, uint32_t &reg,
					   uint32_t slot)
{
	cs_refill(c, wb);
	uint32_t st, rl, xa, xb, rm, t0, nm, bin, vcol, vrm, vnm;
	uint64_t m64;
	/* (the engine state is kept on the scalar side by cs_pin() behind every
	 * other operation on it) */
	slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
	cs_pin(c);
	/* the lane select travels in M0: `slot` may come straight out of a vector
	 * instruction (v_readfirstlane), and a lane select written by one needs
	 * four wait states the assembler does not see into this block to add */
	asm volatile("s_mov_b32 m0, %[slot]\n\t"
		     "v_readlane_b32 %[st], %[reg], m0\n\t"
		     CS_DEC_BODY
		     : [O] "+s"(c.O), [R] "+s"(c.R), [reg] "+v"(reg), [st] "=&s"(st),
		       [rl] "=&s"(rl), [xa] "=&s"(xa), [xb] "=&s"(xb), [m64] "=&s"(m64), [rm] "=&s"(rm),
		       [t0] "=&s"(t0), [nm] "=&s"(nm), [bin] "=&s"(bin),
		       [vcol] "=&v"(vcol), [vrm] "=&v"(vrm), [vnm] "=&v"(vnm)
		     : [slot] "s"(slot), [lps4] "v"(c.lps4), [xl] "v"(c.xl),
		       [xm] "v"(c.xm)
		     : "m0", "scc");
	return bin & 1u;
}

/* 9.3.3.2.3 DecodeBypass: 2 * codIOffset + bit against codIRange is O against
 * R / 2 (exact: R has 23 zero bits), then O moves up.  Five scalar
 * instructions */
__device__ __forceinline__ uint32_t cs_bypass(CbEng &c, WaveBits &wb)
{
	cs_refill(c, wb);
	cs_pin(c);
	uint32_t t, u, bin;
	asm volatile("s_lshr_b32 %[t], %[R], 1\n\t"
		     "s_sub_u32 %[u], %[O], %[t]\n\t" /* SCC: the bin is 0 */
		     "s_cselect_b32 %[O], %[O], %[u]\n\t"
		     "s_cselect_b32 %[bin], 0, 1\n\t"
		     "s_lshl_b32 %[O], %[O], 1"
		     : [O] "+s"(c.O), [t] "=&s"(t), [u] "=&s"(u), [bin] "=&s"(bin)
		     : [R] "s"(c.R)
		     : "scc");
	return bin;
}

/* 9.3.3.2.2.3 DecodeTerminate */
__device__ __forceinline__ uint32_t cs_term(CbEng &c, WaveBits &wb)
{
	cs_refill(c, wb);
	c.R -= 2u << CS_AL;
	if (c.O >= c.R)
		return 1;
	const uint32_t n = (uint32_t)__builtin_clz(c.R);
	c.R <<= n;
	c.O <<= n;
	cs_pin(c);
	return 0;
}

/* bypass bins that are 1, up to `maxn` of them and the 0 behind them (the
 * prefix of an Exp-Golomb suffix); six scalar instructions per bin */
__device__ __forceinline__ uint32_t cs_bypass_ones(CbEng &c, WaveBits &wb,
						   uint32_t maxn)
{
	uint32_t n = 0, t, u, again;
	maxn = (uint32_t)__builtin_amdgcn_readfirstlane((int)maxn);
	for (;;) {
		cs_pin(c);
		asm volatile("1:\n\t"
			     "s_and_b32 %[t], %[O], 0x1ffff\n\t"
			     "s_cbranch_scc0 8f\n\t"
			     "s_lshr_b32 %[t], %[R], 1\n\t"
			     "s_sub_u32 %[u], %[O], %[t]\n\t"
			     "s_cbranch_scc1 5f\n\t" /* a 0: done */
			     "s_lshl_b32 %[O], %[u], 1\n\t"
			     "s_add_u32 %[n], %[n], 1\n\t"
			     "s_cmp_lt_u32 %[n], %[maxn]\n\t"
			     "s_cbranch_scc1 1b\n\t"
			     "s_mov_b32 %[again], 0\n\t"
			     "s_branch 7f\n"
			     "5:\n\t"
			     "s_lshl_b32 %[O], %[O], 1\n\t"
			     "s_mov_b32 %[again], 0\n\t"
			     "s_branch 7f\n"
			     "8:\n\t"
			     "s_mov_b32 %[again], 1\n"
			     "7:"
			     : [O] "+s"(c.O), [n] "+s"(n), [t] "=&s"(t), [u] "=&s"(u),
			       [again] "=&s"(again)
			     : [R] "s"(c.R), [maxn] "s"(maxn)
			     : "scc");
		if (!again)
			return n;
		cs_refill_now(c, wb);
	}
}

/* nb <= 31 bypass bins as a number, first bin in the highest place; seven
 * scalar instructions per bin (the bins are collected inverted: the borrow of
 * the compare is the carry of an s_addc) */
__device__ __forceinline__ uint32_t cs_bypass_bits(CbEng &c, WaveBits &wb,
						   uint32_t nb)
{
	if (nb == 0)
		return 0;
	uint32_t acc = 0, t, u, again;
	uint32_t left = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
	for (;;) {
		cs_pin(c);
		asm volatile("1:\n\t"
			     "s_and_b32 %[t], %[O], 0x1ffff\n\t"
			     "s_cbranch_scc0 8f\n\t"
			     "s_lshr_b32 %[t], %[R], 1\n\t"
			     "s_sub_u32 %[u], %[O], %[t]\n\t" /* SCC: the bin is 0 */
			     "s_cselect_b32 %[O], %[O], %[u]\n\t"
			     "s_addc_u32 %[acc], %[acc], %[acc]\n\t"
			     "s_lshl_b32 %[O], %[O], 1\n\t"
			     "s_sub_u32 %[left], %[left], 1\n\t"
			     "s_cmp_lg_u32 %[left], 0\n\t"
			     "s_cbranch_scc1 1b\n\t"
			     "s_mov_b32 %[again], 0\n\t"
			     "s_branch 7f\n"
			     "8:\n\t"
			     "s_mov_b32 %[again], 1\n"
			     "7:"
			     : [O] "+s"(c.O), [acc] "+s"(acc), [left] "+s"(left),
			       [t] "=&s"(t), [u] "=&s"(u), [again] "=&s"(again)
			     : [R] "s"(c.R)
			     : "scc");
		if (!again)
			return ~acc & ((1u << nb) - 1u);
		cs_refill_now(c, wb);
	}
}

/* suffix of UEGk (9.3.2.3); ok cleared after 24 prefix bits */
__device__ __forceinline__ uint32_t cs_egk(CbEng &c, WaveBits &wb, uint32_t k,
					   bool &ok)
{
	const uint32_t p = cs_bypass_ones(c, wb, 24);
	if (p >= 24u) {
		ok = false;
		return 0;
	}
	/* 2^k + 2^(k+1) + ... for the p ones, then k + p bits */
	return (((1u << p) - 1u) << k) + cs_bypass_bits(c, wb, k + p);
}

/* ---- context registers -------------------------------------------------------- */
struct CbCtx {
	uint32_t mb0, mb1;
	uint32_t sig[3], last[3], abs[3], m8[3];
};

/* ---- per-lane roles in the neighbour grids ---------------------------------- */
#define CS_ROLE_TOP 1u /* border: the macroblock above */
#define CS_ROLE_LEFT 2u /* border: the macroblock to the left */
#define CS_ROLE_BOT 4u /* bottom row of the current macroblock */
#define CS_ROLE_IN 8u /* a block of the current macroblock */

struct CbLanesWide {
	/* coded_block_flag grid, register 0: luma 5x5 at lanes 0..24; without
	 * 4:4:4 the chroma AC grids (3 wide) at 25.. (Cb) and 40.. (Cr) */
	uint32_t g_role, g_off, g_src;
	/* list registers: |mvd x| 5x5 at 0..24, |mvd y| at 25..49, ref_idx 3x3
	 * at 50..58 */
	uint32_t l_role, l_off, l_src;
	int32_t l_x, l_y; /* block coordinates of the lane, -1 on a border */
	uint32_t l_kind; /* 0 mvd x, 1 mvd y, 2 ref_idx, 3 nothing */
};

__device__ __forceinline__ void cs_lane_roles_wide(CbLanesWide &L, uint32_t lane,
						   uint32_t cat)
{
	L.g_role = 0;
	L.g_off = 0;
	L.g_src = lane;
	if (lane < 25u) {
		const uint32_t gy = lane / 5u, gx = lane % 5u;
		if (gy == 0 && gx >= 1) {
			L.g_role = CS_ROLE_TOP;
			L.g_off = 4u + gx - 1u;
		} else if (gx == 0 && gy >= 1) {
			L.g_role = CS_ROLE_LEFT;
			L.g_src = lane + 4u;
		} else if (gx >= 1 && gy >= 1) {
			L.g_role = CS_ROLE_IN | (gy == 4 ? CS_ROLE_BOT : 0u);
			L.g_off = 4u + gx - 1u;
		}
	} else if (lane < 55u && (cat == 1 || cat == 2)) {
		const uint32_t c = (lane - 25u) / 15u, l2 = (lane - 25u) % 15u;
		const uint32_t gy = l2 / 3u, gx = l2 % 3u, h = cat == 1 ? 2u : 4u;
		if (gy == 0 && gx >= 1) {
			L.g_role = CS_ROLE_TOP;
			L.g_off = 8u + 2u * c + gx - 1u;
		} else if (gx == 0 && gy >= 1 && gy <= h) {
			L.g_role = CS_ROLE_LEFT;
			L.g_src = lane + 2u;
		} else if (gx >= 1 && gy >= 1 && gy <= h) {
			L.g_role = CS_ROLE_IN | (gy == h ? CS_ROLE_BOT : 0u);
			L.g_off = 8u + 2u * c + gx - 1u;
		}
	}
	L.l_role = 0;
	L.l_off = 0;
	L.l_src = lane;
	L.l_x = L.l_y = -1;
	L.l_kind = 3;
	if (lane < 50u) {
		const uint32_t k = lane / 25u, l2 = lane % 25u;
		const uint32_t gy = l2 / 5u, gx = l2 % 5u;
		L.l_kind = k;
		if (gy == 0 && gx >= 1) {
			L.l_role = CS_ROLE_TOP;
			L.l_off = 16u + 4u * k + gx - 1u;
		} else if (gx == 0 && gy >= 1) {
			L.l_role = CS_ROLE_LEFT;
			L.l_src = lane + 4u;
		} else if (gx >= 1 && gy >= 1) {
			L.l_role = CS_ROLE_IN | (gy == 4 ? CS_ROLE_BOT : 0u);
			L.l_off = 16u + 4u * k + gx - 1u;
			L.l_x = (int32_t)gx - 1;
			L.l_y = (int32_t)gy - 1;
		}
	} else if (lane < 59u) {
		const uint32_t l2 = lane - 50u, gy = l2 / 3u, gx = l2 % 3u;
		L.l_kind = 2;
		if (gy == 0 && gx >= 1) {
			L.l_role = CS_ROLE_TOP;
			L.l_off = 24u + gx - 1u;
		} else if (gx == 0 && gy >= 1) {
			L.l_role = CS_ROLE_LEFT;
			L.l_src = lane + 2u;
		} else if (gx >= 1 && gy >= 1) {
			L.l_role = CS_ROLE_IN | (gy == 2 ? CS_ROLE_BOT : 0u);
			L.l_off = 24u + gx - 1u;
			L.l_x = (int32_t)gx - 1;
			L.l_y = (int32_t)gy - 1;
		}
	}
}

/* the same, packed: nine per-lane constants in two registers (they are looked
 * at a few times per macroblock; kept apart they cost the kernel six registers
 * it does not have, spilled and reloaded around every residual).
 * gw: g_role | g_off << 4 | g_src << 10 | l_kind << 16 | (l_x + 1) << 18 |
 *     (l_y + 1) << 21;   lw: l_role | l_off << 4 | l_src << 10 */
struct CbLanes {
	uint32_t gw, lw;
};

__device__ __forceinline__ void cs_lane_roles(CbLanes &P, uint32_t lane, uint32_t cat)
{
	CbLanesWide L;
	cs_lane_roles_wide(L, lane, cat);
	P.gw = L.g_role | (L.g_off << 4) | (L.g_src << 10) | (L.l_kind << 16) |
	       ((uint32_t)(L.l_x + 1) << 18) | ((uint32_t)(L.l_y + 1) << 21);
	P.lw = L.l_role | (L.l_off << 4) | (L.l_src << 10);
}

#define CL_G_ROLE(P) ((P).gw & 15u)
#define CL_G_OFF(P) (((P).gw >> 4) & 63u)
#define CL_G_SRC(P) (((P).gw >> 10) & 63u)
#define CL_L_KIND(P) (((P).gw >> 16) & 3u)
#define CL_L_X(P) ((int32_t)(((P).gw >> 18) & 7u) - 1)
#define CL_L_Y(P) ((int32_t)(((P).gw >> 21) & 7u) - 1)
#define CL_L_ROLE(P) ((P).lw & 15u)
#define CL_L_OFF(P) (((P).lw >> 4) & 63u)
#define CL_L_SRC(P) (((P).lw >> 10) & 63u)

/* lane of a grid cell; x, y may be -1 */
__device__ __forceinline__ uint32_t cs_g5(uint32_t x, uint32_t y)
{
	return (y + 1u) * 5u + x + 1u;
}

/* the border of a grid register for a new macroblock: the upper macroblock's
 * bottom row from the row context, the left macroblock's right column out of
 * the previous register; the inside starts as zeros.  fill: value of a border
 * cell whose macroblock is not available */
__device__ __forceinline__ uint32_t cs_grid_open(uint32_t prev, uint32_t role,
						 uint32_t off, uint32_t src,
						 const uint8_t *rowent, bool availA,
						 bool availB, uint32_t fill)
{
	uint32_t top = rowent[off];
	uint32_t left = (uint32_t)__shfl((int)prev, (int)src);
	uint32_t v = 0;
	v = (role & CS_ROLE_TOP) ? (availB ? top : fill) : v;
	v = (role & CS_ROLE_LEFT) ? (availA ? left : fill) : v;
	return v;
}

/* ---- the state of one slice --------------------------------------------------- */
struct CbSlice {
	uint32_t type, cat, W, field, t8mode, direct8x8, bdl, bdc, pcm_chroma;
	uint32_t nref0, nref1;
	uint32_t is_b, is_i, is_si;
	uint32_t x264_444; /* H264_AMD_READER_FLAGS_CABAC_X264_PRE151 */
};

/* ---- phase timers (diagnostic build only: -DCS_PHASE_TIMERS) -----------------
 * tools/phase_timers.py --cabac: cycles (s_memtime) between consecutive marks,
 * summed per phase in lane `phase` of one vector register, calls counted next
 * to them; the sums replace the first 64 T0 words of the slice. */
#ifdef CS_PHASE_TIMERS
struct CsTimers {
	uint32_t acc, cnt, lane; /* per lane */
	uint32_t last; /* scalar */
};
#define CS_TARG , CsTimers &T
#define CS_TPASS , T
#define CS_T(k)                                                                \
	do {                                                                   \
		uint32_t t_ = (uint32_t)__builtin_readcyclecounter();          \
		uint32_t d_ = t_ - T.last;                                     \
		T.last = t_;                                                   \
		T.acc += T.lane == (uint32_t)(k) ? d_ : 0u;                    \
		T.cnt += T.lane == (uint32_t)(k) ? 1u : 0u;                    \
	} while (0)
#else
#define CS_TARG
#define CS_TPASS
#define CS_T(k)                                                                \
	do {                                                                   \
	} while (0)
#endif

/* ---- a unary run (TU / U prefixes: coeff_abs_level_minus1, mvd, ref_idx,
 * mb_qp_delta) -----------------------------------------------------------------
 * Decisions on lane min(lane0 + k, cap) of `reg`, k = 0, 1, ..., until a bin is
 * 0 or `maxn` bins were 1; returns the number of 1 bins.  The lane select
 * stays in M0 between the bins; per bin 17 scalar instructions and three
 * branches (a compiled loop around cs_dec(): 20 to 22 and three to five). */
__device__ __forceinline__ uint32_t cs_run(CbEng &c, WaveBits &wb, uint32_t &reg,
					   uint32_t lane0, uint32_t cap, uint32_t maxn)
{
	uint32_t st, rl, xa, xb, rm, t0, nm, bin, vcol, vrm, vnm, again;
	uint64_t m64;
	uint32_t cnt = 0;
	lane0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)lane0);
	cap = (uint32_t)__builtin_amdgcn_readfirstlane((int)cap);
	maxn = (uint32_t)__builtin_amdgcn_readfirstlane((int)maxn);
	for (;;) {
		cs_pin(c);
		asm volatile("s_mov_b32 m0, %[lane]\n"
			     "1:\n\t"
			     "s_and_b32 %[t0], %[O], 0x1ffff\n\t"
			     "s_cbranch_scc0 8f\n\t"
			     "v_readlane_b32 %[st], %[reg], m0\n\t"
			     CS_DEC_TAIL("%[reg]") "\n\t"
			     "s_bitcmp1_b32 %[bin], 0\n\t"
			     "s_cbranch_scc0 6f\n\t"
			     "s_add_u32 %[cnt], %[cnt], 1\n\t"
			     "s_add_u32 m0, m0, 1\n\t"
			     "s_min_u32 m0, m0, %[cap]\n\t"
			     "s_cmp_lt_u32 %[cnt], %[maxn]\n\t"
			     "s_cbranch_scc1 1b\n"
			     "6:\n\t"
			     "s_mov_b32 %[again], 0\n\t"
			     "s_branch 7f\n"
			     "8:\n\t"
			     "s_mov_b32 %[again], 1\n"
			     "7:\n\t"
			     "s_mov_b32 %[lane], m0"
			     : [O] "+s"(c.O), [R] "+s"(c.R), [reg] "+v"(reg),
			       [lane] "+s"(lane0), [cnt] "+s"(cnt), [st] "=&s"(st),
			       [rl] "=&s"(rl), [xa] "=&s"(xa), [xb] "=&s"(xb),
			       [m64] "=&s"(m64), [rm] "=&s"(rm), [t0] "=&s"(t0), [nm] "=&s"(nm),
			       [bin] "=&s"(bin), [vcol] "=&v"(vcol), [vrm] "=&v"(vrm), [vnm] "=&v"(vnm),
			       [again] "=&s"(again)
			     : [cap] "s"(cap), [maxn] "s"(maxn), [lps4] "v"(c.lps4),
			       [xl] "v"(c.xl), [xm] "v"(c.xm)
			     : "m0", "scc");
		if (!again)
			return cnt;
		cs_refill_now(c, wb);
	}
}

/* ---- the significance map of a 4x4 block (7.3.5.3.3, first loop) ---------------
 * significant_coeff_flag and last_significant_coeff_flag of coefficient i use
 * the same ctxIdxInc (i), i.e. the same LANE of two registers: the lane select
 * stays in M0 and counts up, nothing is set up per bin.  One pass of this block
 * runs until the map is complete (0: a last flag was 1; 1: the coefficient
 * before the final one was reached, the final one is inferred) or the
 * read-ahead has to be topped up (2: in front of a significant_coeff_flag, 3:
 * in front of a last_significant_coeff_flag; the caller refills and comes
 * back with `phase` = reason - 2).  `sig` collects the significant positions
 * by LANE (64 bits).  Per coefficient without a level: 16 scalar instructions
 * and three branches (the compiler's loop around two cs_dec(): 22 and five). */
__device__ __forceinline__ uint32_t cs_sigmap_run(CbEng &c, uint32_t &sreg,
						  uint32_t &lreg, uint32_t &lane_io,
						  uint32_t end_lane, uint32_t phase,
						  uint64_t &sig)
{
	uint32_t st, rl, xa, xb, rm, t0, nm, bin, vcol, vrm, vnm, reason;
	uint64_t m64;
	cs_pin(c);
	lane_io = (uint32_t)__builtin_amdgcn_readfirstlane((int)lane_io);
	end_lane = (uint32_t)__builtin_amdgcn_readfirstlane((int)end_lane);
	phase = (uint32_t)__builtin_amdgcn_readfirstlane((int)phase);
	asm volatile("s_mov_b32 m0, %[lane]\n\t"
		     "s_cmp_lg_u32 %[phase], 0\n\t"
		     "s_cbranch_scc1 3f\n"
		     "1:\n\t" /* significant_coeff_flag */
		     "s_and_b32 %[t0], %[O], 0x1ffff\n\t"
		     "s_cbranch_scc0 8f\n\t"
		     "v_readlane_b32 %[st], %[sreg], m0\n\t"
		     CS_DEC_TAIL("%[sreg]") "\n\t"
		     "s_bitcmp1_b32 %[bin], 0\n\t"
		     "s_cbranch_scc0 5f\n\t"
		     "s_bitset1_b64 %[sig], m0\n"
		     "3:\n\t" /* last_significant_coeff_flag */
		     "s_and_b32 %[t0], %[O], 0x1ffff\n\t"
		     "s_cbranch_scc0 9f\n\t"
		     "v_readlane_b32 %[st], %[lreg], m0\n\t"
		     CS_DEC_TAIL("%[lreg]") "\n\t"
		     "s_bitcmp1_b32 %[bin], 0\n\t"
		     "s_cbranch_scc1 6f\n"
		     "5:\n\t" /* next coefficient */
		     "s_add_u32 m0, m0, 1\n\t"
		     "s_cmp_lt_u32 m0, %[end]\n\t"
		     "s_cbranch_scc1 1b\n\t"
		     "s_mov_b32 %[reason], 1\n\t"
		     "s_branch 7f\n"
		     "6:\n\t"
		     "s_mov_b32 %[reason], 0\n\t"
		     "s_branch 7f\n"
		     "8:\n\t"
		     "s_mov_b32 %[reason], 2\n\t"
		     "s_branch 7f\n"
		     "9:\n\t"
		     "s_mov_b32 %[reason], 3\n"
		     "7:\n\t"
		     "s_mov_b32 %[lane], m0"
		     : [O] "+s"(c.O), [R] "+s"(c.R), [sreg] "+v"(sreg),
		       [lreg] "+v"(lreg), [lane] "+s"(lane_io), [sig] "+s"(sig),
		       [st] "=&s"(st), [rl] "=&s"(rl), [xa] "=&s"(xa), [xb] "=&s"(xb),
		       [m64] "=&s"(m64), [rm] "=&s"(rm), [t0] "=&s"(t0), [nm] "=&s"(nm),
		       [bin] "=&s"(bin), [vcol] "=&v"(vcol), [vrm] "=&v"(vrm), [vnm] "=&v"(vnm), [reason] "=&s"(reason)
		     : [end] "s"(end_lane), [phase] "s"(phase), [lps4] "v"(c.lps4),
		       [xl] "v"(c.xl), [xm] "v"(c.xm)
		     : "m0", "scc");
	return reason;
}

/* ---- residual_block_cabac (7.3.5.3.3) ------------------------------------------
 * IS8: an 8x8 block (contexts by Table 9-43, all in the M8 register);
 * CDC: chroma DC (ctxIdxInc = Min(levelListIdx / NumC8x8, 2)).
 * sreg / lreg / areg: the registers with this block's significant_coeff_flag,
 * last_significant_coeff_flag and coeff_abs_level_minus1 contexts, s_off /
 * a_off the first lane of its category.  Returns the coefficients as a per-lane
 * value (lane = scanning position), 16 bits each. */
template <bool IS8, bool CDC>
__device__ __forceinline__ uint32_t
cs_block(CbEng &c, WaveBits &wb, uint32_t &sreg, uint32_t &lreg, uint32_t &areg,
	 uint32_t s_off, uint32_t a_off, uint32_t numc, uint32_t nc8, bool &ok
	 CS_TARG)
{
	/* positions with a coefficient: 64 bits only where a block has them */
	typedef typename std::conditional<IS8, uint64_t, uint32_t>::type mask_t;
	mask_t sig = 0;
	const uint32_t nm1 = numc - 1u;
	if (!IS8 && !CDC) {
		uint64_t sl = 0;
		uint32_t ln = s_off, phase = 0, why;
		while ((why = cs_sigmap_run(c, sreg, lreg, ln, s_off + nm1, phase,
					    sl)) >= 2u) {
			phase = why - 2u;
			cs_refill_now(c, wb);
		}
		sig = (mask_t)(sl >> s_off);
		if (why == 1u)
			sig |= (mask_t)1 << nm1; /* the last coefficient is inferred */
	} else
	/* (no "last" flag carried out of the loop: the compiler turns one into
	 * 64-bit lane masks and four scalar instructions per bin) */
	for (uint32_t i = 0;; i++) {
		if (i >= nm1) {
			/* the last coefficient is inferred */
			sig |= (mask_t)1 << nm1;
			break;
		}
		uint32_t si, li;
		if (IS8) {
			const uint32_t p = cs_rl(c.pos8, i);
			si = CXL_SIG8 + (p & 0xffu);
			li = CXL_LAST8 + (p >> 8);
		} else if (CDC) {
			const uint32_t q = i / nc8;
			si = li = s_off + (q < 2u ? q : 2u);
		} else {
			si = li = s_off + i;
		}
		if (cs_dec(c, wb, sreg, si)) {
			sig |= (mask_t)1 << i;
			if (cs_dec(c, wb, lreg, li))
				break;
		}
	}
	CS_T(12);
	uint32_t coef = 0, eq1 = 0, gt1 = 0;
	while (sig) {
		const uint32_t p = IS8 ? 63u - (uint32_t)__builtin_clzll(sig)
				       : 31u - (uint32_t)__builtin_clz((uint32_t)sig);
		sig &= ~((mask_t)1 << p);
		uint32_t a = 0;
		if (cs_dec(c, wb, areg,
			   a_off + (gt1 ? 0u : (eq1 < 3u ? 1u + eq1 : 4u)))) {
			const uint32_t lim = CDC ? 3u : 4u;
			const uint32_t cn = a_off + 5u + (gt1 < lim ? gt1 : lim);
			a = 1u + cs_run(c, wb, areg, cn, cn, 13);
			if (a == 14u)
				a += cs_egk(c, wb, 0, ok);
			gt1++;
		} else {
			eq1++;
		}
		const uint32_t sgn = cs_bypass(c, wb);
		const uint32_t v = a + 1u;
		coef = cs_wl(coef, (sgn ? 0u - v : v) & 0xffffu, p);
	}
	CS_T(13);
	return coef;
}

/* the slice data of one CABAC slice; returns the status */
__device__ __forceinline__ int
cs_slice_data(CbEng &c, WaveBits &wb, CbCtx &x, uint8_t *stage, uint8_t *row,
	      OutCtx &o, uint32_t lane, const CbSlice &S, uint32_t first_mb,
	      uint32_t mb_capacity, uint32_t &mb_count, uint32_t prio_slot,
	      uint32_t prio_on CS_TARG)
{
	/* cabac_alignment_one_bit */
	{
		uint32_t p = wb_pos(wb) & 7u;
		if (p) {
			if (wb_all_in(wb) && wb_pos(wb) + 8u - p > wb_total_bits(wb))
				return -EIO;
			if (wb_u(wb, lane, 8u - p) != (1u << (8u - p)) - 1u)
				return -EIO;
		}
	}
	{
		const bool legal = cs_start(c, wb, lane);
		if (cs_eof(c, wb) || !legal)
			return -EIO;
	}

	CbLanes LN;
	cs_lane_roles(LN, lane, S.cat);
	const bool c12 = S.cat == 1 || S.cat == 2;
	const bool c444 = S.cat == 3;
	const uint32_t nc8 = S.cat == 2 ? 2u : 1u; /* NumC8x8 */
	const uint32_t nchroma = 4u * nc8; /* chroma blocks per component */
	const uint32_t field_flag = S.field ? (uint32_t)H264_AMD_MB_FIELD : 0u;
	const uint32_t skip_word0 =
		(S.is_b ? (uint32_t)H264_MB_TYPE_B_SKIP : (uint32_t)H264_MB_TYPE_P_SKIP) |
		(((uint32_t)H264_AMD_MB_SKIPPED | field_flag) << 16);
	const uint32_t skip_ctx = S.is_b ? 24u : 11u;

	uint32_t mbx = first_mb % S.W, mbidx = 0;
	uint32_t left = 0; /* row word 0 of the previous macroblock */
	uint32_t prev_qpd_nz = 0;
	/* neighbour grids of the previous macroblock (left borders come out of
	 * them): coded_block_flag per component, the two lists */
	uint32_t G[3] = {0, 0, 0}, LL[2] = {0, 0};

	for (;;) {
		if (mbidx >= mb_capacity)
			return H264_AMD_E_MB_OVERFLOW;
		if (prio_on && (mbidx & 63u) == 0u) {
			switch (((mbidx >> 6) + prio_slot) & 3u) {
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
		asm volatile("" : "+v"(LN.gw), "+v"(LN.lw));
		/* neighbours inside the slice (6.4.9; ref src/h264_macroblock.c:306-351
		 * for the same rule) */
		uint8_t *rowent = row + CS_ROW_STRIDE * mbx;
		const bool availA = mbx != 0 && mbidx >= 1;
		const bool availB = mbidx >= S.W;
		const uint32_t nbA = availA ? left : 0u;
		const uint32_t nbB =
			availB ? wb_uniform(*reinterpret_cast<const uint32_t *>(rowent)) : 0u;
		const uint32_t vA = nbA >> 7 & 1u, vB = nbB >> 7 & 1u;
		uint32_t me = CS_NB_VALID;
		uint32_t skip = 0;

		CS_T(0);
		if (!S.is_i) {
			/* mb_skip_flag: neighbour available and not skipped */
			skip = cs_dec(c, wb, x.mb0,
				      skip_ctx + (vA & ~nbA) + (vB & ~nbB));
		}
		CS_T(1);
		if (skip) {
			if (cs_eof(c, wb))
				return -EIO;
			sd_emit_skipped(o, skip_word0, mbidx, lane);
			me |= CS_NB_SKIP | (S.is_b ? CS_NB_BDIR : 0u);
			prev_qpd_nz = 0;
			G[0] = G[1] = G[2] = 0;
			LL[0] = LL[1] = 0;
			CS_T(2);
		} else {
			/* ---- macroblock_layer (7.3.5) ---- */
			uint32_t raw = 0; /* mb_type as Tables 7-11 / 7-13 / 7-14 number it */
			uint32_t it = 0xffu; /* I slice numbering of an intra type */
			{
				/* the bin string of the intra types at ctxIdxOffset
				 * 3 / 17 / 32; c0: context of its first bin */
				uint32_t off = 3, c0 = 0;
				bool intra = true;
				if (S.is_i && !S.is_si) {
					c0 = 3u + (vA & ~(nbA >> 4)) + (vB & ~(nbB >> 4));
				} else if (S.is_si) {
					if (cs_dec(c, wb, x.mb0,
						   0u + (vA & ~(nbA >> 5)) + (vB & ~(nbB >> 5)))) {
						c0 = 3u + (vA & ~(nbA >> 4)) +
						     (vB & ~(nbB >> 4));
						raw = 1;
					} else {
						intra = false; /* mb_type SI: raw 0 */
						it = 0xfeu;
					}
				} else if (!S.is_b) {
					if (cs_dec(c, wb, x.mb0, 14)) {
						off = c0 = 17;
						raw = 5;
					} else {
						intra = false;
						if (!cs_dec(c, wb, x.mb0, 15))
							raw = cs_dec(c, wb, x.mb0, 16) ? 3u : 0u;
						else
							raw = cs_dec(c, wb, x.mb0, 17) ? 1u : 2u;
					}
				} else {
					intra = false;
					if (!cs_dec(c, wb, x.mb0,
						    27u + (vA & ~(nbA >> 6)) +
							    (vB & ~(nbB >> 6)))) {
						raw = 0;
					} else if (!cs_dec(c, wb, x.mb0, 27 + 3)) {
						raw = 1u + cs_dec(c, wb, x.mb0, 27 + 5);
					} else {
						uint32_t bits = cs_dec(c, wb, x.mb0, 27 + 4) << 3;
						bits |= cs_dec(c, wb, x.mb0, 27 + 5) << 2;
						bits |= cs_dec(c, wb, x.mb0, 27 + 5) << 1;
						bits |= cs_dec(c, wb, x.mb0, 27 + 5);
						if (bits < 8u) {
							raw = 3u + bits;
						} else if (bits == 13u) {
							intra = true;
							off = c0 = 32;
							raw = 23;
						} else if (bits == 14u) {
							raw = 11;
						} else if (bits == 15u) {
							raw = 22;
						} else {
							raw = 2u * bits +
							      cs_dec(c, wb, x.mb0, 27 + 5) - 4u;
						}
					}
				}
				if (intra) {
					const bool i3 = off == 3;
					if (!cs_dec(c, wb, x.mb0, c0)) {
						it = 0;
					} else if (cs_term(c, wb)) {
						it = 25;
					} else {
						uint32_t luma = cs_dec(c, wb, x.mb0,
								       i3 ? 3u + 3u : off + 1u);
						uint32_t chroma = cs_dec(c, wb, x.mb0,
									 i3 ? 3u + 4u : off + 2u);
						if (chroma)
							chroma += cs_dec(c, wb, x.mb0,
									 i3 ? 3u + 5u : off + 2u);
						uint32_t pred = cs_dec(c, wb, x.mb0,
								       i3 ? 3u + 6u : off + 3u)
								<< 1;
						pred |= cs_dec(c, wb, x.mb0,
							       i3 ? 3u + 7u : off + 3u);
						it = 1u + pred + 4u * chroma + 12u * luma;
					}
					raw += it;
				}
			}

			CS_T(3);
			/* the record's head: dword j in lane j */
			uint32_t H = 0;
			uint32_t mb_type, cbp_luma = 0, cbp_chroma = 0, t8 = 0;
			uint32_t flags = field_flag, icp = 0, i16pred = 0;
			int32_t qpd = 0;
			const bool is_intra = it != 0xffu;
			bool i16 = false;
			bool ok = true;
			bool nz_bytes = false;
			uint32_t nz_val = 0;

			if (it == 25u) {
				/* ---- I_PCM: the terminating bin ended the arithmetic
				 * codeword; pcm_alignment_zero_bit, the samples, and a
				 * fresh start of the decoding engine (9.3.1.2) ---- */
				if (cs_eof(c, wb))
					return -EIO;
				wb.bitpos -= cs_ahead(c); /* hand the read-ahead back */
				uint32_t p = wb_pos(wb) & 7u;
				if (p) {
					if (wb_all_in(wb) &&
					    wb_pos(wb) + 8u - p > wb_total_bits(wb))
						return -EIO;
					if (wb_u(wb, lane, 8u - p) != 0)
						return -EIO;
				}
				const uint32_t total = 256u + 2u * S.pcm_chroma;
				for (uint32_t part = 0; part < CS_STAGE_BYTES / 1024u; part++)
					*reinterpret_cast<uint4 *>(stage + part * 1024u +
								   16u * lane) =
						make_uint4(0, 0, 0, 0);
				for (uint32_t k0 = 0; k0 < total; k0 += 64u) {
					wb_sync(wb, lane);
					if (wb_overrun(wb))
						return -EIO;
					const uint32_t bd = k0 < 256u ? S.bdl : S.bdc;
					uint32_t off = wb.bitpos + lane * bd;
					uint32_t hi = (uint32_t)__shfl((int)wb.wreg,
								       (int)(off >> 5));
					uint32_t lo = (uint32_t)__shfl(
						(int)wb.wreg, (int)((off >> 5) + 1u));
					uint64_t v = (((uint64_t)hi << 32) | lo)
						     << (off & 31u);
					uint32_t smp = (uint32_t)(v >> 32) >> (32u - bd);
					*reinterpret_cast<uint16_t *>(
						stage + H264_AMD_T2_HEAD + 2u * (k0 + lane)) =
						(uint16_t)smp;
					wb_skip(wb, 64u * bd);
				}
				wb_sync(wb, lane);
				if (wb_overrun(wb))
					return -EIO;
				/* (a start past the end of the data is caught by the
				 * test behind the macroblock: -EIO as well) */
				if (!cs_start(c, wb, lane))
					return -EIO;
				mb_type = H264_MB_TYPE_I_PCM;
				H = lane == 0 ? (mb_type | (flags << 16) | (raw << 24))
				    : lane >= 4 && lane < 16 ? 0x10101010u
							     : 0u;
				me |= CS_NB_PCM | (0x2fu << 8) | (7u << 16);
				prev_qpd_nz = 0;
				G[0] = G[1] = G[2] = 1;
				LL[0] = LL[1] = 0;
			} else {
				/* ---- derived macroblock variables ---- */
				/* per partition, two bits each: which lists it uses
				 * (L0 = 1, L1 = 2, Bi = 3, none / direct = 0) and the
				 * shape of its sub-macroblock partitions (0 8x8, 1 8x4,
				 * 2 4x8, 3 4x4) */
				uint32_t nparts = 1, usem = 0, shapem = 0;
				uint32_t pshape = 0; /* 0 16x16, 1 16x8, 2 8x16, 3 8x8 */
				bool direct16 = false;
				if (is_intra) {
					if (it == 0xfeu) {
						mb_type = H264_MB_TYPE_SI;
					} else if (it == 0) {
						mb_type = H264_MB_TYPE_I_NxN;
						me |= CS_NB_INXN;
					} else {
						mb_type = H264_MB_TYPE_I_16x16;
						i16 = true;
						i16pred = (it - 1u) & 3u;
						cbp_chroma = ((it - 1u) >> 2) % 3u;
						cbp_luma = it <= 12u ? 0u : 15u;
					}
					if (mb_type == H264_MB_TYPE_SI)
						me |= CS_NB_SI;
				} else if (!S.is_b) {
					mb_type = H264_MB_TYPE_P_16x16 + raw;
					pshape = raw;
					nparts = raw == 0 ? 1u : raw == 3 ? 4u : 2u;
					usem = 0x55;
				} else if (raw == 0) {
					mb_type = H264_MB_TYPE_B_Direct_16x16;
					direct16 = true;
					me |= CS_NB_BDIR;
				} else if (raw <= 3u) {
					mb_type = H264_MB_TYPE_B_16x16;
					usem = raw;
				} else if (raw <= 21u) {
					const uint32_t e = (raw - 4u) >> 1;
					mb_type = (raw & 1u) ? H264_MB_TYPE_B_8x16
							     : H264_MB_TYPE_B_16x8;
					pshape = (raw & 1u) ? 2u : 1u;
					nparts = 2;
					/* Table 7-14: L0 = 1, L1 = 2, Bi = 3 */
					usem = ((0x3f999u >> (2u * e)) & 3u) |
					       (((0x39f69u >> (2u * e)) & 3u) << 2);
				} else {
					mb_type = H264_MB_TYPE_B_8x8;
					pshape = 3;
					nparts = 4;
				}

				bool no_sub8 = true;
				uint32_t pred_lo = 0, pred_hi = 0;
				if (!is_intra) {
					/* grids of |mvd| and ref_idx */
#pragma unroll
					for (uint32_t l = 0; l < 2; l++)
						LL[l] = cs_grid_open(
							LL[l], CL_L_ROLE(LN), CL_L_OFF(LN) + 12u * l,
							CL_L_SRC(LN), rowent, vA != 0, vB != 0, 0);
				} else {
					LL[0] = LL[1] = 0;
				}

				CS_T(4);
				if (!is_intra && nparts == 4) {
					/* ---- sub_mb_pred (7.3.5.2) ---- */
#pragma unroll
					for (uint32_t p = 0; p < 4; p++) {
						uint32_t st;
						if (!S.is_b) {
							if (cs_dec(c, wb, x.mb0, 21))
								st = 0;
							else if (!cs_dec(c, wb, x.mb0, 22))
								st = 1;
							else
								st = cs_dec(c, wb, x.mb0, 23) ? 2u : 3u;
							shapem |= st << (2u * p);
						} else {
							if (!cs_dec(c, wb, x.mb0, 36)) {
								st = 0;
							} else if (!cs_dec(c, wb, x.mb0, 37)) {
								st = 1u + cs_dec(c, wb, x.mb0, 39);
							} else {
								st = 3;
								bool done = false;
								if (cs_dec(c, wb, x.mb0, 38)) {
									if (cs_dec(c, wb, x.mb0, 39)) {
										st = 11u +
										     cs_dec(c, wb, x.mb0, 39);
										done = true;
									} else {
										st += 4;
									}
								}
								if (!done) {
									st += 2u * cs_dec(c, wb, x.mb0, 39);
									st += cs_dec(c, wb, x.mb0, 39);
								}
							}
							/* Table 7-18 */
							const uint32_t u = st == 0   ? 0u
									   : st <= 3 ? st
									   : st <= 9 ? 1u + ((st - 4u) >> 1)
										     : st - 9u;
							const uint32_t sh = st <= 3   ? 0u
									    : st <= 9 ? 1u + (st & 1u)
										      : 3u;
							usem = (usem & ~(3u << (2u * p))) | (u << (2u * p));
							shapem |= sh << (2u * p);
						}
						pred_lo |= st << (8u * p);
						if (S.is_b && st == 0) {
							if (!S.direct8x8)
								no_sub8 = false;
						} else if ((shapem >> (2u * p)) & 3u) {
							no_sub8 = false;
						}
					}
				} else if (is_intra && !i16) {
					/* transform_size_8x8_flag of I_NxN, then the
					 * prediction modes */
					if (S.t8mode && mb_type == H264_MB_TYPE_I_NxN) {
						t8 = cs_dec(c, wb, x.mb1,
							    CXL_T8 + (nbA >> 2 & 1u) +
								    (nbB >> 2 & 1u));
					}
					const uint32_t n = t8 ? 4u : 16u;
					for (uint32_t b = 0; b < n; b++) {
						uint32_t nib = 0;
						if (!cs_dec(c, wb, x.mb1, 68 - 64)) {
							uint32_t v = cs_dec(c, wb, x.mb1, 69 - 64);
							v |= cs_dec(c, wb, x.mb1, 69 - 64) << 1;
							v |= cs_dec(c, wb, x.mb1, 69 - 64) << 2;
							nib = v + 1u;
						}
						if (b < 8)
							pred_lo |= nib << (4u * b);
						else
							pred_hi |= nib << (4u * (b - 8u));
					}
				}
				/* (into the record's head at once: two registers less
				 * to carry over the residual) */
				H = cs_wl(H, pred_lo, 2);
				H = cs_wl(H, pred_hi, 3);
				if (is_intra && c12) {
					/* intra_chroma_pred_mode */
					icp = cs_dec(c, wb, x.mb1,
						     64 - 64 + (nbA >> 1 & 1u) + (nbB >> 1 & 1u));
					if (icp == 1)
						icp += cs_dec(c, wb, x.mb1, 64 - 64 + 3);
					if (icp == 2)
						icp += cs_dec(c, wb, x.mb1, 64 - 64 + 3);
					me |= icp ? CS_NB_ICPNZ : 0u;
				}
				if (!is_intra && !direct16) {
					CS_T(5);
					/* ---- ref_idx (9.3.3.1.1.6), then mvd (9.3.3.1.1.7)
					 * of the partitions, list 0 before list 1 ---- */
#pragma unroll
					for (uint32_t l = 0; l < 2; l++) {
						const uint32_t nref = l ? S.nref1 : S.nref0;
						if (nref == 0)
							continue;
						uint32_t refw = 0;
						for (uint32_t p = 0; p < nparts; p++) {
							if (!((usem >> (2u * p)) & (1u << l)))
								continue;
							uint32_t x8, y8, w8, h8;
							if (pshape == 3) {
								x8 = p & 1u;
								y8 = p >> 1;
								w8 = h8 = 1;
							} else {
								x8 = pshape == 2 ? p : 0u;
								y8 = pshape == 1 ? p : 0u;
								w8 = pshape == 2 ? 1u : 2u;
								h8 = pshape == 1 ? 1u : 2u;
							}
							const uint32_t ra = cs_rl(
								LL[l], 50u + (y8 + 1u) * 3u + x8);
							const uint32_t rb = cs_rl(
								LL[l], 50u + y8 * 3u + x8 + 1u);
							uint32_t v = 0;
							if (cs_dec(c, wb, x.mb0,
								   54u + (ra > 0) + 2u * (rb > 0))) {
								/* bins on ctxIdx 58, then 59s; the
								 * 255th 1 is one too many */
								v = 1u + cs_run(c, wb, x.mb0, 54 + 4,
										54 + 5, 255);
								if (v > 255u)
									return -EIO;
							}
							refw |= v << (8u * p);
							const bool in =
								CL_L_KIND(LN) == 2 &&
								(uint32_t)(CL_L_X(LN) - (int32_t)x8) < w8 &&
								(uint32_t)(CL_L_Y(LN) - (int32_t)y8) < h8;
							LL[l] = in ? v : LL[l];
						}
						H = cs_wl(H, refw, 16u + l);
					}
					CS_T(6);
#pragma unroll
					for (uint32_t l = 0; l < 2; l++) {
						for (uint32_t p = 0; p < nparts; p++) {
							if (!((usem >> (2u * p)) & (1u << l)))
								continue;
							const uint32_t shp = (shapem >> (2u * p)) & 3u;
							const uint32_t nsub = pshape != 3 ? 1u
									      : shp == 0  ? 1u
									      : shp == 3  ? 4u
											  : 2u;
							for (uint32_t s = 0; s < nsub; s++) {
								uint32_t x4, y4, w4, h4;
								if (pshape != 3) {
									x4 = pshape == 2 ? 2u * p : 0u;
									y4 = pshape == 1 ? 2u * p : 0u;
									w4 = pshape == 2 ? 2u : 4u;
									h4 = pshape == 1 ? 2u : 4u;
								} else {
									const uint32_t sh = shp;
									x4 = 2u * (p & 1u);
									y4 = 2u * (p >> 1);
									w4 = h4 = 2;
									if (sh == 1) {
										h4 = 1;
										y4 += s;
									} else if (sh == 2) {
										w4 = 1;
										x4 += s;
									} else if (sh == 3) {
										w4 = h4 = 1;
										x4 += s & 1u;
										y4 += s >> 1;
									}
								}
								uint32_t mv[2];
#pragma unroll
								for (uint32_t k = 0; k < 2; k++) {
									const uint32_t off = k ? 47u : 40u;
									const uint32_t sum =
										cs_rl(LL[l], 25u * k +
												     cs_g5(x4 - 1u, y4)) +
										cs_rl(LL[l], 25u * k +
												     cs_g5(x4, y4 - 1u));
									const uint32_t inc =
										sum < 3u ? 0u : sum > 32u ? 2u : 1u;
									uint32_t m = 0;
									if (cs_dec(c, wb, x.mb0, off + inc)) {
										/* ctxIdxInc 3, 4, 5, 6, 6, ... */
										m = 1u + cs_run(c, wb, x.mb0, off + 3u,
												off + 6u, 8);
										if (m == 9u)
											m += cs_egk(c, wb, 3, ok);
									}
									uint32_t sv = m;
									if (m && cs_bypass(c, wb))
										sv = 0u - m;
									mv[k] = sv & 0xffffu;
									/* what the neighbours' contexts see is the
									 * magnitude of the value as it is STORED, 16
									 * bits (a damaged stream can carry more; the
									 * oracle keeps an int16_t like the
									 * reference's macroblock struct) */
									const int32_t sw = (int16_t)(sv & 0xffffu);
									const uint32_t aw = (uint32_t)(sw < 0 ? -sw : sw);
									const uint32_t am = aw < 255u ? aw : 255u;
									const bool in =
										CL_L_KIND(LN) == k &&
										(uint32_t)(CL_L_X(LN) - (int32_t)x4) < w4 &&
										(uint32_t)(CL_L_Y(LN) - (int32_t)y4) < h4;
									LL[l] = in ? am : LL[l];
								}
								if (!ok)
									return -EIO;
								H = cs_wl(H, mv[0] | (mv[1] << 16),
									  32u + 16u * l + 4u * p + s);
							}
						}
					}
					CS_T(7);
				}

				CS_T(8);
				if (!i16) {
					/* ---- coded_block_pattern (9.3.2.6, 9.3.3.1.1.4) ---- */
					const uint32_t ca = nbA >> 8, cb = nbB >> 8;
					/* a neighbouring 8x8 block without coefficients
					 * counts; I_PCM keeps 0x2f, a skipped one 0 */
					uint32_t b0 = cs_dec(c, wb, x.mb1,
							     73 - 64 + (vA & ~(ca >> 1)) +
								     2u * (vB & ~(cb >> 2)));
					uint32_t b1 = cs_dec(c, wb, x.mb1,
							     73 - 64 + (b0 ^ 1u) +
								     2u * (vB & ~(cb >> 3)));
					uint32_t b2 = cs_dec(c, wb, x.mb1,
							     73 - 64 + (vA & ~(ca >> 3)) +
								     2u * (b0 ^ 1u));
					uint32_t b3 = cs_dec(c, wb, x.mb1,
							     73 - 64 + (b2 ^ 1u) + 2u * (b1 ^ 1u));
					cbp_luma = b0 | (b1 << 1) | (b2 << 2) | (b3 << 3);
					if (c12) {
						const uint32_t cca = (ca >> 4) & 3u,
							       ccb = (cb >> 4) & 3u;
						if (cs_dec(c, wb, x.mb1,
							   77 - 64 + (cca != 0) + 2u * (ccb != 0)))
							cbp_chroma =
								1u + cs_dec(c, wb, x.mb1,
									    77 - 64 + 4 + (cca == 2) +
										    2u * (ccb == 2));
					}
					if (cbp_luma && S.t8mode &&
					    mb_type != H264_MB_TYPE_I_NxN && no_sub8 &&
					    (!direct16 || S.direct8x8))
						t8 = cs_dec(c, wb, x.mb1,
							    CXL_T8 + (nbA >> 2 & 1u) +
								    (nbB >> 2 & 1u));
				}
				const uint32_t cbp = cbp_luma | (cbp_chroma << 4);
				me |= (cbp << 8) | (t8 ? CS_NB_T8 : 0u);
				flags |= t8 ? (uint32_t)H264_AMD_MB_TRANSFORM_8X8 : 0u;

				/* record: coefficient area zeroed, head kept in H */
				for (uint32_t part = 0; part < CS_STAGE_BYTES / 1024u; part++)
					*reinterpret_cast<uint4 *>(stage + part * 1024u +
								   16u * lane) =
						make_uint4(0, 0, 0, 0);
				uint32_t NZ = 0; /* lane b: nz_coeff[b] */
				uint32_t dc = 0;

				CS_T(9);
				if (cbp || i16) {
					/* ---- mb_qp_delta (9.3.2.7, 9.3.3.1.1.5) ---- */
					uint32_t k = 0;
					if (cs_dec(c, wb, x.mb0, 60u + prev_qpd_nz)) {
						k = 1u + cs_run(c, wb, x.mb0, 60 + 2, 60 + 3, 255);
						if (k > 255u)
							return -EIO;
					}
					qpd = (k & 1u) ? (int32_t)((k + 1u) / 2u)
						       : -(int32_t)(k / 2u);
					prev_qpd_nz = k != 0 ? 1u : 0u;

					/* ---- residual (7.3.5.3) ---- */
					const uint32_t fill = is_intra ? 1u : 0u;
					const uint32_t ncomp = c444 ? 3u : 1u;
					/* grids: borders now that "intra" is known */
#pragma unroll
					for (uint32_t q = 0; q < 3; q++) {
						if (q >= ncomp)
							continue;
						G[q] = cs_grid_open(G[q], CL_G_ROLE(LN),
								    CL_G_OFF(LN) + 4u * q, CL_G_SRC(LN),
								    rowent, vA != 0, vB != 0, fill);
					}
					/* an 8x8 block coded with its own flag (4:4:4) only
					 * counts a neighbouring MACROBLOCK that was
					 * transformed 8x8 (or is I_PCM) */
					if (c444 && t8) {
						bool okA = !vA || (nbA & (CS_NB_T8 | CS_NB_PCM));
						bool okB = !vB || (nbB & (CS_NB_T8 | CS_NB_PCM));
						uint32_t off = 0;
						if (S.x264_444) {
							/* x264 before build 151 treated such a
							 * neighbour as if it were not there
							 * (libavcodec reproduces it for those
							 * streams, h264_cabac.c); not the
							 * standard, on request only */
							okA = !vA || (nbA & CS_NB_T8);
							okB = !vB || (nbB & CS_NB_T8);
							off = fill;
						}
#pragma unroll
						for (uint32_t q = 0; q < 3; q++) {
							G[q] = (CL_G_ROLE(LN) & CS_ROLE_LEFT) && !okA ? off : G[q];
							G[q] = (CL_G_ROLE(LN) & CS_ROLE_TOP) && !okB ? off : G[q];
						}
					}
					CS_T(10);
					/* the three colour components of 4:4:4 as three copies of the
					 * code, each naming its own registers (a loop that rotates
					 * the register sets costs moves and spills at its head) */
#pragma unroll
					for (uint32_t comp = 0; comp < 3; comp++) {
						if (comp >= ncomp)
							break;
						const uint32_t cbase = 256u * comp;
						if (i16) {
							/* Intra16x16DCLevel */
							const uint32_t dA =
								vA ? (nbA >> (16u + comp)) & 1u : fill;
							const uint32_t dB =
								vB ? (nbB >> (16u + comp)) & 1u : fill;
							if (cs_dec(c, wb, x.m8[comp],
								   CXL_CBF + 0u + dA + 2u * dB)) {
								CS_T(11);
								uint32_t cf = cs_block<false, false>(
									c, wb, x.sig[comp], x.last[comp],
									x.abs[comp], 0, 0, 16, 1, ok CS_TPASS);
								if (lane < 16u)
									*reinterpret_cast<uint16_t *>(
										stage + H264_AMD_T2_HEAD +
										2u * (cbase + lane)) =
										(uint16_t)cf;
								dc |= 1u << comp;
								CS_T(14);
							}
						}
						if (t8) {
							for (uint32_t b8 = 0; b8 < 4; b8++) {
								if (!((cbp_luma >> b8) & 1u))
									continue;
								const uint32_t x4 = 2u * (b8 & 1u),
									       y4 = 2u * (b8 >> 1);
								uint32_t coded = 1;
								if (c444)
									coded = cs_dec(
										c, wb, x.m8[comp],
										CXL_CBF8 +
											cs_rl(G[comp], cs_g5(x4 - 1u, y4)) +
											2u * cs_rl(G[comp], cs_g5(x4, y4 - 1u)));
								if (!coded)
									continue;
								CS_T(11);
								uint32_t cf = cs_block<true, false>(
									c, wb, x.m8[comp], x.m8[comp], x.abs[comp],
									0, CXL_ABS8, 64, 1, ok CS_TPASS);
								/* 4x4 view of the record: coefficient
								 * 4 k + j is coefficient k of array j */
								*reinterpret_cast<uint16_t *>(
									stage + H264_AMD_T2_HEAD +
									2u * (cbase + 16u * (4u * b8 + (lane & 3u)) +
									      (lane >> 2))) = (uint16_t)cf;
								const uint64_t nzm =
									__builtin_amdgcn_ballot_w64((cf & 0xffffu) != 0);
#pragma unroll
								for (uint32_t j = 0; j < 4; j++)
									NZ = cs_wl(NZ,
										   (uint32_t)__builtin_popcountll(
											   nzm & (0x1111111111111111ull << j)),
										   16u * comp + 4u * b8 + j);
								const uint32_t gl = cs_g5(x4, y4);
								G[comp] = (lane == gl || lane == gl + 1u ||
									lane == gl + 5u || lane == gl + 6u)
									       ? 1u
									       : G[comp];
								CS_T(14);
							}
						} else {
							for (uint32_t b = 0; b < 16; b++) {
								if (!((cbp_luma >> (b >> 2)) & 1u))
									continue;
								/* 6.4.3: position of luma4x4BlkIdx b */
								const uint32_t x4 = (b & 1u) | ((b >> 1) & 2u);
								const uint32_t y4 = ((b >> 1) & 1u) | ((b >> 2) & 2u);
								const uint32_t inc =
									cs_rl(G[comp], cs_g5(x4 - 1u, y4)) +
									2u * cs_rl(G[comp], cs_g5(x4, y4 - 1u));
								if (!cs_dec(c, wb, x.m8[comp],
									    CXL_CBF + (i16 ? 4u : 8u) + inc))
									continue;
								CS_T(11);
								uint32_t cf;
								if (i16)
									cf = cs_block<false, false>(
										c, wb, x.sig[comp], x.last[comp],
										x.abs[comp], 15, 10, 15, 1, ok CS_TPASS);
								else
									cf = cs_block<false, false>(
										c, wb, x.sig[comp], x.last[comp],
										x.abs[comp], 29, 20, 16, 1, ok CS_TPASS);
								const uint32_t numc = i16 ? 15u : 16u;
								if (lane < numc)
									*reinterpret_cast<uint16_t *>(
										stage + H264_AMD_T2_HEAD +
										2u * (cbase +
										      (i16 ? 16u + 15u * b : 16u * b) +
										      lane)) = (uint16_t)cf;
								NZ = cs_wl(NZ,
									   (uint32_t)__builtin_popcountll(
										   __builtin_amdgcn_ballot_w64(
											   (cf & 0xffffu) != 0)),
									   16u * comp + b);
								G[comp] = lane == cs_g5(x4, y4) ? 1u : G[comp];
								CS_T(14);
							}
						}
					}
					if (c12) {
						if (cbp_chroma & 3u) {
#pragma unroll
							for (uint32_t cc = 0; cc < 2; cc++) {
								const uint32_t dA =
									vA ? (nbA >> (17u + cc)) & 1u : fill;
								const uint32_t dB =
									vB ? (nbB >> (17u + cc)) & 1u : fill;
								if (!cs_dec(c, wb, x.mb1,
									    97 - 64 + dA + 2u * dB))
									continue;
								CS_T(11);
								uint32_t cf = cs_block<false, true>(
									c, wb, x.sig[0], x.last[0],
									x.abs[0], 44, 30, nchroma, nc8, ok CS_TPASS);
								if (lane < nchroma)
									*reinterpret_cast<uint16_t *>(
										stage + H264_AMD_T2_HEAD +
										2u * (256u + nchroma * cc + lane)) =
										(uint16_t)cf;
								dc |= 2u << cc;
								CS_T(14);
							}
						}
						if (cbp_chroma & 2u) {
#pragma unroll
							for (uint32_t cc = 0; cc < 2; cc++) {
								for (uint32_t b = 0; b < nchroma; b++) {
									const uint32_t gl =
										25u + 15u * cc +
										((b >> 1) + 1u) * 3u + (b & 1u) + 1u;
									const uint32_t inc =
										cs_rl(G[0], gl - 1u) +
										2u * cs_rl(G[0], gl - 3u);
									if (!cs_dec(c, wb, x.mb1,
										    101 - 64 + inc))
										continue;
									CS_T(11);
									uint32_t cf = cs_block<false, false>(
										c, wb, x.sig[0], x.last[0],
										x.abs[0], 47, 39, 15, 1, ok CS_TPASS);
									if (lane < 15u)
										*reinterpret_cast<uint16_t *>(
											stage + H264_AMD_T2_HEAD +
											2u * (256u + 2u * nchroma +
											      15u * (nchroma * cc + b) +
											      lane)) = (uint16_t)cf;
									NZ = cs_wl(
										NZ,
										(uint32_t)__builtin_popcountll(
											__builtin_amdgcn_ballot_w64(
												(cf & 0xffffu) != 0)),
										16u * (1u + cc) + b);
									G[0] = lane == gl ? 1u : G[0];
									CS_T(14);
								}
							}
						}
					}
					if (!ok)
						return -EIO;
					/* only the inside of a grid travels on */
#pragma unroll
					for (uint32_t q = 0; q < 3; q++)
						G[q] = (CL_G_ROLE(LN) & CS_ROLE_IN) && q < ncomp ? G[q] : 0u;
				} else {
					prev_qpd_nz = 0;
					G[0] = G[1] = G[2] = 0;
				}
				me |= dc << 16;

				CS_T(15);
				/* ---- the record ---- */
				H = lane == 0 ? (mb_type | (cbp << 8) | (flags << 16) |
						 (raw << 24))
				    : lane == 1
					    ? (((uint32_t)qpd & 0xffffu) | (icp << 16) |
					       (i16pred << 24))
						 : H;
				nz_bytes = true;
				nz_val = NZ;
			}
			if (cs_eof(c, wb))
				return -EIO;
			CS_T(16);
			*reinterpret_cast<uint32_t *>(stage + 4u * lane) = H;
			/* nz_coeff[48]: lane b holds the count of block b -- a byte
			 * each, behind the head's dwords (one LDS instruction
			 * instead of five cross-lane gathers to pack them) */
			if (nz_bytes && lane < 48u)
				stage[16u + lane] = (uint8_t)nz_val;
			sd_emit_staged(o, stage, mbidx, lane);
			CS_T(17);
		}

		/* what the next macroblocks look at: the row context of this
		 * column (bottom rows of the grids) and word 0 */
		if (skip) {
			/* nothing coded: the entry is word 0 and zeros */
			if (lane < CS_ROW_STRIDE / 4u)
				reinterpret_cast<uint32_t *>(rowent)[lane] = lane ? 0u : me;
		} else {
			const bool gb = (CL_G_ROLE(LN) & CS_ROLE_BOT) != 0;
			const bool lb = (CL_L_ROLE(LN) & CS_ROLE_BOT) != 0;
			if (gb) {
				rowent[CL_G_OFF(LN)] = (uint8_t)G[0];
				if (c444) {
					rowent[CL_G_OFF(LN) + 4u] = (uint8_t)G[1];
					rowent[CL_G_OFF(LN) + 8u] = (uint8_t)G[2];
				}
			}
			if (!S.is_i && lb) {
				rowent[CL_L_OFF(LN)] = (uint8_t)LL[0];
				rowent[CL_L_OFF(LN) + 12u] = (uint8_t)LL[1];
			}
			if (lane == 0)
				*reinterpret_cast<uint32_t *>(rowent) = me;
		}
		left = me;
		mbx++;
		if (mbx == S.W)
			mbx = 0;
		mbidx++;
		mb_count = mbidx;
		CS_T(18);
		const uint32_t end = cs_term(c, wb); /* end_of_slice_flag */
		if (cs_eof(c, wb))
			return -EIO;
		CS_T(19);
		CS_T(20);
		if (end)
			break;
	}
	return 0;
}

extern "C" __global__ void __launch_bounds__(64 * CS_WAVES_PER_BLOCK, 4)
h264_amd_k_decode_cabac(cs_args a)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds_all[];
	/* the (m, n) tables of all 1024 contexts: shared by the workgroup */
	{
		const uint4 *s = reinterpret_cast<const uint4 *>(&a.tab->mn[0][0][0]);
		uint4 *d = reinterpret_cast<uint4 *>(lds_all);
		for (uint32_t i = threadIdx.x; i < CS_MN_BYTES / 16u;
		     i += 64u * CS_WAVES_PER_BLOCK)
			d[i] = s[i];
	}
	__syncthreads();
	/* from here on the waves of the workgroup are independent */
	const int8_t *mn = reinterpret_cast<const int8_t *>(lds_all);
	const uint32_t wave = wb_uniform(threadIdx.x >> 6);
	uint8_t *lds = lds_all + CS_MN_BYTES + wave * a.wave_lds_bytes;
	uint8_t *ring = lds + CS_RING_OFF;
	uint8_t *stage = lds + CS_STAGE_OFF;
	uint8_t *row = lds + CS_ROW_OFF;
	const uint32_t lane = wb_lane();
	/* launch positions stand for the slices longest first (slice_decode.hip,
	 * h264_amd_k_order_slices) */
	const uint32_t q = blockIdx.x * CS_WAVES_PER_BLOCK + wave;
	if (q >= a.n)
		return;
	const uint32_t slice = a.order ? wb_uniform(a.order[q]) : q;
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
	CbSlice S;
	S.type = (uint32_t)r.slice_type;
	S.cat = r.chroma_array_type;
	S.W = r.pic_width_in_mbs;
	S.field = cb_hdr(hdr, WOFF(field_pic_flag)) ? 1u : 0u;
	S.is_b = S.type == H264_SLICE_TYPE_B;
	S.is_si = S.type == H264_SLICE_TYPE_SI;
	S.is_i = S.type == H264_SLICE_TYPE_I || S.is_si;
	S.nref0 = cb_hdr(hdr, WOFF(num_ref_idx_l0_active_minus1));
	S.nref1 = cb_hdr(hdr, WOFF(num_ref_idx_l1_active_minus1));
	S.t8mode = S.direct8x8 = 0;
	S.x264_444 = 0;
	S.bdl = S.bdc = 8;
	S.pcm_chroma = 0;
	if (status == 0) {
		if ((P.sps.mb_adaptive_frame_field_flag && !P.sps.frame_mbs_only_flag &&
		     !S.field) ||
		    P.pps.num_slice_groups_minus1 > 0 || S.W == 0 || S.W > a.row_pitch)
			status = H264_AMD_E_UNSUPPORTED;
		S.t8mode = P.pps.transform_8x8_mode_flag;
		S.x264_444 = a.quirks & 1u;
		S.direct8x8 = P.sps.direct_8x8_inference_flag;
		S.bdl = P.sps.bit_depth_luma;
		S.bdc = P.sps.bit_depth_chroma;
		S.pcm_chroma = S.cat == 1 ? 64u : S.cat == 2 ? 128u : S.cat == 3 ? 256u : 0u;
	}

	WaveBits wb;
	wb_open(wb, a.buf + d.offset, d.size, ring,
		reinterpret_cast<uint32_t *>(lds + CS_STATE_OFF), lane);
#ifdef CS_PHASE_TIMERS
	CsTimers T;
	T.acc = T.cnt = 0;
	T.lane = lane;
	T.last = (uint32_t)__builtin_readcyclecounter();
#endif
	CbEng c;
	c.O = 1;
	c.R = 510u << CS_AL;
	if (status == 0) {
		/* to the end of the slice header, where the first kernel stopped */
		uint32_t left = r.end_bit;
		while (left) {
			uint32_t k = left < 512u ? left : 512u;
			wb_skip(wb, k);
			wb_sync(wb, lane);
			left -= k;
		}
		/* 9.3.1.1: the context variables, from the (m, n) tables in LDS
		 * straight into the lanes of their registers */
		CbCtx x;
		{
			const uint32_t idc = cb_hdr(hdr, WOFF(cabac_init_idc));
			const bool valid = S.is_i || idc <= 2;
			const uint32_t col = S.is_i ? 0u : 1u + (valid ? idc : 0u);
			/* SliceQPY clipped the reference's way, or the standard's
			 * (quirk bit 1, H264_AMD_READER_FLAGS_CABAC_STD_QP) */
			const int32_t qlo = (a.quirks & 2u) ? 0 : 1;
			const int32_t iq = r.slice_qp < qlo ? qlo :
					   r.slice_qp > 51 ? 51 : r.slice_qp;
			uint32_t regs[CX_REGS];
#pragma unroll
			for (uint32_t q = 0; q < CX_REGS; q++) {
				const uint32_t ci = a.dtab->slot_ctx[S.field][q][lane];
				const uint32_t cj = ci < 1024u ? ci : 0u;
				uint32_t s = cabac_state_init_q(iq,
							mn[8u * cj + 2u * col],
							mn[8u * cj + 2u * col + 1u]);
				/* pStateIdx | !valMPS << 6 (cs_dec) */
				s = (s & 63u) | (((s >> 8) ^ 1u) << 6);
				/* (cabac_init_idc > 2: all states 0, valMPS 0) */
				regs[q] = valid && ci < 1024u ? s : 64u;
			}
			x.mb0 = regs[CX_MB0];
			x.mb1 = regs[CX_MB1];
#pragma unroll
			for (uint32_t q = 0; q < 3; q++) {
				x.sig[q] = regs[CX_SIG + q];
				x.last[q] = regs[CX_LAST + q];
				x.abs[q] = regs[CX_ABS + q];
				x.m8[q] = regs[CX_M8 + q];
			}
		}
		{
			const uint8_t *q = &a.tab->range_lps[lane][0];
			c.lps4 = (uint32_t)q[0] | ((uint32_t)q[1] << 8) |
				 ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
			c.xl = (lane ^ (uint32_t)a.tab->next_lps[lane]) |
			       (lane == 0 ? 64u : 0u);
			c.xm = lane ^ (uint32_t)a.tab->next_mps[lane];
			c.pos8 = a.dtab->pos8[S.field][lane];
		}
		status = cs_slice_data(c, wb, x, stage, row, o, lane, S,
				       r.first_mb_in_slice, a.mb_capacity, mb_count,
				       a.prio_div ? (blockIdx.x / a.prio_div) & 3u : 0u,
				       a.prio_div ? 1u : 0u CS_TPASS);
	}

	/* T0: flush the partial last group of 64 words */
	if (o.tier == H264_AMD_TIER_T0 && (mb_count & 63u) != 0) {
		uint32_t base = mb_count & ~63u;
		uint32_t l = lane < (mb_count & 63u) ? lane : (mb_count & 63u) - 1u;
		uint32_t v = (uint32_t)__shfl((int)o.t0acc, (int)l);
		*reinterpret_cast<uint32_t *>(o.mb_out + (size_t)(base + l) * o.mb_stride) = v;
	}
#ifdef CS_PHASE_TIMERS
	if (o.tier == H264_AMD_TIER_T0 && mb_count >= 64u && lane < 32u) {
		*reinterpret_cast<uint32_t *>(o.mb_out + (size_t)lane * o.mb_stride) = T.acc;
		*reinterpret_cast<uint32_t *>(o.mb_out + (size_t)(32u + lane) * o.mb_stride) =
			T.cnt;
	}
#endif
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
	r.end_bit = status == 0 ? cs_pos(c, wb) : 0u;
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

#endif /* H264_AMD_CABAC_SLICE_HIP */
