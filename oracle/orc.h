/**
 * oracle/ -- CPU restatement of the reference algorithm.  TEST INFRASTRUCTURE.
 *
 * Plain-C, bit-serial restatement of Parrot-Developers/libh264's reader,
 * CAVLC slice_data parser, writer and CABAC encoder, function by function,
 * each citing the reference file:line it follows.  It exists to CHECK the HIP
 * path; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it.  The product (libh264_amd/) never links, imports or executes
 * anything in this directory.
 *
 * PINNING STATUS -- "parity unpinned" except where stated:
 *   The reference cannot be built in this image without writing a stand-in
 *   for <ulog.h> (libulog is absent), and it ships no golden vectors.  What
 *   pins this restatement to the real reference is therefore limited to
 *     (a) oracle/_ref: the reference's OWN public header
 *         include/h264/h264_bitstream.h compiled as it lies (its static-inline
 *         bit reader needs no missing dependency) -- pins fetch/EPB/read_bits;
 *     (b) the constant tables, compared entry by entry with the reference's
 *         source text when /root/reference is present
 *         (tests/test_spec_tables.py);
 *     (c) known answers the survey recorded from a real reference run
 *         (BASELINE.md section 2): struct sizes,
 *         SPS/PPS byte strings, synthetic-slice sizes -- tests/golden/.
 *     (d) an independent CAVLC encoder (libh264_amd/csrc/gen) whose recorded
 *         macroblock syntax must equal what this parser reads back
 *         (tests/test_oracle_vs_generator.py), and the dense-CAVLC recipe of
 *         SURVEY Appendix E with its reference-measured size.
 *   Everything else is restated from reading the source and is unpinned.
 *
 * DELIBERATE DEVIATIONS (memory safety, shared with the product): array
 * counts equal to the array size are -EIO where the reference writes one
 * element past the array (src/h264_syntax.h:43,545,566,688 and the PPS
 * slice-group arrays); ue(v) with 32 or more leading zeros is -EIO instead of
 * an undefined shift; the escape test never reads past the buffer end; MBAFF
 * and slice groups return H264_AMD_E_UNSUPPORTED.
 *
 * All oracle symbols carry the orc_ prefix and operate on the public struct
 * types of include/h264/ so that records are byte-comparable with the product.
 */
#ifndef ORC_H
#define ORC_H

#include <h264/h264.h>
#include <h264_amd.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- private macroblock state (ref src/h264_macroblock.h:36-167) --------- */

enum orc_sub_mb_type {
	ORC_SUB_P_8x8,
	ORC_SUB_P_8x4,
	ORC_SUB_P_4x8,
	ORC_SUB_P_4x4,
	ORC_SUB_B_Direct_8x8,
	ORC_SUB_B_8x8,
	ORC_SUB_B_8x4,
	ORC_SUB_B_4x8,
	ORC_SUB_B_4x4,
};

enum orc_pred_mode {
	ORC_PRED_Intra_4x4,
	ORC_PRED_Intra_8x8,
	ORC_PRED_Intra_16x16,
	ORC_PRED_L0,
	ORC_PRED_L1,
	ORC_PRED_Bi,
	ORC_PRED_Direct,
};

/* residual block classes (ref enum Level, src/h264_macroblock.h:62-74) */
enum orc_level_kind {
	ORC_LVL_I16DC,
	ORC_LVL_I16AC,
	ORC_LVL_CB_I16DC,
	ORC_LVL_CB_I16AC,
	ORC_LVL_CR_I16DC,
	ORC_LVL_CR_I16AC,
	ORC_LVL_LUMA4x4,
	ORC_LVL_CB4x4,
	ORC_LVL_CR4x4,
	ORC_LVL_CHROMA_DC,
	ORC_LVL_CHROMA_AC,
};

#define ORC_MB_ADDR_INVALID 0xffffffffu

/* per-MB neighbour context kept for the whole slice
 * (ref struct h264_macroblock_info, src/h264_macroblock.h:94-101) */
struct orc_mb_info {
	uint8_t mb_type; /* reference stores 4 bits */
	uint8_t intra_chroma_pred_mode;
	uint8_t available;
	uint8_t skipped;
	uint8_t field_flag;
	uint8_t nz_coeff[48];
	/* ---- CABAC decoding only (engine extension, orc_cabac_dec.c): what the
	 * context index rules of 9.3.3.1.1 ask of a neighbouring macroblock ---- */
	uint8_t mbt; /* enum h264_mb_type, not truncated */
	uint8_t cbp; /* coded_block_pattern; I_PCM: 0x2f */
	uint8_t t8; /* transform_size_8x8_flag */
	uint8_t cbf_dc; /* coded_block_flag of DC blocks: bit 0 Y, 1 Cb, 2 Cr */
	/* coded_block_flag per block: [0] by luma4x4BlkIdx, [1] [2] Cb / Cr by
	 * cb4x4BlkIdx (ChromaArrayType 3) or chroma4x4BlkIdx; an 8x8 block sets
	 * the four bits of its 4x4 blocks */
	uint16_t cbf[3];
	uint8_t ref_idx[2][4]; /* as transmitted, per 8x8 block */
	int16_t mvd[2][16][2]; /* as transmitted, per luma4x4BlkIdx */
};

/* current macroblock (ref struct h264_macroblock, :105-167) */
struct orc_mb {
	uint32_t mbAddr;
	int mb_field_decoding_flag;
	uint32_t mbAddrA, mbAddrB;
	const struct orc_mb_info *infoA, *infoB;

	uint32_t mb_type;
	uint32_t raw_mb_type;
	uint32_t NumMbPart;
	uint32_t MbPartPredMode[4];
	uint32_t raw_sub_mb_type[4];
	uint32_t sub_mb_type[4];
	uint32_t NumSubMbPart[4];
	uint32_t SubMbPredMode[4];
	int transform_size_8x8_flag;
	int32_t mb_qp_delta;

	uint16_t pcm_sample_luma[256];
	uint16_t pcm_sample_chroma[2][256];

	int8_t intra4x4_pred_mode[16];
	int8_t intra8x8_pred_mode[4];
	uint8_t intra_chroma_pred_mode;
	uint8_t Intra16x16PredMode;

	uint8_t max_ref_idx_0, max_ref_idx_1;
	uint8_t ref_idx_l0[4], ref_idx_l1[4];
	int16_t mvd_l0[4][4][2], mvd_l1[4][4][2];

	uint8_t coded_block_pattern;
	uint8_t CodedBlockPatternLuma;
	uint8_t CodedBlockPatternChroma;

	/* [0]=Y, [1]=Cb, [2]=Cr for the 4:4:4 triple; chroma DC/AC for 4:2:x */
	int16_t I16DC[3][16];
	int16_t I16AC[3][16][15];
	int16_t Lvl4x4[3][16][16];
	int16_t Lvl8x8[3][4][64];
	int16_t ChromaDC[2][16];
	int16_t ChromaAC[2][16][15];
};

struct orc_sei_entry {
	struct h264_sei sei;
};

/* parsing context (ref struct h264_ctx, src/h264_priv.h:67-140) */
struct orc_ctx {
	struct {
		uint32_t type;
		struct h264_nalu_header hdr;
		int unknown;
		int is_first_vcl;
		int is_prev_vcl;
		int is_prev_filler;
	} nalu;
	struct h264_aud aud;
	struct h264_sps *sps;
	struct h264_pps *pps;
	struct h264_sps *sps_table[32];
	struct h264_pps *pps_table[256];
	struct h264_sei *sei_table;
	uint32_t sei_count;
	size_t filler_len;
	struct {
		int type; /* enum h264_slice_type */
		struct h264_slice_header hdr;
		size_t hdr_len;
		struct h264_slice_header saved_hdr;
		struct {
			uint8_t partial;
			uint8_t partialbits;
			const uint8_t *buf;
			size_t len;
		} rawdata;
		struct {
			struct orc_mb_info *info;
			size_t len, maxlen;
		} mb_table;
		uint32_t *group_map; /* ref src/h264_priv.h:109-110 */
		uint32_t group_map_maxlen;
		struct h264_nalu_header prev_slice_nalu_hdr;
		struct h264_slice_header prev_slice_hdr;
	} slice;
	struct orc_mb _mb;
	struct orc_mb *mb;
	struct h264_sps_derived sps_derived;
	struct {
		uint32_t SliceGroupChangeRate;
		int MbaffFrameFlag;
		uint32_t PicHeightInMbs;
		uint32_t PicSizeInMbs;
		uint32_t MaxPicNum, CurrPicNum;
		int32_t SliceQPLuma, QSLuma;
		uint32_t FilterOffsetA, FilterOffsetB;
		uint32_t MapUnitsInSliceGroup0;
	} derived;
	/* bits of RBSP consumed inside the current NAL (diagnostic, for
	 * h264_amd_slice_result.end_bit) */
	const struct h264_bitstream *cur_bs;
};

/* Callbacks: same table as the public API but ctx is the oracle's. */
struct orc_cbs {
	void (*au_end)(struct orc_ctx *, void *);
	void (*nalu_begin)(struct orc_ctx *, uint32_t, const uint8_t *, size_t,
			   const struct h264_nalu_header *, void *);
	void (*nalu_end)(struct orc_ctx *, uint32_t, const uint8_t *, size_t,
			 const struct h264_nalu_header *, void *);
	void (*slice)(struct orc_ctx *, const uint8_t *, size_t,
		      const struct h264_slice_header *, void *);
	void (*slice_data_begin)(struct orc_ctx *,
				 const struct h264_slice_header *, void *);
	void (*slice_data_end)(struct orc_ctx *,
			       const struct h264_slice_header *, uint32_t,
			       void *);
	void (*slice_data_mb)(struct orc_ctx *,
			      const struct h264_slice_header *, uint32_t,
			      uint32_t, void *);
	void (*sps)(struct orc_ctx *, const uint8_t *, size_t,
		    const struct h264_sps *, void *);
	void (*pps)(struct orc_ctx *, const uint8_t *, size_t,
		    const struct h264_pps *, void *);
	void (*aud)(struct orc_ctx *, const uint8_t *, size_t,
		    const struct h264_aud *, void *);
	void (*sei)(struct orc_ctx *, uint32_t, const uint8_t *, size_t, void *);
	void (*sei_typed)(struct orc_ctx *, uint32_t type, const uint8_t *,
			  size_t, const struct h264_sei *, void *);
};

struct orc_reader {
	struct orc_cbs cbs;
	void *userdata;
	int stop;
	struct orc_ctx *ctx;
	uint32_t flags;
};

/* ---- bit level (orc_bits.c; ref include/h264/h264_bitstream.h,
 *      src/h264_bitstream.c) -------------------------------------------- */
extern uint64_t orc_stat_dropped_coeffs; /* test instrumentation */
int orc_find_nalu(const uint8_t *buf, size_t len, size_t *start, size_t *end);
int orc_bs_fetch(struct h264_bitstream *bs);
int orc_bs_read_bits(struct h264_bitstream *bs, uint32_t *v, uint32_t n);
int orc_bs_read_bits_i(struct h264_bitstream *bs, int32_t *v, uint32_t n);
int orc_bs_read_ue(struct h264_bitstream *bs, uint32_t *v);
int orc_bs_read_se(struct h264_bitstream *bs, int32_t *v);
int orc_bs_read_te(struct h264_bitstream *bs, uint32_t *v, uint32_t m);
int orc_bs_read_ff(struct h264_bitstream *bs, uint32_t *v);
int orc_bs_more_rbsp_data(const struct h264_bitstream *bs);
int orc_bs_next_bits(const struct h264_bitstream *bs, uint32_t *v, uint32_t n);
int orc_bs_read_trailing(struct h264_bitstream *bs);
int orc_bs_write_bits(struct h264_bitstream *bs, uint64_t v, uint32_t n);
int orc_bs_write_ue(struct h264_bitstream *bs, uint32_t v);
int orc_bs_write_se(struct h264_bitstream *bs, int32_t v);
int orc_bs_write_te(struct h264_bitstream *bs, uint32_t v, uint32_t m);
int orc_bs_write_ff(struct h264_bitstream *bs, uint32_t v);
int orc_bs_write_trailing(struct h264_bitstream *bs);
int orc_bs_write_raw_bytes(struct h264_bitstream *bs, const uint8_t *buf,
			   size_t len);
/* number of RBSP bits consumed so far by a reader cursor (ours, diagnostic) */
uint64_t orc_bs_rbsp_bits_consumed(const struct h264_bitstream *bs);

/* ---- context (orc_ctx.c; ref src/h264_ctx.c, src/h264.c) -------------- */
int orc_ctx_new(struct orc_ctx **ret);
int orc_ctx_destroy(struct orc_ctx *ctx);
int orc_ctx_clear(struct orc_ctx *ctx);
int orc_ctx_clear_nalu(struct orc_ctx *ctx);
int orc_ctx_clear_slice(struct orc_ctx *ctx);
int orc_ctx_clear_sei_table(struct orc_ctx *ctx);
int orc_ctx_set_nalu_header(struct orc_ctx *ctx,
			    const struct h264_nalu_header *nh);
int orc_ctx_set_sps(struct orc_ctx *ctx, const struct h264_sps *sps);
int orc_ctx_set_pps(struct orc_ctx *ctx, const struct h264_pps *pps);
int orc_ctx_set_active_sps(struct orc_ctx *ctx, uint32_t id);
int orc_ctx_set_active_pps(struct orc_ctx *ctx, uint32_t id);
int orc_ctx_set_slice_header(struct orc_ctx *ctx,
			     const struct h264_slice_header *sh);
int orc_ctx_add_sei_internal(struct orc_ctx *ctx, struct h264_sei **ret);
int orc_ctx_add_sei(struct orc_ctx *ctx, const struct h264_sei *sei);
int orc_get_sps_derived(const struct h264_sps *sps,
			struct h264_sps_derived *d);
int orc_get_info_from_ps(const struct h264_sps *sps,
			 const struct h264_sps_derived *d,
			 struct h264_info *info);
int orc_sei_update_internal_buf(struct h264_sei *sei);
uint32_t orc_intlog2(uint32_t x);
/* orc_fmo.c (ref src/h264_fmo.c) */
int orc_gen_slice_group_map(struct orc_ctx *ctx);
void orc_clear_slice_group_map(struct orc_ctx *ctx);
uint32_t orc_next_mb_addr(struct orc_ctx *ctx, uint32_t mbAddr);

/* ---- syntax (orc_syntax.c; ref src/h264_syntax.h) --------------------- */
int orc_read_nalu_header(struct h264_bitstream *bs, struct h264_nalu_header *nh);
int orc_read_nalu(struct h264_bitstream *bs, struct orc_reader *rd);
int orc_write_nalu(struct h264_bitstream *bs, struct orc_ctx *ctx);
int orc_write_one_sei(struct h264_bitstream *bs, struct orc_ctx *ctx,
		      const struct h264_sei *sei);
int orc_parse_sps(const uint8_t *buf, size_t len, struct h264_sps *sps);
int orc_parse_pps(const uint8_t *buf, size_t len, const struct h264_sps *sps,
		  struct h264_pps *pps);

/* AVCC <-> Annex-B byte stream, in place (orc_ctx.c; ref src/h264.c:184-272) */
int orc_byte_stream_to_avcc(uint8_t *data, size_t len);
int orc_avcc_to_byte_stream(uint8_t *data, size_t len);

/* ---- CAVLC slice data (orc_cavlc.c; ref src/h264_syntax_slice_data.h,
 *      src/h264_slice_data.c, src/h264_macroblock.c) ------------------ */
int orc_read_slice_data(struct h264_bitstream *bs, struct orc_reader *rd);
uint32_t orc_mb_off(const struct orc_ctx *ctx, uint32_t mbAddr);
int orc_apply_mb_type(struct orc_ctx *ctx, struct orc_mb *mb, uint32_t type);
int orc_apply_sub_mb_type(struct orc_ctx *ctx, struct orc_mb *mb, uint32_t p,
			  uint32_t type);
void orc_neigh_loc(const struct orc_ctx *ctx, const struct orc_mb *mb,
		   uint32_t maxW, uint32_t maxH, int32_t xN, int32_t yN,
		   uint32_t *addrN, uint32_t *xW, uint32_t *yW);
void orc_neigh_luma4x4(const struct orc_ctx *ctx, const struct orc_mb *mb,
		       uint32_t idx, uint32_t *addrA, uint32_t *idxA,
		       uint32_t *addrB, uint32_t *idxB);
void orc_neigh_chroma4x4(const struct orc_ctx *ctx, const struct orc_mb *mb,
			 uint32_t idx, uint32_t *addrA, uint32_t *idxA,
			 uint32_t *addrB, uint32_t *idxB);
int orc_new_macroblock(struct orc_ctx *ctx, uint32_t mbAddr, int skipped,
		       int field_flag);
void orc_clear_macroblock_table(struct orc_ctx *ctx);
/* serialise ctx->mb into the engine's T2 macroblock record */
void orc_mb_to_record(const struct orc_ctx *ctx, uint8_t *rec, uint32_t stride);

/* ---- CABAC encoder + synthetic slices (orc_cabac.c; ref src/h264_bac.c,
 *      src/h264_cabac.c, src/h264_cabac_ctx_tables.c, src/h264_writer.c) */
void orc_cabac_init_states(int slice_type, uint32_t cabac_init_idc,
			   int32_t SliceQPLuma, uint8_t states[1024][2]);
/* engine extension (no reference behaviour, spec-only, parity unpinned beyond
 * what the reference's writers emit): CABAC slice data, orc_cabac_dec.c */
int orc_cabac_slice_data(struct h264_bitstream *bs, struct orc_reader *rd);
void orc_cabac_init_states_floor(int slice_type, uint32_t cabac_init_idc,
				 int32_t SliceQPLuma, int32_t qp_floor,
				 uint8_t states[1024][2]);
extern uint64_t orc_stat_cabac_bins; /* bins decoded (bench statistics) */

extern int orc_quirk_x264_444_cbf; /* decode 4:4:4 streams of x264 < build 151 (test only) */
extern uint64_t orc_stat_cabac_hist[1026]; /* by ctxIdx; 1024 bypass, 1025 terminate */
int orc_write_grey_i_slice(struct h264_bitstream *bs, struct orc_ctx *ctx,
			   uint32_t mb_count);
int orc_write_skipped_p_slice(struct h264_bitstream *bs, struct orc_ctx *ctx,
			      uint32_t mb_count);
int orc_rewrite_slice_header(struct h264_bitstream *bs, struct orc_ctx *ctx,
			     const struct h264_slice_header *sh);

/* ---- reader object + recorder (orc_api.c; ref src/h264_reader.c) ------ */
int orc_reader_new(const struct orc_cbs *cbs, void *userdata,
		   struct orc_reader **ret);
int orc_reader_destroy(struct orc_reader *rd);
int orc_reader_parse(struct orc_reader *rd, uint32_t flags, const uint8_t *buf,
		     size_t len, size_t *off);
int orc_reader_parse_nalu(struct orc_reader *rd, uint32_t flags,
			  const uint8_t *buf, size_t len);

#ifdef __cplusplus
}
#endif
#endif /* ORC_H */
