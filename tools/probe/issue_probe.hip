// Issue-port calibration for the slice decode kernel: what do scalar, vector,
// cross-lane and LDS instructions cost on gfx950 when 1, 2 or 4 waves share a
// SIMD?  Each test is a loop of ITER iterations around UNROLL copies of a short
// instruction pattern (inline asm, .rept); the grid puts `w` waves on every SIMD
// (256 CUs x 4 SIMDs).  Output: cycles per pattern instance per SIMD (assuming
// 2.4 GHz) = how often one SIMD can start the pattern, and ns per instance per
// wave (the latency view when w = 1).
// Build: hipcc --offload-arch=gfx950 -O3 issue_probe.hip -o issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
	fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define ITER 2000
#define UNROLL 64

#define STR2(x) #x
#define STR(x) STR2(x)

// body: asm text using s10..s19 / v10..v19 as scratch (declared clobbered)
#define PROBE(name, body) PROBEU(name, body, UNROLL)
#define PROBEU(name, body, UNR)                                                \
	__global__ void __launch_bounds__(256) name(uint32_t *out, uint32_t seed) \
	{                                                                      \
		__shared__ uint32_t lds[1024];                                 \
		lds[threadIdx.x] = threadIdx.x * 4u;                           \
		lds[threadIdx.x + 256] = seed;                                 \
		__syncthreads();                                               \
		uint32_t acc = seed + threadIdx.x;                             \
		uint32_t addr = (threadIdx.x & 63u) * 4u;                      \
		for (int it = 0; it < ITER; it++) {                            \
			asm volatile(                                          \
				"v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n"      \
				"v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n"      \
				"v_mov_b32 v14, %1\n v_mov_b32 v15, 0\n"       \
				"v_readfirstlane_b32 s10, %0\n"                \
				"s_mov_b32 s11, s10\n s_mov_b32 s12, s10\n"    \
				"s_mov_b32 s13, s10\n s_mov_b32 s14, 3\n"      \
				"s_mov_b32 s15, 0\n"                           \
				".rept " STR(UNR) "\n" body "\n.endr\n"        \
				"s_waitcnt lgkmcnt(0)\n"                       \
				"v_add_u32 %0, v10, v11\n v_add_u32 %0, %0, v12\n" \
				"v_add_u32 %0, %0, v13\n v_add_u32 %0, %0, s10\n"  \
				"v_add_u32 %0, %0, s11\n v_add_u32 %0, %0, v15\n"  \
				: "+v"(acc)                                    \
				: "v"(addr)                                    \
				: "v10", "v11", "v12", "v13", "v14", "v15",    \
				  "v16", "v17", "v18", "v19", "s10", "s11",    \
				  "s12", "s13", "s14", "s15", "s16", "s17",    \
				  "s18", "s19", "m0", "vcc", "scc",            \
				  "memory");                                   \
		}                                                              \
		if (acc == 0x12345678u)                                        \
			out[0] = acc;                                          \
	}

// ---- scalar ----
PROBE(k_salu_dep, "s_add_u32 s10, s10, 1")
PROBE(k_salu_ind4, "s_add_u32 s10, s10, 1\n s_add_u32 s11, s11, 1\n s_add_u32 s12, s12, 1\n s_add_u32 s13, s13, 1")
PROBE(k_salu_shift64, "s_lshl_b64 s[10:11], s[10:11], 1")
// ---- vector ----
PROBE(k_valu_dep, "v_add_u32 v10, v10, 1")
PROBE(k_valu_ind4, "v_add_u32 v10, v10, 1\n v_add_u32 v11, v11, 1\n v_add_u32 v12, v12, 1\n v_add_u32 v13, v13, 1")
PROBE(k_valu_shift64, "v_lshlrev_b64 v[10:11], 1, v[10:11]")
PROBE(k_valu_shift64_ind, "v_lshlrev_b64 v[10:11], 1, v[10:11]\n v_lshlrev_b64 v[12:13], 1, v[12:13]")
PROBE(k_valu_ffbh, "v_ffbh_u32 v10, v10")
PROBE(k_valu_cndmask, "v_cmp_eq_u32 vcc, v10, v11\n v_cndmask_b32 v12, v12, v13, vcc")
PROBE(k_valu_bfe, "v_bfe_u32 v10, v10, v11, 5")
// ---- mixed: scalar and vector side by side (different registers) ----
PROBE(k_mix_1s1v, "s_add_u32 s10, s10, 1\n v_add_u32 v10, v10, 1")
PROBE(k_mix_1s2v, "s_add_u32 s10, s10, 1\n v_add_u32 v10, v10, 1\n v_add_u32 v11, v11, 1")
PROBE(k_mix_1s3v, "s_add_u32 s10, s10, 1\n v_add_u32 v10, v10, 1\n v_add_u32 v11, v11, 1\n v_add_u32 v12, v12, 1")
PROBE(k_mix_2s1v, "s_add_u32 s10, s10, 1\n s_add_u32 s11, s11, 1\n v_add_u32 v10, v10, 1")
// ---- cross domain ----
// the decode kernel's peek: two v_readlane with a scalar lane index, 64-bit shift
PROBE(k_peek, "s_lshr_b32 s16, s15, 5\n s_and_b32 s17, s15, 31\n v_readlane_b32 s13, v10, s16\n v_readlane_b32 s12, v11, s16\n s_lshl_b64 s[12:13], s[12:13], s17\n s_add_u32 s15, s15, s14")
// scalar -> vector -> scalar round trip
PROBE(k_s2v2s, "v_add_u32 v10, s10, 1\n s_nop 0\n v_readfirstlane_b32 s10, v10")
// vector compare feeding a branch (never taken)
PROBE(k_vcmp_branch, "v_cmp_eq_u32 vcc, 77, v15\n s_cbranch_vccnz 1f\n1:")
// scalar compare feeding a branch (never taken)
PROBE(k_scmp_branch, "s_cmp_eq_u32 s15, 77\n s_cbranch_scc1 1f\n1:")
// taken branch
PROBE(k_branch_taken, "s_cmp_eq_u32 s15, 0\n s_cbranch_scc1 1f\n s_nop 0\n1:")
PROBE(k_sbranch, "s_branch 1f\n s_nop 0\n1:")
// ballot -> ff1 -> readlane (the table match tail)
PROBE(k_match_tail, "v_cmp_eq_u32 s[16:17], v14, v14\n s_ff1_i32_b64 s16, s[16:17]\n s_nop 0\n v_readlane_b32 s10, v10, s16")
// LDS read with dependent use
PROBE(k_lds_dep, "ds_read_b32 v14, v14\n s_waitcnt lgkmcnt(0)")
PROBE(k_lds_b64_dep, "ds_read_b64 v[16:17], v14\n s_waitcnt lgkmcnt(0)\n v_and_b32 v14, 0xfc, v16")
// bpermute
PROBE(k_bpermute_dep, "ds_bpermute_b32 v14, v14, v14\n s_waitcnt lgkmcnt(0)")
// writelane with scalar index
PROBE(k_writelane, "s_mov_b32 m0, s14\n v_writelane_b32 v10, s10, m0")
// DPP add
PROBE(k_dpp, "v_add_u32_dpp v10, v10, v10 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1")
PROBE(k_nop, "s_nop 0")


// ---- single instruction costs (dependent chains unless noted) ----
PROBE(k_v_and, "v_and_b32 v10, v10, v11")
PROBE(k_v_or_s, "v_or_b32 v10, s10, v10")
PROBE(k_v_lshl, "v_lshlrev_b32 v10, 1, v10")
PROBE(k_v_lshr_v, "v_lshrrev_b32 v10, v15, v10")
PROBE(k_v_sub, "v_sub_u32 v10, v10, v11")
PROBE(k_v_add3, "v_add3_u32 v10, v10, v11, 1")
PROBE(k_v_lshl_add, "v_lshl_add_u32 v10, v10, 1, v11")
PROBE(k_v_cndmask_s, "v_cndmask_b32_e64 v10, v10, v11, s[16:17]")
PROBE(k_v_cndmask_vcc, "v_cndmask_b32 v10, v10, v11, vcc")
PROBE(k_v_cmp_vcc, "v_cmp_eq_u32 vcc, v10, v11")
PROBE(k_v_cmp_sgpr, "v_cmp_eq_u32 s[16:17], v10, v11")
PROBE(k_v_mov, "v_mov_b32 v10, v11")
PROBE(k_v_mov_s, "v_mov_b32 v10, s10")
PROBE(k_v_readlane_c, "v_readlane_b32 s10, v10, 5")
PROBE(k_v_readlane_s, "v_readlane_b32 s10, v10, s14")
PROBE(k_v_readfirstlane, "v_readfirstlane_b32 s10, v10")
PROBE(k_v_min, "v_min_u32 v10, v10, v11")
PROBE(k_v_bitop3, "v_bitop3_b32 v10, v10, v11, v12 bitop3:0x48")
PROBE(k_v_mul_lo, "v_mul_lo_u32 v10, v10, v11")
PROBE(k_v_mul_u24, "v_mul_u32_u24 v10, v10, v11")
PROBE(k_v_mad_u24, "v_mad_u32_u24 v10, v10, v11, v12")
PROBE(k_v_alignbit, "v_alignbit_b32 v10, v10, v11, v15")
PROBE(k_v_xad, "v_xad_u32 v10, v10, v11, v12")
PROBE(k_v_addc, "v_addc_co_u32 v10, vcc, v10, v11, vcc")
PROBE(k_v_and_ind4, "v_and_b32 v10, v10, v14\n v_and_b32 v11, v11, v14\n v_and_b32 v12, v12, v14\n v_and_b32 v13, v13, v14")
PROBE(k_v_bfe_ind4, "v_bfe_u32 v10, v10, v15, 5\n v_bfe_u32 v11, v11, v15, 5\n v_bfe_u32 v12, v12, v15, 5\n v_bfe_u32 v13, v13, v15, 5")
PROBE(k_s_bfe, "s_bfe_u32 s10, s10, 0x100000")
PROBE(k_s_ff1_64, "s_ff1_i32_b64 s10, s[10:11]")
PROBE(k_s_cselect, "s_cmp_eq_u32 s10, 0\n s_cselect_b32 s10, s11, s12")
PROBE(k_s_bitset0, "s_bitset0_b64 s[10:11], s14")
PROBE(k_s_movrels, "s_mov_b32 m0, s14\n s_movrels_b32 s10, s10")
// lone-wave ILP: 2 scalar chains + 2 vector chains interleaved
PROBE(k_ilp_2s2v, "s_add_u32 s10, s10, 1\n v_add_u32 v10, v10, 1\n s_add_u32 s11, s11, 1\n v_add_u32 v11, v11, 1")
// uniform-VGPR level step sketch: shift64, ffbh, add3, sub, lshr, add, lshl_add, cndmask, add, cmp, addc, min
PROBE(k_level_valu, "v_lshlrev_b64 v[16:17], v15, v[10:11]\n v_ffbh_u32 v12, v17\n v_add3_u32 v13, v12, v14, 1\n v_sub_u32 v16, 32, v13\n v_lshrrev_b32 v16, v16, v17\n v_add_u32 v12, -1, v12\n v_lshl_add_u32 v16, v12, v14, v16\n v_cndmask_b32_e64 v10, v10, v16, s[12:13]\n v_add_u32 v15, v15, v13\n v_cmp_lt_u32 vcc, 2, v12\n v_addc_co_u32 v14, vcc, 0, v14, vcc\n v_min_u32 v14, 6, v14\n v_and_b32 v15, 31, v15\n s_lshl_b64 s[12:13], s[12:13], 1\n s_cmp_lg_u64 s[12:13], 0")

// ---- what bounds a CABAC decision: scalar instructions alone and with the
// vector-side work of the decision beside them
PROBEU(k_dec_s12, "s_add_u32 s10, s10, 1\n s_add_u32 s11, s11, 1\n s_add_u32 s10, s10, 1\n s_add_u32 s11, s11, 1\n s_add_u32 s10, s10, 1\n s_add_u32 s11, s11, 1\n s_add_u32 s10, s10, 1\n s_add_u32 s11, s11, 1\n s_add_u32 s10, s10, 1\n s_add_u32 s11, s11, 1\n s_add_u32 s10, s10, 1\n s_add_u32 s11, s11, 1", 8)
PROBEU(k_dec_s12_rl6, "s_add_u32 s10, s10, 1\n v_readlane_b32 s12, v10, s14\n s_add_u32 s11, s11, 1\n v_readlane_b32 s13, v11, s14\n s_add_u32 s10, s10, 1\n v_readlane_b32 s16, v12, s14\n s_add_u32 s11, s11, 1\n v_readlane_b32 s12, v10, s14\n s_add_u32 s10, s10, 1\n v_readlane_b32 s13, v11, s14\n s_add_u32 s11, s11, 1\n v_readlane_b32 s16, v12, s14\n s_add_u32 s10, s10, 1\n s_add_u32 s11, s11, 1\n s_add_u32 s10, s10, 1\n s_add_u32 s11, s11, 1\n s_add_u32 s10, s10, 1\n s_add_u32 s11, s11, 1", 8)
PROBEU(k_dec_s12_vp6, "s_add_u32 s10, s10, 1\n v_and_b32 v10, v10, v11\n s_add_u32 s11, s11, 1\n v_and_b32 v12, v12, v13\n s_add_u32 s10, s10, 1\n v_and_b32 v10, v10, v11\n s_add_u32 s11, s11, 1\n v_and_b32 v12, v12, v13\n s_add_u32 s10, s10, 1\n v_and_b32 v10, v10, v11\n s_add_u32 s11, s11, 1\n v_and_b32 v12, v12, v13\n s_add_u32 s10, s10, 1\n s_add_u32 s11, s11, 1\n s_add_u32 s10, s10, 1\n s_add_u32 s11, s11, 1\n s_add_u32 s10, s10, 1\n s_add_u32 s11, s11, 1", 8)
PROBEU(k_dec_s12_vs6, "s_add_u32 s10, s10, 1\n v_or_b32 v10, s15, v10\n s_add_u32 s11, s11, 1\n v_or_b32 v12, s15, v12\n s_add_u32 s10, s10, 1\n v_or_b32 v10, s15, v10\n s_add_u32 s11, s11, 1\n v_or_b32 v12, s15, v12\n s_add_u32 s10, s10, 1\n v_or_b32 v10, s15, v10\n s_add_u32 s11, s11, 1\n v_or_b32 v12, s15, v12\n s_add_u32 s10, s10, 1\n s_add_u32 s11, s11, 1\n s_add_u32 s10, s10, 1\n s_add_u32 s11, s11, 1\n s_add_u32 s10, s10, 1\n s_add_u32 s11, s11, 1", 8)
PROBEU(k_dec_s9, "s_add_u32 s10, s10, 1\n s_add_u32 s11, s11, 1\n s_add_u32 s10, s10, 1\n s_add_u32 s11, s11, 1\n s_add_u32 s10, s10, 1\n s_add_u32 s11, s11, 1\n s_add_u32 s10, s10, 1\n s_add_u32 s11, s11, 1\n s_add_u32 s10, s10, 1", 8)
PROBEU(k_dec_full, "v_readlane_b32 s12, v10, s14\n v_lshrrev_b32 v16, 29, s10\n v_perm_b32 v16, v11, 0, v16\n v_lshlrev_b32 v16, 23, v16\n v_sub_u32 v17, s10, v16\n v_lshrrev_b32 v13, 6, v10\n v_readlane_b32 s13, v16, s14\n v_readlane_b32 s16, v17, s14\n v_readlane_b32 s17, v11, s14\n v_readlane_b32 s12, v12, s14\n v_readlane_b32 s13, v13, s14\n s_sub_u32 s11, s11, s16\n s_cselect_b32 s11, s11, s16\n s_cselect_b32 s10, s16, s13\n s_cselect_b32 s17, s17, s12\n s_addc_u32 s12, s13, 0\n s_xor_b32 s17, s17, s12\n s_mov_b32 m0, s14\n v_writelane_b32 v10, s17, m0\n s_flbit_i32_b32 s16, s10\n s_lshl_b32 s10, s10, s16\n s_lshl_b32 s11, s11, s16", 8)

PROBEU(k_dec_valu, "s_mov_b32 m0, s14\n v_readlane_b32 s10, v10, m0\n v_lshrrev_b32 v16, 29, v15\n v_perm_b32 v16, v11, 0, v16\n v_lshlrev_b32 v16, 23, v16\n v_sub_u32 v17, v15, v16\n v_readlane_b32 s11, v17, s10\n v_cmp_gt_u32_e64 s[16:17], s11, v14\n v_subrev_u32 v18, s11, v14\n v_cndmask_b32_e64 v14, v18, v14, s[16:17]\n v_cndmask_b32_e64 v19, v16, v17, s[16:17]\n v_cndmask_b32_e64 v18, v12, v13, s[16:17]\n v_readlane_b32 s12, v19, s10\n v_readlane_b32 s13, v18, s10\n s_xor_b32 s13, s13, s10\n v_writelane_b32 v10, s13, m0\n v_ffbh_u32 v19, s12\n v_lshlrev_b32 v15, v19, s12\n v_lshlrev_b32 v14, v19, v14\n s_bfe_i64 s[18:19], s[10:11], 0x10006\n s_xor_b64 vcc, s[16:17], s[18:19]\n v_and_b32 v18, 0x1ffff, v14\n v_cmp_eq_u32 vcc, 0, v18", 8)

PROBEU(k_dec_v5, "v_readlane_b32 s12, v10, s14\n v_lshrrev_b32 v16, 29, s10\n v_perm_b32 v16, v11, 0, v16\n v_lshlrev_b32 v16, 23, v16\n v_sub_u32 v17, s10, v16\n v_lshrrev_b32 v13, 6, v10\n v_readlane_b32 s13, v16, s14\n v_readlane_b32 s16, v17, s14\n v_readlane_b32 s13, v13, s14\n s_sub_u32 s11, s11, s16\n s_cselect_b32 s11, s11, s16\n s_cselect_b32 s10, s16, s13\n s_cselect_b64 s[18:19], -1, 0\n v_cndmask_b32_e64 v18, v11, v12, s[18:19]\n s_addc_u32 s12, s13, 0\n v_readlane_b32 s17, v18, s14\n s_xor_b32 s17, s17, s12\n s_mov_b32 m0, s14\n v_writelane_b32 v10, s17, m0\n s_flbit_i32_b32 s16, s10\n s_lshl_b32 s10, s10, s16\n s_lshl_b32 s11, s11, s16", 8)
PROBEU(k_dec_v2, "s_mov_b32 m0, s14\n v_readlane_b32 s12, v10, m0\n s_bfe_u32 s16, s10, 0x3001d\n v_perm_b32 v16, v11, 0, s16\n v_lshlrev_b32 v16, 23, v16\n s_lshr_b32 s17, s12, 6\n s_nop 0\n v_readlane_b32 s13, v16, s14\n v_readlane_b32 s16, v11, s14\n v_readlane_b32 s18, v12, s14\n s_sub_u32 s19, s10, s13\n s_sub_u32 s15, s11, s19\n s_cselect_b32 s11, s11, s15\n s_cselect_b32 s10, s19, s13\n s_cselect_b32 s16, s18, s16\n s_addc_u32 s17, s17, 0\n s_xor_b32 s16, s16, s12\n v_writelane_b32 v10, s16, m0\n s_flbit_i32_b32 s15, s10\n s_lshl_b32 s10, s10, s15\n s_lshl_b32 s11, s11, s15", 8)

// ---- constants against registers as operands of the same operation
PROBE(k_c_lshl_v, "v_lshlrev_b32 v10, v15, v10")
PROBE(k_c_lshr_c, "v_lshrrev_b32 v10, 3, v10")
PROBE(k_c_add_v, "v_add_u32 v10, v10, v11")
PROBE(k_c_and_c, "v_and_b32 v10, 15, v10")
PROBE(k_c_and_lit, "v_and_b32 v10, 0x1234, v10")
PROBE(k_c_or_v, "v_or_b32 v10, v10, v11")
PROBE(k_c_xor_v, "v_xor_b32 v10, v10, v11")
PROBE(k_c_xor_c, "v_xor_b32 v10, 1, v10")
PROBE(k_c_sub_c, "v_subrev_u32 v10, 1, v10")
PROBE(k_c_cmp_c_cnd, "v_cmp_eq_u32 vcc, 5, v10\n v_cndmask_b32 v12, v12, v13, vcc")
PROBE(k_c_max_v, "v_max_u32 v10, v10, v11")
PROBE(k_c_lshl_or, "v_lshl_or_b32 v10, v10, v15, v11")
PROBE(k_c_and_or, "v_and_or_b32 v10, v10, v11, v12")
PROBE(k_c_mbcnt, "v_mbcnt_lo_u32_b32 v10, v11, v10")
PROBE(k_c_ashr, "v_ashrrev_i32 v10, v15, v10")
PROBE(k_c_perm, "v_perm_b32 v10, v10, v11, v12")
PROBE(k_c_cndmask_after_scmp, "s_cmp_eq_u32 s10, 0\n s_cselect_b64 s[16:17], -1, 0\n v_cndmask_b32_e64 v10, v10, v11, s[16:17]")

struct T { const char *name; void (*fn)(uint32_t *, uint32_t); int insts; int unroll = UNROLL; };

int main(int argc, char **argv)
{
	uint32_t *d;
	CHECK(hipMalloc(&d, 4096));
	hipEvent_t a, b;
	CHECK(hipEventCreate(&a));
	CHECK(hipEventCreate(&b));
	T tests[] = {
		{"salu_dep", k_salu_dep, 1}, {"salu_ind4", k_salu_ind4, 4},
		{"salu_shift64_dep", k_salu_shift64, 1},
		{"valu_dep", k_valu_dep, 1}, {"valu_ind4", k_valu_ind4, 4},
		{"valu_shift64_dep", k_valu_shift64, 1},
		{"valu_shift64_ind2", k_valu_shift64_ind, 2},
		{"valu_ffbh_dep", k_valu_ffbh, 1},
		{"valu_cmp_cndmask", k_valu_cndmask, 2},
		{"valu_bfe_dep", k_valu_bfe, 1},
		{"mix_1s1v", k_mix_1s1v, 2}, {"mix_1s2v", k_mix_1s2v, 3},
		{"mix_1s3v", k_mix_1s3v, 4}, {"mix_2s1v", k_mix_2s1v, 3},
		{"peek(4s+2readlane)", k_peek, 6}, {"s2v2s", k_s2v2s, 3},
		{"vcmp_branch_nt", k_vcmp_branch, 2},
		{"scmp_branch_nt", k_scmp_branch, 2},
		{"scmp_branch_taken", k_branch_taken, 2},
		{"s_branch", k_sbranch, 1},
		{"match_tail", k_match_tail, 4}, {"lds_b32_dep", k_lds_dep, 1},
		{"lds_b64_dep", k_lds_b64_dep, 2},
		{"bpermute_dep", k_bpermute_dep, 1},
		{"m0+writelane", k_writelane, 2}, {"dpp_add+nop", k_dpp, 2},
		{"s_nop", k_nop, 1},

		{"v_and", k_v_and, 1}, {"v_or sgpr", k_v_or_s, 1}, {"v_lshl c", k_v_lshl, 1},
		{"v_lshr v", k_v_lshr_v, 1}, {"v_sub", k_v_sub, 1}, {"v_add3", k_v_add3, 1},
		{"v_lshl_add", k_v_lshl_add, 1}, {"v_cndmask sgpr", k_v_cndmask_s, 1},
		{"v_cndmask vcc", k_v_cndmask_vcc, 1}, {"v_cmp->vcc", k_v_cmp_vcc, 1},
		{"v_cmp->sgpr", k_v_cmp_sgpr, 1}, {"v_mov", k_v_mov, 1}, {"v_mov sgpr", k_v_mov_s, 1},
		{"v_readlane const", k_v_readlane_c, 1}, {"v_readlane sgpr", k_v_readlane_s, 1},
		{"v_readfirstlane", k_v_readfirstlane, 1}, {"v_min", k_v_min, 1},
		{"v_bitop3", k_v_bitop3, 1}, {"v_mul_lo_u32", k_v_mul_lo, 1},
		{"v_mul_u32_u24", k_v_mul_u24, 1}, {"v_mad_u32_u24", k_v_mad_u24, 1},
		{"v_alignbit", k_v_alignbit, 1}, {"v_xad", k_v_xad, 1}, {"v_addc", k_v_addc, 1},
		{"v_and ind4", k_v_and_ind4, 4}, {"v_bfe ind4", k_v_bfe_ind4, 4},
		{"s_bfe", k_s_bfe, 1}, {"s_ff1_b64", k_s_ff1_64, 1}, {"s_cmp+cselect", k_s_cselect, 2},
		{"s_bitset0_b64", k_s_bitset0, 1}, {"m0+s_movrels", k_s_movrels, 2},
		{"ilp_2s2v", k_ilp_2s2v, 4}, {"level_valu(13v+2s)", k_level_valu, 15},
		{"opnd v_lshl vgpr", k_c_lshl_v, 1}, {"opnd v_lshr const", k_c_lshr_c, 1},
		{"opnd v_add vgpr", k_c_add_v, 1}, {"opnd v_and const", k_c_and_c, 1},
		{"opnd v_and literal", k_c_and_lit, 1}, {"opnd v_or vgpr", k_c_or_v, 1},
		{"opnd v_xor vgpr", k_c_xor_v, 1}, {"opnd v_xor const", k_c_xor_c, 1},
		{"opnd v_subrev const", k_c_sub_c, 1},
		{"opnd v_cmp const + cndmask", k_c_cmp_c_cnd, 2},
		{"opnd v_max vgpr", k_c_max_v, 1}, {"opnd v_lshl_or", k_c_lshl_or, 1},
		{"opnd v_and_or", k_c_and_or, 1}, {"opnd v_mbcnt", k_c_mbcnt, 1},
		{"opnd v_ashr vgpr", k_c_ashr, 1}, {"opnd v_perm vgpr", k_c_perm, 1},
		{"opnd s_cmp+cselect64+cndmask", k_c_cndmask_after_scmp, 3},
		{"12 salu", k_dec_s12, 12, 8}, {"12 salu + 6 readlane", k_dec_s12_rl6, 18, 8},
		{"12 salu + 6 valu vgpr", k_dec_s12_vp6, 18, 8},
		{"12 salu + 6 valu sgpr", k_dec_s12_vs6, 18, 8}, {"9 salu", k_dec_s9, 9, 8},
		{"decision (10s+12v)", k_dec_full, 22, 8},
		{"decision on the vector side", k_dec_valu, 23, 8},
		{"decision, state change picked in the lanes", k_dec_v5, 22, 8},
		{"decision, 13 scalar + 4 readlane", k_dec_v2, 21, 8},
	};
	const double ghz = 2.4;
	printf("%-22s %5s | cycles per pattern per SIMD at w waves/SIMD (cycles per pattern per wave)\n",
	       "pattern", "insts");
	for (auto &t : tests) {
		if (argc > 1 && strstr(t.name, argv[1]) == NULL)
			continue;
		printf("%-22s %5d |", t.name, t.insts);
		for (int w : {1, 2, 4}) {
			int blocks = 256 * w; // 256-thread blocks: one wave per SIMD each
			hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, d, 1u);
			CHECK(hipDeviceSynchronize());
			std::vector<float> ms;
			for (int r = 0; r < 3; r++) {
				CHECK(hipEventRecord(a));
				hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, d, 1u);
				CHECK(hipEventRecord(b));
				CHECK(hipEventSynchronize(b));
				float m;
				CHECK(hipEventElapsedTime(&m, a, b));
				ms.push_back(m);
			}
			std::sort(ms.begin(), ms.end());
			double cyc = ms[1] * 1e-3 * ghz * 1e9; // per wave wall cycles
			double per_wave = cyc / ((double)ITER * t.unroll);
			printf("  w=%d %6.2f (%6.2f)", w, per_wave / w, per_wave);
		}
		printf("\n");
		fflush(stdout);
	}
	return 0;
}
