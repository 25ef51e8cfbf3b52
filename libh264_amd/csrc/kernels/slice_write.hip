/**
 * slice_write.hip -- device-side serialisation of slice NAL units (placeholder
 * header: the kernels are added in the next step).
 */
#ifndef H264_AMD_SLICE_WRITE_HIP
#define H264_AMD_SLICE_WRITE_HIP

#include <stdint.h>

/* CABAC constants: Table 9-44 / 9-45 and the (m, n) pairs of Tables 9-12..33,
 * [ctxIdx][I, P idc0, idc1, idc2][m, n] */
struct cabac_tables {
	uint8_t range_lps[64][4];
	uint8_t next_lps[64];
	uint8_t next_mps[64];
	int8_t mn[1024][4][2];
};

static void cabac_build_tables(cabac_tables *t)
{
	memset(t, 0, sizeof(*t));
#define H264_LPS(i, a, b, c, d)                                                \
	t->range_lps[i][0] = a;                                                \
	t->range_lps[i][1] = b;                                                \
	t->range_lps[i][2] = c;                                                \
	t->range_lps[i][3] = d;
#define H264_TRANS(i, l, m)                                                    \
	t->next_lps[i] = l;                                                    \
	t->next_mps[i] = m;
#define H264_MN(i, mi, ni, m0, n0, m1, n1, m2, n2)                             \
	t->mn[i][0][0] = mi;                                                   \
	t->mn[i][0][1] = ni;                                                   \
	t->mn[i][1][0] = m0;                                                   \
	t->mn[i][1][1] = n0;                                                   \
	t->mn[i][2][0] = m1;                                                   \
	t->mn[i][2][1] = n1;                                                   \
	t->mn[i][3][0] = m2;                                                   \
	t->mn[i][3][1] = n2;
#include "../common/h264_cabac_tables.inc"
#undef H264_LPS
#undef H264_TRANS
#undef H264_MN
}

#endif /* H264_AMD_SLICE_WRITE_HIP */
