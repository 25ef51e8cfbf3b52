#!/usr/bin/env python3
"""Regenerates tests/helpers/fuzz_seeds.h: parameter sets and the first 72
bytes of slice NAL units (all slice types, header extras on: reference list
modification, prediction weight tables, marking operations; field pictures of
an MBAFF sequence with pic_order_cnt_type 1; slice groups with a change cycle)
from the seeded stream generator, for tests/helpers/fuzz_ctl.c to mutate.

    python tests/helpers/make_fuzz_seeds.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import streams  # noqa: E402
from libh264_amd import gen  # noqa: E402
from test_oracle_vs_generator import fmo_cfg  # noqa: E402


def main():
    out = ["/* tests/helpers/fuzz_seeds.h -- DATA: parameter sets and the first "
           "bytes of slice NAL",
           " * units made by the stream generator (libh264_amd/csrc/gen), as "
           "seeds for",
           " * tests/helpers/fuzz_ctl.c to mutate.  Regenerate: "
           "tests/helpers/make_fuzz_seeds.py */"]

    def arr(name, b):
        out.append("static const uint8_t %s[] = {%s};" %
                   (name, ",".join("0x%02x" % x for x in b)))

    cfgs = [("a", streams.small_cfg(weighted_pred_flag=1,
                                    weighted_bipred_idc=1), {}),
            ("b", streams.small_cfg(
                field_coding=2, pic_order_cnt_type=1,
                bottom_field_pic_order_in_frame_present_flag=1),
             {"field_pic": 1}),
            ("c", fmo_cfg(3)[0], {"slice_group_change_cycle": 5})]
    sets = []
    for tag, cfg, kw in cfgs:
        arr("seed_sps_" + tag, gen.sps(cfg))
        arr("seed_pps_" + tag, gen.pps(cfg))
        k = 0
        for st in (0, 1, 2, 5, 6, 7, 3, 4):
            for seed in (3, 7):
                s = streams.rich_slice(cfg, 40 + seed + st, st, 0, 6,
                                       hdr_extras=7 if st % 5 != 2 else 0,
                                       **kw)
                arr("seed_slice_%s_%d" % (tag, k), gen.slice_nal(cfg, s)[:72])
                k += 1
        sets.append((tag, k))
    out.append("struct fuzz_seed_set { const uint8_t *sps; size_t sps_len; "
               "const uint8_t *pps; size_t pps_len; const uint8_t *const "
               "*slices; const size_t *lens; size_t n; };")
    for tag, k in sets:
        out.append("static const uint8_t *const seed_slices_%s[] = {%s};" %
                   (tag, ",".join("seed_slice_%s_%d" % (tag, i)
                                  for i in range(k))))
        out.append("static const size_t seed_lens_%s[] = {%s};" %
                   (tag, ",".join("sizeof seed_slice_%s_%d" % (tag, i)
                                  for i in range(k))))
    out.append("static const struct fuzz_seed_set fuzz_seed_sets[] = {" +
               ",".join("{seed_sps_%s,sizeof seed_sps_%s,seed_pps_%s,sizeof "
                        "seed_pps_%s,seed_slices_%s,seed_lens_%s,%d}" %
                        (t, t, t, t, t, t, k) for t, k in sets) + "};")
    with open(os.path.join(HERE, "fuzz_seeds.h"), "w") as f:
        f.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
