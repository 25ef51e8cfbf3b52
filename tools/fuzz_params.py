#!/usr/bin/env python3
"""Hostile PARAMETER SETS: the reference range-checks almost nothing in an
SPS/PPS, so every 32-bit value can reach the slice header and slice data
parsers.  Each seed writes SPS/PPS structures with extreme field values through
the oracle's writer, re-parses them (what is tested is what a parse yields),
and decodes slices -- intact and damaged, made for the ORIGINAL parameters --
under them.  Engine vs oracle: status, result record, header image,
macroblock records.

    python tools/fuzz_params.py [first_seed] [count]
"""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import libh264_amd  # noqa: E402
import orc  # noqa: E402
import streams  # noqa: E402
from libh264_amd import gen  # noqa: E402
from libh264_amd.ctypes_abi import TIER_T2, TIER_NONE, t2_stride  # noqa: E402
from test_gpu_decode import make_params, result_fields  # noqa: E402
from test_gpu_fuzz import mutate  # noqa: E402

WIDTHS = [0, 0, 0, 1, 2, 8, 24, 28, 29, 60, 61, 250, 252, 4000, 0xfffffffc,
          0xfffffffd]


def hostile_sps(sps, rnd):
    s = type(sps).from_buffer_copy(bytes(sps))
    for _ in range(rnd.randrange(1, 4)):
        f = rnd.randrange(11)
        if f == 0:
            s.chroma_format_idc = rnd.choice([0, 1, 2, 3, 4, 255, 256, 70000])
        elif f == 1:
            s.log2_max_frame_num_minus4 = rnd.choice(WIDTHS)
        elif f == 2:
            s.pic_order_cnt_type = rnd.choice([0, 1, 2, 3, 255, 256, 257])
        elif f == 3:
            s.log2_max_pic_order_cnt_lsb_minus4 = rnd.choice(WIDTHS)
        elif f == 4:
            s.frame_mbs_only_flag = rnd.randrange(2)
            s.mb_adaptive_frame_field_flag = rnd.randrange(2)
        elif f == 5:
            s.pic_width_in_mbs_minus1 = rnd.choice(
                [0, 1, 6, 7, 8, 47, 1055, 1056, 5000, 0xfffffffe])
        elif f == 6:
            s.pic_height_in_map_units_minus1 = rnd.choice(
                [0, 1, 5, 6, 100, 0xfffffffe])
        elif f == 7:
            s.bit_depth_luma_minus8 = rnd.randrange(7)
            s.bit_depth_chroma_minus8 = rnd.randrange(7)
        elif f == 8:
            s.separate_colour_plane_flag = rnd.randrange(2)
        elif f == 9:
            s.delta_pic_order_always_zero_flag = rnd.randrange(2)
            s.pic_order_cnt_type = 1
        else:
            s.direct_8x8_inference_flag = rnd.randrange(2)
    return s


def hostile_pps(pps, rnd):
    p = type(pps).from_buffer_copy(bytes(pps))
    for _ in range(rnd.randrange(1, 4)):
        f = rnd.randrange(10)
        if f == 0:
            p.num_ref_idx_l0_default_active_minus1 = rnd.choice(
                [0, 1, 3, 31, 32, 33, 255, 0xfffffffe])
            p.num_ref_idx_l1_default_active_minus1 = rnd.choice(
                [0, 1, 3, 31, 32, 33, 255, 0xfffffffe])
        elif f == 1:
            p.weighted_pred_flag = rnd.randrange(2)
            p.weighted_bipred_idc = rnd.randrange(4)
        elif f == 2:
            p.pic_init_qp_minus26 = rnd.choice([-26, 25, -27, 26, -1000, 1000,
                                                -0x7fffffff, 0x7fffffff])
        elif f == 3:
            p.bottom_field_pic_order_in_frame_present_flag = rnd.randrange(2)
        elif f == 4:
            p.redundant_pic_cnt_present_flag = rnd.randrange(2)
        elif f == 5:
            p.deblocking_filter_control_present_flag = rnd.randrange(2)
        elif f == 6:
            p.transform_8x8_mode_flag = rnd.randrange(2)
            p._more_rbsp_data_present = 1
        elif f == 7:
            p.num_slice_groups_minus1 = rnd.choice([0, 1, 2, 7])
            p.slice_group_map_type = rnd.choice([0, 1, 2, 3, 4, 5, 6, 7, 255, 259])
            p.slice_group_change_rate_minus1 = rnd.choice([0, 1, 47, 0xfffffffe])
            p.pic_size_in_map_units_minus1 = rnd.choice([0, 47, 255])
        elif f == 8:
            p.entropy_coding_mode_flag = rnd.randrange(2)
        else:
            p.pic_init_qs_minus26 = rnd.choice([-26, 25, 1000, -1000])
    return p


def one(engine, seed):
    rnd = random.Random(seed)
    cf, bd = rnd.choice([(1, 0), (1, 0), (0, 0), (2, 0), (3, 0), (1, 2)])
    cfg = streams.small_cfg(cf=cf, bd=bd,
                            transform_8x8_mode_flag=rnd.randrange(2),
                            weighted_pred_flag=rnd.randrange(2),
                            weighted_bipred_idc=rnd.randrange(3))
    base = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg)])
    b = orc.Builder()
    total = cfg.width_mbs * cfg.height_mbs
    nals = [gen.sps(cfg), gen.pps(cfg)]
    slice_pos = []
    for k in range(60):
        if k % 6 == 3:
            try:
                sps = hostile_sps(base.sps[0], rnd)
                nals.append(b.nalu(3, 7, sps=sps))
                if rnd.randrange(2):
                    nals.append(b.nalu(3, 8, sps=sps,
                                       pps=hostile_pps(base.pps[0], rnd)))
            except OSError:
                pass  # the writer refused the structure
        elif k % 6 == 0 and k:
            try:
                nals.append(b.nalu(3, 8, sps=base.sps[0],
                                   pps=hostile_pps(base.pps[0], rnd)))
            except OSError:
                pass
        st = rnd.choice([2, 0, 1, 7, 5, 6, 3, 4])
        first = rnd.choice([0, 0, rnd.randrange(total - 1)])
        n = rnd.randrange(1, total - first + 1)
        s = streams.rich_slice(cfg, seed * 991 + k, st, first, n,
                               density=rnd.randrange(0, 12),
                               pm_pcm=rnd.choice([0, 30, 300]),
                               hdr_extras=rnd.randrange(8) if st % 5 != 2 else 0)
        nal = gen.slice_nal(cfg, s)
        slice_pos.append(len(nals))
        nals.append(mutate(nal, rnd) if k % 4 == 1 else nal)
    # 0x100: the engine's CABAC subset decoder (a hostile PPS turns CAVLC
    # slices into 'CABAC' ones: error paths of the arithmetic decoder)
    flags = rnd.choice([1, 1, 0x101, 0x101, 0])
    ref = orc.parse_nalus(nals, flags=flags)
    sl = [nals[i] for i in slice_pos]
    # every slice binds to the parameter sets current at ITS position
    params = libh264_amd.Params()
    epochs, it = {}, iter(slice_pos)
    nxt = next(it)
    ev = [(e.kind, e.a) for e in ref.events if e.kind in (8, 9, 13)]
    nal_i = 0
    for kind, a in ev:
        if kind == 8:
            params.put_sps(ref.sps[a])
        elif kind == 9:
            params.put_pps(ref.pps[a])
        else:  # parse result of NAL nal_i
            if nal_i in slice_pos:
                epochs[nal_i] = params.epoch
            nal_i += 1
    buf, offs, sizes = streams.pack_nals(sl, lead=seed % 4)
    d_buf = engine.upload(buf)
    descs = [(o, s, epochs[p]) for o, s, p in zip(offs, sizes, slice_pos)]
    cap = 4 * total
    tier = TIER_T2 if flags else TIER_NONE
    if flags == 0:
        ref_w = orc.parse_nalus(nals, flags=0, want=orc.ORC_REC_WANT_REWRITE)
    out = engine.decode_slices(params, d_buf, descs, reader_flags=flags,
                               tier=tier, mb_capacity=cap, want_headers=True)
    res = engine.fetch_results(out)
    hdrs = engine.fetch_headers(out)
    k_ok = 0
    for i in range(len(sl)):
        want, got = result_fields(ref.results[i]), result_fields(res[i])
        hdr_idx = k_ok
        k_ok += want["status"] == 0
        if want["mb_count"] > cap or (want["mb_count"] == cap and
                                      got["status"] == -75):
            if not (got["status"] == -75 and got["mb_count"] == cap):
                return "seed %d slice %d overflow mismatch %r" % (seed, i, got)
            continue
        if got != want:
            diff = {k: (got[k], want[k]) for k in got if got[k] != want[k]}
            return "seed %d slice %d (nal %d): %r" % (seed, i, slice_pos[i],
                                                       diff)
        if want["status"] == 0 and bytes(hdrs[i]) != bytes(ref.headers[hdr_idx]):
            return "seed %d slice %d header image differs" % (seed, i)
        if want["mb_count"] and flags:
            st = t2_stride(want["chroma_array_type"])
            mb = engine.fetch_mb(out, i, want["mb_count"])[:, :st]
            if not np.array_equal(mb, ref.mb_records(i)):
                return "seed %d slice %d records differ" % (seed, i)
    if flags == 0:
        # write-back of every parsed header under the hostile parameters:
        # same bytes as the oracle's h264_write_nalu restatement, or the same
        # refusal
        d_out, ooffs, osizes = engine.write_slices(params, d_buf, out,
                                                   int(max(sizes)) + 4200)
        raw = engine.download(d_out)
        rew = {w.nal_index: w for w in ref_w.rewrite_idx}
        for i, p in enumerate(slice_pos):
            if res[i].status != 0 or p not in rew:
                continue
            w = rew[p]
            if w.res < 0:
                if int(osizes[i]) != w.res:
                    return "seed %d slice %d write status %d != %d" % (
                        seed, i, int(osizes[i]), w.res)
                continue
            got = raw[int(ooffs[i]):int(ooffs[i]) + max(int(osizes[i]), 0)]
            if int(osizes[i]) != w.len or \
                    got.tobytes() != ref_w.rewrites[w.off:w.off + w.len]:
                return "seed %d slice %d (nal %d) write-back differs (%d vs " \
                       "%d bytes)" % (seed, i, p, int(osizes[i]), w.len)
    return None


def main():
    import resource
    resource.setrlimit(resource.RLIMIT_AS, (96 << 30, 96 << 30))
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    eng = libh264_amd.Engine(0)
    bad = 0
    for seed in range(first, first + count):
        msg = one(eng, seed)
        if msg:
            print(msg, flush=True)
            bad += 1
        if (seed - first) % 50 == 49:
            print("... %d seeds, %d divergences" % (seed - first + 1, bad),
                  flush=True)
    print("seeds %d..%d divergences: %d" % (first, first + count - 1, bad))


if __name__ == "__main__":
    main()
