"""GPU parity for the CABAC kernel (h264_amd_k_decode_cabac, engine extension
H264_AMD_READER_FLAGS_CABAC) on the part of the syntax that has reference
behaviour -- what the reference's writers emit -- through the C-ABI,
against the oracle's CPU statement of the same decoder: results, header
images, macroblock records in every tier -- on the reference writers' own
slices (round trip), on slices from the independent Python encoder, and on
damaged / opaque CABAC data."""
import random

import numpy as np
import pytest

import cabac_enc
import orc
import streams
from libh264_amd import gen, Params
from libh264_amd.ctypes_abi import (READER_FLAGS_CABAC_SUBSET, TIER_NONE,
                                    TIER_T0, TIER_T1, TIER_T2, t2_stride)
from test_cabac_subset import (FLAGS, GRID, custom_i_slice, header,
                               param_sets, writer_slices)
from test_gpu_decode import make_params, result_fields
from test_gpu_fuzz import mutate

pytestmark = pytest.mark.gpu


def compare(engine, head, nals, tier=TIER_T2, cap=None, flags=FLAGS,
            lead=0):
    ref = orc.parse_nalus(head + nals, flags=flags)
    params = make_params(ref)
    buf, offs, sizes = streams.pack_nals(nals, lead=lead)
    d_buf = engine.upload(buf)
    descs = [(o, s, params.epoch) for o, s in zip(offs, sizes)]
    if cap is None:
        cap = max([r.mb_count for r in ref.results] + [1]) + 2
    out = engine.decode_slices(params, d_buf, descs, reader_flags=flags,
                               tier=tier, mb_capacity=cap, want_headers=True)
    res = engine.fetch_results(out)
    k_ok = 0
    hdrs = engine.fetch_headers(out)
    for i in range(len(nals)):
        want, got = result_fields(ref.results[i]), result_fields(res[i])
        hdr_idx = k_ok
        k_ok += want["status"] == 0
        if want["mb_count"] > cap or (want["mb_count"] == cap and
                                      got["status"] == -75):
            # room for `cap` records: the kernel stops there, before it would
            # look at what follows the cap-th macroblock
            assert got["status"] == -75 and got["mb_count"] == cap, (i, got)
            n = cap
        else:
            assert got == want, (i, got, want)
            n = want["mb_count"]
            if want["status"] == 0:
                assert bytes(hdrs[i]) == bytes(ref.headers[hdr_idx]), i
        if n and tier != TIER_NONE:
            st = t2_stride(want["chroma_array_type"])
            width = {TIER_T2: st, TIER_T1: 64, TIER_T0: 4}[tier]
            mb = engine.fetch_mb(out, i, n)[:, :width]
            assert np.array_equal(mb, ref.mb_records(i)[:n, :width]), i
    return ref, res


@pytest.mark.parametrize("tier", [TIER_T2, TIER_T1, TIER_T0, TIER_NONE])
def test_roundtrip_of_the_synthetic_slice_kernels(engine, tier):
    """encode on the GPU (h264_write_grey_i_slice / skipped_p_slice kernels),
    decode on the GPU: every macroblock comes back."""
    sps, pps = param_sets()
    b = orc.Builder()
    head = [b.nalu(3, 7, sps=sps), b.nalu(3, 8, sps=sps, pps=pps)]
    params = Params()
    params.put_sps(sps)
    epoch = params.put_pps(pps)
    descs, hdrs, want = [], [], []
    for k, (mbc, first, qpd, idc) in enumerate(GRID):
        for kind, st, nal_type, ref_idc in ((0, 7, 5, 3), (1, 5, 1, 2),
                                            (1, 1, 1, 0)):
            descs.append((kind, epoch, mbc, ref_idc, nal_type))
            hdrs.append(header(st, first, qpd, idc, k))
            want.append(mbc)
    nals = engine.encode_synth(params, descs, hdrs, 1 << 14)
    assert all(isinstance(n, bytes) for n in nals)
    ref, res = compare(engine, head, nals, tier=tier)
    for i, mbc in enumerate(want):
        assert (res[i].status, res[i].mb_count, res[i].flags) == (0, mbc, 7)
    # without the extension flag: the reference's behaviour (raw extent only)
    ref0, res0 = compare(engine, head, nals, tier=tier, flags=1)
    assert all(r.mb_count == 0 and r.flags == 5 for r in res0)


@pytest.mark.parametrize("width,cf,lead", [(120, 1, 0), (8, 1, 3), (11, 2, 1),
                                           (5, 0, 2), (300, 1, 0)])
def test_values_from_the_python_encoder(engine, width, cf, lead):
    rnd = random.Random(width * 7 + cf)
    height = 9 if width < 200 else 3
    sps, pps = param_sets(width, height, cf)
    b = orc.Builder()
    head = [b.nalu(3, 7, sps=sps), b.nalu(3, 8, sps=sps, pps=pps)]
    nals = []
    for k in range(24):
        first = rnd.choice([0, 0, rnd.randrange(width * height - 1)])
        n = rnd.randrange(1, width * height - first + 1)
        mbs = [(rnd.randrange(4), rnd.randrange(4),
                rnd.choice([0, 0, 0, 1, -1, 2, -3, 7, -26, 25]))
               for _ in range(n)]
        nals.append(custom_i_slice(b, sps, pps, mbs, first,
                                   rnd.randrange(-10, 10), k))
    ref, res = compare(engine, head, nals, lead=lead)
    assert all(r.status == 0 for r in res)


def test_damaged_and_foreign_cabac_data(engine):
    """bit flips, truncations and spliced tails of subset slices, opaque CABAC
    payloads of P/B/I slices, CAVLC slices in the same batch."""
    rnd = random.Random(11)
    sps, pps = param_sets(8, 6, 1)
    b = orc.Builder()
    head = [b.nalu(3, 7, sps=sps), b.nalu(3, 8, sps=sps, pps=pps)]
    nals = []
    for k in range(160):
        kind = k % 4
        if kind == 0:
            n = rnd.randrange(1, 49)
            mbs = [(rnd.randrange(4), rnd.randrange(4),
                    rnd.choice([0, 0, 1, -2, 5])) for _ in range(n)]
            nal = custom_i_slice(b, sps, pps, mbs, 0, rnd.randrange(-5, 5), k)
        elif kind == 1:
            st = rnd.choice([5, 6])
            sh = header(st, rnd.randrange(0, 20), rnd.randrange(-5, 5),
                        rnd.randrange(3), k)
            nal = b.nalu(2, 1, sps=sps, pps=pps, sh=sh, kind=2,
                         mb_count=rnd.randrange(1, 28), cap=1 << 14)
        elif kind == 2:
            sh = header(7, 0, 0, 0, k)
            nal = b.nalu(3, 5, sps=sps, pps=pps, sh=sh, kind=1,
                         mb_count=rnd.randrange(1, 48), cap=1 << 14)
        else:   # opaque payload behind a valid header
            st = rnd.choice([5, 6, 7])
            sh = header(st, 0, 0, rnd.randrange(3), k)
            hd = b.nalu(3 if st == 7 else 2, 5 if st == 7 else 1, sps=sps,
                        pps=pps, sh=sh, kind=1 if st == 7 else 2, mb_count=1,
                        cap=1 << 14)
            r = orc.parse_nalus(head + [hd], flags=0).results[0]
            nal = cabac_enc.escape_tail(
                hd[:r.rawdata_off],
                bytes(rnd.randrange(256) for _ in range(rnd.randrange(0, 60))))
        nals.append(mutate(nal, rnd) if (k % 8) >= 3 else nal)
    ref, res = compare(engine, head, nals, cap=200)
    st = [r.status for r in res]
    assert st.count(0) > 40 and st.count(-5) > 5


def test_mixed_batch_with_cavlc_and_capacity(engine):
    """CAVLC and CABAC parameter sets side by side; record capacity smaller
    than a CABAC slice delivers (H264_AMD_E_MB_OVERFLOW after `cap`)."""
    cfg = streams.small_cfg()
    sps, pps = param_sets(8, 6, 1)
    pps.pic_parameter_set_id = 3
    sps.seq_parameter_set_id = 2
    pps.seq_parameter_set_id = 2
    b = orc.Builder()
    head = [gen.sps(cfg), gen.pps(cfg), b.nalu(3, 7, sps=sps),
            b.nalu(3, 8, sps=sps, pps=pps)]
    nals = []
    for k in range(12):
        nals.append(gen.slice_nal(cfg, streams.rich_slice(cfg, k, (2, 0, 1)[k % 3])))
        sh = header(5, 0, 0, 0, k)
        sh.pic_parameter_set_id = 3
        nals.append(b.nalu(2, 1, sps=sps, pps=pps, sh=sh, kind=2,
                           mb_count=20 + k, cap=1 << 14))
    compare(engine, head, nals, cap=60)
    compare(engine, head, nals, cap=25, tier=TIER_T0)
