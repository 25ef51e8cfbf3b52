"""GPU leg of tests/test_cabac_real_streams.py: the CABAC kernel on the two
third-party streams the image holds, through the C-ABI, against the oracle's
decoder -- every slice to its last macroblock and to the stop bit, checksums
over the full syntax equal."""
import pytest

from libh264_amd.ctypes_abi import (READER_FLAGS_CABAC,
                                    READER_FLAGS_CABAC_X264_PRE151, TIER_T0,
                                    TIER_T2)
from test_cabac_real_streams import (ends_at_stop_bit, quirk, sample,
                                     slices_of)
from test_gpu_cabac import compare

pytestmark = pytest.mark.gpu
FLAGS = 1 | READER_FLAGS_CABAC


def test_high_profile_stream(engine):
    head, nals = slices_of(sample("realshort.mp4"))
    ref, res = compare(engine, head, nals, tier=TIER_T2, cap=302, flags=FLAGS)
    assert [(r.status, r.mb_count) for r in res] == [(0, 300)] * 36
    assert all(ends_at_stop_bit(n, r.end_bit) for n, r in zip(nals, res))


def test_x264_444_stream(engine):
    head, nals = slices_of(sample("cockatoo.mp4"))
    # as the standard has it: lost in the first picture, kernel and oracle at
    # the same macroblock
    ref, res = compare(engine, head, nals[:8], tier=TIER_T0, cap=3602,
                       flags=FLAGS)
    assert res[0].mb_count < 3600
    # as x264 before build 151 wrote it
    try:
        quirk(True)
        ref, res = compare(engine, head, nals, tier=TIER_T0, cap=3602,
                           flags=FLAGS | READER_FLAGS_CABAC_X264_PRE151)
    finally:
        quirk(False)
    assert [(r.status, r.mb_count) for r in res] == [(0, 3600)] * 280
    assert all(ends_at_stop_bit(n, r.end_bit) for n, r in zip(nals, res))


def all_nals(path):
    import mp4_demux
    sps, pps, samples = mp4_demux.video_nals(path)
    return [bytes(x) for x in sps] + [bytes(x) for x in pps] + \
        [bytes(n) for s in samples for n in s]


@pytest.mark.parametrize("name", ["realshort.mp4", "cockatoo.mp4"])
def test_drop_in_library_on_real_streams(name, tmp_path):
    """The reference's own behaviour on these files (CABAC slice data is not
    walked: header, raw extent, no slice_data callbacks): callback logs,
    parameter sets and the 3968-byte slice header images of libh264.so against
    the oracle's reader -- real SPS / PPS / SEI (x264's user data), reference
    list modification, weighted-prediction tables, memory management
    commands; then tools/h264_dump/h264-dump on the Annex-B form: JSON for every
    NAL unit and "OK" = h264_write_nalu reproduces a real encoder's bytes."""
    import json
    import orc
    import prod
    import streams
    from test_gpu_dropin import same_log
    from test_gpu_dump_tool import run_tool
    nals = all_nals(sample(name))
    data = streams.annexb(nals)
    for flags in (0, 1):
        ref = orc.parse_stream(data, flags=flags, want=0)
        got = prod.parse_stream(data, flags=flags)
        assert got.result == 0 and got.consumed == ref.consumed
        same_log(ref, got, stream=True)
        assert all(r.status == 0 for r in ref.results)
    f = tmp_path / (name + ".h264")
    f.write_bytes(data)
    lines = run_tool(f).stdout.splitlines()
    assert len(lines) == 2 * len(nals)
    for k in range(len(nals)):
        json.loads(lines[2 * k])
        assert lines[2 * k + 1] == "OK", (k, nals[k][0] & 31)


def test_damaged_real_slices(engine):
    """bit flips, truncations and splices of the real slices: kernel and
    oracle end every one of them the same way (status, macroblocks before the
    error, checksum)"""
    import random
    from test_gpu_fuzz import mutate
    rnd = random.Random(2024)
    head, nals = slices_of(sample("cockatoo.mp4"))
    head2, nals2 = slices_of(sample("realshort.mp4"))
    picked = [nals[k] for k in range(0, 280, 7)]
    bad = [mutate(n, rnd) for n in picked for _ in range(3)]
    for flags, q in ((FLAGS, False),
                     (FLAGS | READER_FLAGS_CABAC_X264_PRE151, True)):
        try:
            quirk(q)
            compare(engine, head, bad, tier=TIER_T0, cap=3602, flags=flags)
        finally:
            quirk(False)
    bad2 = [mutate(n, rnd) for n in nals2 for _ in range(4)]
    compare(engine, head2, bad2, tier=TIER_T2, cap=302, flags=FLAGS)


def test_against_committed_expectations(engine):
    """the kernel alone against tests/golden/real_streams_expect.json (made by
    tests/golden/make_real_expect.py from the oracle's decoder; the clips are
    identified by SHA-256): the oracle is not consulted here"""
    import hashlib
    import json
    import os
    import numpy as np
    import orc
    import streams
    from libh264_amd import Params
    exp = json.load(open(os.path.join(os.path.dirname(__file__), "golden",
                                      "real_streams_expect.json")))
    for name, e in exp.items():
        path = sample(name)
        if hashlib.sha256(open(path, "rb").read()).hexdigest() != e["sha256"]:
            pytest.skip("another %s than the one the expectations are for" % name)
        head, nals = slices_of(path)
        ps = orc.parse_nalus(head, flags=0, want=0)  # parameter sets only
        params = Params()
        params.put_sps(ps.sps[0])
        epoch = params.put_pps(ps.pps[0])
        buf, offs, sizes = streams.pack_nals(nals)
        cap = max(r[1] for r in e["slices"]) + 2
        flags = FLAGS | (READER_FLAGS_CABAC_X264_PRE151
                         if e["x264_before_151"] else 0)
        out = engine.decode_slices(
            params, engine.upload(buf),
            [(o, z, epoch) for o, z in zip(offs, sizes)], reader_flags=flags,
            tier=TIER_T0, mb_capacity=cap, want_headers=True)
        res = engine.fetch_results(out)
        got = [[r.status, r.mb_count, r.end_bit, "%016x" % r.checksum]
               for r in res]
        assert got == e["slices"], name


def test_fuzz_seeds_that_found_something(engine):
    """tools/fuzz_cabac_real.py seeds kept as regressions.  130473: a damaged P
    slice whose motion vector difference no longer fits 16 bits -- the contexts
    of the neighbouring blocks see the magnitude of the STORED value (int16_t,
    as in the reference's macroblock struct), the kernel used the unwrapped
    one."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "tools"))
    import fuzz_cabac_real as fr
    sample("cockatoo.mp4")
    sample("realshort.mp4")
    clips = fr.load_clips()
    for seed in (130473,):
        fr.one(engine, clips, seed)
