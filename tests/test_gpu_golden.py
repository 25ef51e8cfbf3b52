"""GPU parity against the COMMITTED golden fixtures (tests/golden/*.npz):
the product alone -- scan kernel, slice-decode kernel (T2, full macroblock
syntax), the drop-in reader, the synthetic-slice and CABAC kernels -- must
reproduce the recorded expectations.  The oracle is not consulted at run time
here; the fixtures carry what it said when they were made
(tests/golden/make_golden.py) and tests/test_oracle_pins.py keeps the two in
step on the CPU."""
import ctypes as C
import hashlib
import importlib.util
import os

import numpy as np
import pytest

import golden_io
import prod
from libh264_amd import Params
from libh264_amd.ctypes_abi import (Sps, Pps, SliceHeader, SliceResult,
                                    TIER_T2, t2_stride)

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STREAMS = ["mixed_small", "mixed_400", "mixed_422_10", "mixed_444",
           "mixed_interlaced"]
EV_SPS, EV_PPS, EV_NALU_BEGIN, EV_PARSE_RESULT = 8, 9, 2, 13


def structs(raw, ctype):
    n = len(raw) // C.sizeof(ctype)
    return [ctype.from_buffer_copy(raw[i * C.sizeof(ctype):
                                       (i + 1) * C.sizeof(ctype)].tobytes())
            for i in range(n)]


@pytest.mark.parametrize("name", STREAMS)
def test_scan_and_decode_kernels_vs_golden(engine, name):
    fx = golden_io.load(os.path.join(GOLDEN, name))
    ex = fx["expect"]
    data = fx["stream"]
    buf = np.frombuffer(data + b"\xff" * 64, dtype=np.uint8)
    d_buf = engine.upload(buf)
    tab, count, consumed = engine.scan(d_buf, len(data), len(data) // 3 + 8)
    assert count == len(ex["nals"])
    assert np.array_equal(np.asarray(tab, dtype=np.uint64), ex["nals"])

    # parameter-set epochs in stream order, slices tagged with the epoch
    # current at their position
    sps = structs(ex["sps"], Sps)
    pps = structs(ex["pps"], Pps)
    params = Params()
    descs = []
    for kind, a, b, c in ex["events"]:
        if kind == EV_SPS:
            params.put_sps(sps[int(a)])
        elif kind == EV_PPS:
            params.put_pps(pps[int(a)])
        elif kind == EV_NALU_BEGIN and (int(a) & 0x1f) in (1, 5):
            descs.append((int(b), int(c), params.epoch))
    want = structs(ex["results"], SliceResult)
    assert len(descs) == len(want) > 0
    cap = max(r.mb_count for r in want) + 2
    out = engine.decode_slices(params, d_buf, descs, reader_flags=1,
                               tier=TIER_T2, mb_capacity=cap,
                               want_headers=True)
    res = engine.fetch_results(out)
    hdrs = engine.fetch_headers(out)
    hsz = C.sizeof(SliceHeader)
    for i, w in enumerate(want):
        assert bytes(res[i]) == bytes(w), i
        assert bytes(hdrs[i]) == ex["headers"][i * hsz:(i + 1) * hsz].tobytes()
        st = t2_stride(w.chroma_array_type)
        off = int(ex["mboffs"][i])
        exp = ex["mbrecs"][off:off + w.mb_count * st].reshape(w.mb_count, st)
        got = engine.fetch_mb(out, i, w.mb_count)[:, :st]
        assert np.array_equal(got, exp), i


@pytest.mark.parametrize("name", STREAMS)
def test_dropin_reader_vs_golden(name):
    fx = golden_io.load(os.path.join(GOLDEN, name))
    ex = fx["expect"]
    got = prod.parse_stream(fx["stream"])
    assert got.result == 0
    want = ex["events"][ex["events"][:, 0] != EV_PARSE_RESULT]
    have = np.array(got.event_tuples(), dtype=np.uint64).reshape(-1, 4)
    assert np.array_equal(have, want)
    assert bytes(got.sps) == ex["sps"].tobytes()
    assert bytes(got.pps) == ex["pps"].tobytes()
    assert bytes(got.headers) == ex["headers"].tobytes()


@pytest.mark.parametrize("name", ["damaged_small", "damaged_overflow"])
def test_dropin_reader_damaged_vs_golden(name):
    fx = golden_io.load(os.path.join(GOLDEN, name))
    sizes = fx["extra"]["nal_sizes"]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
    nals = [fx["stream"][offs[i]:offs[i + 1]] for i in range(len(sizes))]
    got = prod.parse_nalus(nals)
    assert got.statuses == list(fx["extra"]["statuses"])
    have = np.array(got.event_tuples(), dtype=np.uint64).reshape(-1, 4)
    assert np.array_equal(have, fx["expect"]["events"])
    assert bytes(got.headers) == fx["expect"]["headers"].tobytes()


def test_decode_kernel_out_of_block_coefficients_vs_golden(engine):
    """damaged_overflow: slices that put a coefficient at index 15 of a
    15-coefficient block.  The kernel drops it and walks on like the reference
    (full T2 records and results as recorded)."""
    fx = golden_io.load(os.path.join(GOLDEN, "damaged_overflow"))
    ex = fx["expect"]
    sizes = fx["extra"]["nal_sizes"]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
    data = fx["stream"]
    params = Params()
    params.put_sps(structs(ex["sps"], Sps)[0])
    epoch = params.put_pps(structs(ex["pps"], Pps)[0])
    d_buf = engine.upload(np.frombuffer(data + b"\xff" * 64, dtype=np.uint8))
    descs = [(int(offs[i]), int(sizes[i]), epoch)
             for i in range(2, len(sizes))]
    want = structs(ex["results"], SliceResult)
    assert len(want) == len(descs) == 8
    out = engine.decode_slices(params, d_buf, descs, reader_flags=1,
                               tier=TIER_T2,
                               mb_capacity=max(w.mb_count for w in want) + 2,
                               want_headers=True)
    res = engine.fetch_results(out)
    for i, w in enumerate(want):
        assert bytes(res[i]) == bytes(w), i
        st = t2_stride(w.chroma_array_type)
        off = int(ex["mboffs"][i])
        exp = ex["mbrecs"][off:off + w.mb_count * st].reshape(w.mb_count, st)
        assert np.array_equal(engine.fetch_mb(out, i, w.mb_count)[:, :st],
                              exp), i


def test_synth_and_cabac_kernels_vs_golden(engine):
    spec = importlib.util.spec_from_file_location(
        "make_golden", os.path.join(GOLDEN, "make_golden.py"))
    # only the case table and the parameter sets are taken from the script;
    # importing it does not run the oracle
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    z = np.load(os.path.join(GOLDEN, "synth_grid.npz"))
    i = 0
    for entropy in (0, 1):
        sps, pps = mg.ps_1080p(entropy)
        params = Params()
        params.put_sps(sps)
        epoch = params.put_pps(pps)
        descs, hdrs = [], []
        for kind, mbc, nal_type, ref, sh in mg.synth_cases(entropy):
            descs.append((kind, epoch, mbc, ref, nal_type))
            hdrs.append(sh)
        got = engine.encode_synth(params, descs, hdrs, 1 << 14)
        for g in got:
            assert len(g) == z["sizes"][i], i
            assert hashlib.sha256(g).digest() == z["sha256"][i].tobytes(), i
            i += 1
    assert i == len(z["sizes"])

    z = np.load(os.path.join(GOLDEN, "cabac_init.npz"))
    triples = [tuple(int(x) for x in t) for t in z["triples"]]
    got = engine.cabac_init_states(triples)
    for k, d in enumerate(z["sha256"]):
        assert hashlib.sha256(got[k].tobytes()).digest() == d.tobytes(), k
