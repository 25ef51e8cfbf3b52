"""Golden fixtures (tests/golden/*.npz): an input byte stream plus the canonical
records the oracle produced for it when the fixture was made (event log, NAL
table, parameter sets, slice headers, per-slice results incl. checksums, full
T2 macroblock records).  Written by tests/golden/make_golden.py."""
import hashlib

import numpy as np


def _bytes_of(ctypes_array):
    return np.frombuffer(bytes(ctypes_array), dtype=np.uint8).copy()


def summarise(rec):
    """Oracle (or product) recording -> dict of numpy arrays."""
    ev = np.array(rec.event_tuples(), dtype=np.uint64).reshape(-1, 4)
    out = {
        "events": ev,
        "sps": _bytes_of(rec.sps),
        "pps": _bytes_of(rec.pps),
        "headers": _bytes_of(rec.headers),
    }
    if hasattr(rec, "results"):
        out["results"] = _bytes_of(rec.results)
        out["nals"] = np.array([(n.start, n.end) for n in rec.nals],
                               dtype=np.uint64).reshape(-1, 2)
        out["mbrecs"] = np.frombuffer(rec.mbrecs, dtype=np.uint8).copy()
        out["mboffs"] = np.array(rec.mboffs, dtype=np.uint64)
    return out


def save(path, stream, rec, **extra):
    arrays = {"expect_" + k: v for k, v in summarise(rec).items()}
    arrays["stream"] = np.frombuffer(bytes(stream), dtype=np.uint8)
    arrays["sha256"] = np.frombuffer(
        hashlib.sha256(bytes(stream)).hexdigest().encode(), dtype=np.uint8)
    for k, v in extra.items():
        arrays[k] = np.asarray(v)
    np.savez_compressed(path + ".npz", **arrays)


def load(path):
    z = np.load(path + ".npz")
    fx = {"stream": z["stream"].tobytes(),
          "sha256": z["sha256"].tobytes().decode(),
          "expect": {k[len("expect_"):]: z[k] for k in z.files
                     if k.startswith("expect_")},
          "extra": {k: z[k] for k in z.files
                    if not k.startswith("expect_") and
                    k not in ("stream", "sha256")}}
    return fx
