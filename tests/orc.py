"""ctypes binding of oracle/build/liboracle.so -- TEST INFRASTRUCTURE.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module."""
import ctypes as C
import os
import subprocess
import numpy as np
from h264types import (Sps, Pps, SliceHeader, NaluHeader, SliceResult, Ev,
                       Rewrite, Nal, t2_stride)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None

ORC_REC_WANT_T2 = 1
ORC_REC_WANT_REWRITE = 2
READER_FLAGS_SLICE_DATA = 1


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    if os.path.isdir("/root/reference/include/h264"):
        subprocess.check_call(
            ["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "build", "liboracle.so")
        build()  # make is a no-op when the library is up to date
        # a sanitizer build of the oracle made elsewhere (debugging aid)
        path = os.environ.get("H264_ORC_LIB", path)
        L = C.CDLL(path)
        L.orc_rec_new.restype = C.c_void_p
        L.orc_rec_new.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_rec_free.argtypes = [C.c_void_p]
        L.orc_rec_parse_stream.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t,
                                           C.POINTER(C.c_size_t)]
        L.orc_rec_parse_nalu.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_size_t]
        L.orc_rec_ctx.restype = C.c_void_p
        L.orc_rec_ctx.argtypes = [C.c_void_p]
        for name in ("events", "sps", "pps", "headers", "results", "mbrecs",
                     "mboffs", "rewrites", "rewrite_idx", "nals"):
            f = getattr(L, "orc_rec_" + name)
            f.restype = C.c_void_p
            f.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        L.orc_build_nalu.argtypes = [C.c_void_p, C.POINTER(NaluHeader),
                                     C.POINTER(Sps), C.POINTER(Pps),
                                     C.POINTER(SliceHeader), C.c_int,
                                     C.c_uint32, C.c_void_p, C.c_size_t]
        L.orc_ctx_new.argtypes = [C.POINTER(C.c_void_p)]
        L.orc_ctx_destroy.argtypes = [C.c_void_p]
        L.orc_cabac_init_states.argtypes = [C.c_int, C.c_uint32, C.c_int32,
                                            C.c_void_p]
        L.orc_find_nalu.argtypes = [C.c_char_p, C.c_size_t,
                                    C.POINTER(C.c_size_t),
                                    C.POINTER(C.c_size_t)]
        L.orc_parse_sps.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Sps)]
        L.orc_parse_pps.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Sps),
                                    C.POINTER(Pps)]
        L.orc_sizeof.restype = C.c_size_t
        L.orc_sizeof.argtypes = [C.c_char_p]
        L.orc_bench_slices.argtypes = [
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p,
            C.c_size_t, C.c_char_p, C.c_size_t, C.c_uint32, C.c_int,
            C.POINTER(C.c_double), C.POINTER(C.c_uint64),
            C.POINTER(C.c_uint64)]
        _LIB = L
    return _LIB


def _array(ptr, count, ctype):
    if not ptr or count == 0:
        return (ctype * 0)()
    return C.cast(ptr, C.POINTER(ctype * count)).contents


class Recording:
    """Result of parsing a stream with the oracle."""

    def __init__(self, rec, keep):
        L = lib()
        n = C.c_size_t()

        def grab(name, ctype):
            p = getattr(L, "orc_rec_" + name)(rec, C.byref(n))
            arr = _array(p, n.value, ctype)
            return type(arr).from_buffer_copy(arr) if n.value else arr

        self.events = grab("events", Ev)
        self.sps = grab("sps", Sps)
        self.pps = grab("pps", Pps)
        self.headers = grab("headers", SliceHeader)
        self.results = grab("results", SliceResult)
        self.mbrecs = bytes(grab("mbrecs", C.c_uint8))
        self.mboffs = list(grab("mboffs", C.c_uint64))
        self.rewrites = bytes(grab("rewrites", C.c_uint8))
        self.rewrite_idx = grab("rewrite_idx", Rewrite)
        self.nals = grab("nals", Nal)
        self.consumed = None
        self._keep = keep

    def event_tuples(self):
        return [(e.kind, e.a, e.b, e.c) for e in self.events]

    def mb_records(self, slice_index):
        """T2 records of one slice as a (mb_count, stride) uint8 array."""
        r = self.results[slice_index]
        stride = t2_stride(r.chroma_array_type)
        off = self.mboffs[slice_index]
        # the recorder keeps full records for the first 65536 macroblocks of a
        # slice only (ORC_REC_T2_CAP: a damaged skip run can produce millions)
        n = min(r.mb_count, 65536)
        a = np.frombuffer(self.mbrecs, dtype=np.uint8, count=n * stride,
                          offset=off)
        return a.reshape(n, stride)


def parse_stream(data, flags=READER_FLAGS_SLICE_DATA, want=ORC_REC_WANT_T2,
                 stop_after=0):
    """stop_after = n: h264_reader_stop() from the nalu_end callback of the
    n-th NAL unit; the parse is then taken up again, with the same reader, at
    the offset it returned (consumed_first), to the end."""
    L = lib()
    rec = L.orc_rec_new(flags, want)
    off = C.c_size_t()
    buf = C.create_string_buffer(bytes(data), len(data))
    L.orc_rec_stop_after.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_rec_stop_after.restype = None
    L.orc_rec_stop_after(rec, stop_after)
    L.orc_rec_parse_stream(rec, buf, len(data), C.byref(off))
    first = off.value
    if stop_after and first < len(data):
        L.orc_rec_stop_after(rec, 0)
        L.orc_rec_parse_stream(rec, C.c_char_p(C.addressof(buf) + first), len(data) - first,
                               C.byref(off))
        off.value += first
    out = Recording(rec, buf)
    out.consumed = off.value
    out.consumed_first = first
    L.orc_rec_free(rec)
    return out


def parse_nalus(nalus, flags=READER_FLAGS_SLICE_DATA, want=ORC_REC_WANT_T2):
    """Parse a list of NAL units (bytes, no start codes) with one reader."""
    L = lib()
    rec = L.orc_rec_new(flags, want)
    keep = []
    statuses = []
    for n in nalus:
        b = C.create_string_buffer(bytes(n), len(n))
        keep.append(b)
        statuses.append(L.orc_rec_parse_nalu(rec, C.addressof(b),
                                             C.addressof(b), len(n)))
    out = Recording(rec, keep)
    out.statuses = statuses
    L.orc_rec_free(rec)
    return out


class Builder:
    """Serialises parameter sets / headers / synthetic slices with the
    oracle's writer (the restatement of h264_write_nalu & co)."""

    def __init__(self):
        self.ctx = C.c_void_p()
        lib().orc_ctx_new(C.byref(self.ctx))

    def __del__(self):
        try:
            lib().orc_ctx_destroy(self.ctx)
        except Exception:
            pass

    def nalu(self, ref_idc, nal_type, sps=None, pps=None, sh=None, kind=0,
             mb_count=0, cap=1 << 20):
        nh = NaluHeader(0, ref_idc, nal_type)
        out = C.create_string_buffer(cap)
        res = lib().orc_build_nalu(
            self.ctx, C.byref(nh), C.byref(sps) if sps is not None else None,
            C.byref(pps) if pps is not None else None,
            C.byref(sh) if sh is not None else None, kind, mb_count, out, cap)
        if res < 0:
            raise OSError(-res, os.strerror(-res))
        return out.raw[:res]


def cabac_init_states(slice_type, idc, qp):
    st = (C.c_uint8 * 2048)()
    lib().orc_cabac_init_states(slice_type, idc, qp, st)
    return np.frombuffer(bytes(st), dtype=np.uint8).reshape(1024, 2)


def find_nalu(data):
    s, e = C.c_size_t(), C.c_size_t()
    res = lib().orc_find_nalu(bytes(data), len(data), C.byref(s), C.byref(e))
    return res, s.value, e.value
