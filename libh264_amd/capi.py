"""ctypes binding of the engine C-ABI (include/h264_amd.h)."""
import ctypes as C
import os

import numpy as np

from .ctypes_abi import (DecodeArgs, Nal, SliceDesc, SliceHeader, SliceResult,
                         Sps, Pps, SynthArgs, SynthDesc, WriteArgs, t2_stride,
                         TIER_NONE, TIER_T0, TIER_T1, TIER_T2)

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class EngineError(OSError):
    pass


def _check(res, what):
    if res < 0:
        raise EngineError(-res, "%s: %s" % (what, os.strerror(-res)))
    return res


def engine_lib():
    """Load libh264_amd_engine.so.  Raises if it has not been built: the HIP
    path is the product, there is nothing to fall back to."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # H264_AMD_ENGINE_LIB: development knob, another build of the same engine
    # (tools/ab_decode.py times kernel variants side by side)
    path = os.environ.get("H264_AMD_ENGINE_LIB") or \
        os.path.join(HERE, "lib", "libh264_amd_engine.so")
    if not os.path.exists(path):
        raise EngineError(
            2, "%s missing: run `python -m libh264_amd.build`" % path)
    L = C.CDLL(path)
    vp, sz, u32 = C.c_void_p, C.c_size_t, C.c_uint32
    L.h264_amd_available.restype = C.c_int
    L.h264_amd_engine_new.argtypes = [C.c_int, vp, C.POINTER(vp)]
    L.h264_amd_engine_destroy.argtypes = [vp]
    L.h264_amd_engine_sync.argtypes = [vp]
    L.h264_amd_engine_stream.restype = vp
    L.h264_amd_engine_stream.argtypes = [vp]
    L.h264_amd_malloc.argtypes = [vp, sz, C.POINTER(vp)]
    L.h264_amd_free.argtypes = [vp, vp]
    L.h264_amd_upload.argtypes = [vp, vp, vp, sz]
    L.h264_amd_download.argtypes = [vp, vp, vp, sz]
    L.h264_amd_memset.argtypes = [vp, vp, C.c_int, sz]
    L.h264_amd_copy.argtypes = [vp, vp, vp, sz]
    L.h264_amd_last_kernel_ms.argtypes = [vp, C.c_char_p,
                                          C.POINTER(C.c_float)]
    L.h264_amd_params_new.argtypes = [C.POINTER(vp)]
    L.h264_amd_params_destroy.argtypes = [vp]
    L.h264_amd_params_reset.argtypes = [vp]
    L.h264_amd_params_put_sps.argtypes = [vp, C.POINTER(Sps)]
    L.h264_amd_params_put_pps.argtypes = [vp, C.POINTER(Pps)]
    L.h264_amd_params_epoch.argtypes = [vp]
    L.h264_amd_scan.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz),
                                C.POINTER(sz)]
    L.h264_amd_gather_nal_heads.argtypes = [vp, vp, vp, sz, u32, vp]
    L.h264_amd_byte_stream_to_avcc.argtypes = [vp, vp, sz]
    L.h264_amd_decode_slices.argtypes = [vp, vp, C.POINTER(DecodeArgs)]
    L.h264_amd_write_slices.argtypes = [vp, vp, C.POINTER(WriteArgs)]
    L.h264_amd_encode_synth_slices.argtypes = [vp, vp, C.POINTER(SynthArgs)]
    L.h264_amd_cabac_init_states.argtypes = [vp, vp, u32, vp]
    L.h264_amd_probe_read.argtypes = [vp, vp, sz, C.POINTER(C.c_float)]
    L.h264_amd_build_id.restype = C.c_char_p
    _LIB = L
    return L


def build_id():
    """identity of the sources the loaded engine library was built from"""
    return engine_lib().h264_amd_build_id().decode()


def available():
    """True when the HIP library is built and HIP reports a device."""
    try:
        return bool(engine_lib().h264_amd_available())
    except OSError:
        return False


class Params:
    """Parameter-set snapshots (struct h264_amd_params)."""

    def __init__(self):
        self.h = C.c_void_p()
        _check(engine_lib().h264_amd_params_new(C.byref(self.h)),
               "params_new")

    def __del__(self):
        try:
            engine_lib().h264_amd_params_destroy(self.h)
        except Exception:
            pass

    def put_sps(self, sps):
        return _check(engine_lib().h264_amd_params_put_sps(
            self.h, C.byref(sps)), "put_sps")

    def put_pps(self, pps):
        return _check(engine_lib().h264_amd_params_put_pps(
            self.h, C.byref(pps)), "put_pps")

    @property
    def epoch(self):
        return engine_lib().h264_amd_params_epoch(self.h)


class DevBuf:
    def __init__(self, eng, nbytes):
        self.eng = eng
        self.nbytes = nbytes
        self.ptr = C.c_void_p()
        _check(engine_lib().h264_amd_malloc(eng.h, max(nbytes, 1),
                                            C.byref(self.ptr)), "malloc")

    def free(self):
        if self.ptr:
            engine_lib().h264_amd_free(self.eng.h, self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ExtBuf:
    """Device memory owned by somebody else (a torch tensor's storage) seen as
    a DevBuf: pointer + size, never freed here."""

    def __init__(self, ptr, nbytes):
        self.ptr = C.c_void_p(int(ptr))
        self.nbytes = int(nbytes)

    def free(self):
        pass


class Engine:
    """One engine per GPU (struct h264_amd_engine)."""

    def __init__(self, device=0, stream=None):
        self.h = C.c_void_p()
        _check(engine_lib().h264_amd_engine_new(
            device, C.c_void_p(stream) if stream else None,
            C.byref(self.h)), "engine_new")

    def close(self):
        if self.h:
            engine_lib().h264_amd_engine_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- memory ---------------------------------------------------------
    def alloc(self, nbytes):
        return DevBuf(self, int(nbytes))

    def upload(self, array, buf=None):
        a = np.ascontiguousarray(array)
        if buf is None:
            buf = self.alloc(a.nbytes)
        if a.nbytes:
            _check(engine_lib().h264_amd_upload(
                self.h, buf.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes),
                "upload")
        return buf

    def download(self, buf, nbytes=None, dtype=np.uint8, offset=0):
        n = buf.nbytes - offset if nbytes is None else int(nbytes)
        out = np.empty(n, dtype=np.uint8)
        if n:
            _check(engine_lib().h264_amd_download(
                self.h, out.ctypes.data_as(C.c_void_p),
                C.c_void_p(buf.ptr.value + offset), n), "download")
        return out.view(dtype)

    def memset(self, buf, value=0):
        _check(engine_lib().h264_amd_memset(self.h, buf.ptr, value,
                                            buf.nbytes), "memset")

    def copy_d2d(self, dst, dst_off, src, src_off, nbytes):
        _check(engine_lib().h264_amd_copy(
            self.h, C.c_void_p(dst.ptr.value + dst_off),
            C.c_void_p(src.ptr.value + src_off), nbytes), "copy")

    def sync(self):
        _check(engine_lib().h264_amd_engine_sync(self.h), "sync")

    def probe_read_ms(self, d_buf, length):
        """one pure streaming read of the buffer (the box's read ceiling)"""
        ms = C.c_float()
        _check(engine_lib().h264_amd_probe_read(self.h, d_buf.ptr, length,
                                                C.byref(ms)), "probe_read")
        return ms.value

    def last_kernel_ms(self, family):
        ms = C.c_float()
        _check(engine_lib().h264_amd_last_kernel_ms(
            self.h, family.encode(), C.byref(ms)), "last_kernel_ms")
        return ms.value

    # -- scan -----------------------------------------------------------
    def scan(self, d_buf, length, cap, keep=False):
        d_nals = self.alloc(cap * C.sizeof(Nal))
        cnt, consumed = C.c_size_t(), C.c_size_t()
        _check(engine_lib().h264_amd_scan(self.h, d_buf.ptr, length,
                                          d_nals.ptr, cap, C.byref(cnt),
                                          C.byref(consumed)), "scan")
        n = min(cnt.value, cap)
        tab = self.download(d_nals, n * C.sizeof(Nal), dtype=np.uint64)
        if keep:  # also hand back the device table (for gather_nal_heads)
            return tab.reshape(-1, 2), cnt.value, consumed.value, d_nals
        return tab.reshape(-1, 2), cnt.value, consumed.value

    def byte_stream_to_avcc(self, d_buf, length):
        """h264_byte_stream_to_avcc in place on a device buffer"""
        _check(engine_lib().h264_amd_byte_stream_to_avcc(self.h, d_buf.ptr,
                                                         length), "to_avcc")

    def gather_nal_heads(self, d_buf, d_nals, n, head_bytes=16):
        """[n, head_bytes] uint8: the first bytes of every NAL unit."""
        d_out = self.alloc(max(1, n * head_bytes))
        _check(engine_lib().h264_amd_gather_nal_heads(
            self.h, d_buf.ptr, d_nals.ptr, n, head_bytes, d_out.ptr),
            "gather_nal_heads")
        out = self.download(d_out, n * head_bytes)
        d_out.free()
        return out.reshape(n, head_bytes)

    # -- slice decode -----------------------------------------------------
    def decode_slices(self, params, d_buf, descs, reader_flags=1,
                      tier=TIER_NONE, mb_capacity=0, want_headers=False,
                      cat=1, d_descs=None, outputs=None, sync=True):
        """descs: numpy structured/array of (offset,u64; size,u32; epoch,u32)
        rows or a list of tuples.  Returns a dict with numpy views of the
        results (after sync)."""
        n = len(descs)
        if d_descs is None:
            arr = np.zeros(n, dtype=[("offset", "<u8"), ("size", "<u4"),
                                     ("epoch", "<u4")])
            for i, d in enumerate(descs):
                arr[i] = tuple(d)
            d_descs = self.upload(arr)
        stride = {TIER_NONE: 0, TIER_T0: 4, TIER_T1: 64,
                  TIER_T2: t2_stride(3)}[tier]
        out = outputs or {}
        if "results" not in out:
            out["results"] = self.alloc(n * C.sizeof(SliceResult))
        if want_headers and "headers" not in out:
            out["headers"] = self.alloc(n * C.sizeof(SliceHeader))
        if tier != TIER_NONE and "mb" not in out:
            out["mb"] = self.alloc(n * mb_capacity * stride)
        a = DecodeArgs()
        a.d_buf = d_buf.ptr
        a.d_descs = d_descs.ptr
        a.n = n
        a.reader_flags = reader_flags
        a.tier = tier
        a.mb_capacity = mb_capacity
        a.d_results = out["results"].ptr
        a.d_headers = out["headers"].ptr if want_headers else None
        a.d_mb_out = out["mb"].ptr if tier != TIER_NONE else None
        a.mb_stride = stride
        a.d_buf_len = d_buf.nbytes  # descriptors are checked against it
        _check(engine_lib().h264_amd_decode_slices(self.h, params.h,
                                                   C.byref(a)), "decode")
        out["stride"] = stride
        out["n"] = n
        out["mb_capacity"] = mb_capacity
        out["d_descs"] = d_descs
        if sync:
            self.sync()
        return out

    # -- slice write-back ---------------------------------------------------
    def write_slices(self, params, d_src, out, out_capacity, sync=True,
                     fetch=True, d_out=None):
        """Re-serialise the slices of a decode_slices() result `out` (needs
        want_headers=True).  Returns (d_out, offsets, sizes-as-int32 array);
        d_out: reuse an output buffer of an earlier call; fetch=False: leave
        the sizes on the device (timing loops)."""
        n = out["n"]
        cap = (int(out_capacity) + 15) & ~15
        offs = np.arange(n, dtype=np.uint64) * np.uint64(cap)
        d_offs = self.upload(offs)
        d_sizes = self.alloc(4 * n)
        if d_out is None:
            d_out = self.alloc(n * cap)
        a = WriteArgs()
        a.d_src = d_src.ptr
        a.d_descs = out["d_descs"].ptr
        a.d_results = out["results"].ptr
        a.d_headers = out["headers"].ptr
        a.n = n
        a.d_out = d_out.ptr
        a.d_out_offsets = d_offs.ptr
        a.d_out_sizes = d_sizes.ptr
        a.out_capacity = cap
        a.d_src_len = d_src.nbytes
        _check(engine_lib().h264_amd_write_slices(self.h, params.h,
                                                  C.byref(a)), "write")
        if sync:
            self.sync()
        sizes = self.download(d_sizes, 4 * n, dtype=np.int32) if fetch else \
            None
        return d_out, offs, sizes

    # -- synthetic slices ---------------------------------------------------
    def encode_synth(self, params, descs, headers, out_capacity):
        """descs: list of (kind, epoch, mb_count, nal_ref_idc, nal_unit_type);
        headers: list of SliceHeader.  Returns list of bytes (or int errno)."""
        n = len(descs)
        darr = (SynthDesc * n)()
        for i, d in enumerate(descs):
            darr[i] = SynthDesc(d[0], d[1], d[2], d[3], d[4], 0)
        harr = (SliceHeader * n)(*headers)
        d_descs = self.upload(np.frombuffer(bytes(darr), dtype=np.uint8))
        d_hdrs = self.upload(np.frombuffer(bytes(harr), dtype=np.uint8))
        cap = (int(out_capacity) + 15) & ~15
        offs = np.arange(n, dtype=np.uint64) * np.uint64(cap)
        d_offs = self.upload(offs)
        d_sizes = self.alloc(4 * n)
        d_out = self.alloc(n * cap)
        a = SynthArgs()
        a.d_descs = d_descs.ptr
        a.d_headers = d_hdrs.ptr
        a.n = n
        a.out_capacity = cap
        a.d_out = d_out.ptr
        a.d_out_offsets = d_offs.ptr
        a.d_out_sizes = d_sizes.ptr
        _check(engine_lib().h264_amd_encode_synth_slices(
            self.h, params.h, C.byref(a)), "synth")
        self.sync()
        sizes = self.download(d_sizes, 4 * n, dtype=np.int32)
        raw = self.download(d_out, n * cap)
        res = []
        for i in range(n):
            if sizes[i] < 0:
                res.append(int(sizes[i]))
            else:
                res.append(raw[i * cap:i * cap + int(sizes[i])].tobytes())
        return res

    def cabac_init_states(self, triples):
        t = np.ascontiguousarray(np.asarray(triples, dtype=np.int32))
        n = t.shape[0]
        d_t = self.upload(t)
        d_s = self.alloc(n * 2048)
        _check(engine_lib().h264_amd_cabac_init_states(self.h, d_t.ptr, n,
                                                       d_s.ptr), "cabac_init")
        self.sync()
        return self.download(d_s, n * 2048).reshape(n, 1024, 2)

    def fetch_results(self, out):
        raw = self.download(out["results"], out["n"] * C.sizeof(SliceResult))
        return (SliceResult * out["n"]).from_buffer_copy(raw.tobytes())

    def fetch_headers(self, out):
        raw = self.download(out["headers"], out["n"] * C.sizeof(SliceHeader))
        return (SliceHeader * out["n"]).from_buffer_copy(raw.tobytes())

    def fetch_mb(self, out, slice_index, mb_count):
        stride = out["stride"]
        off = slice_index * out["mb_capacity"] * stride
        raw = self.download(out["mb"], mb_count * stride, offset=off)
        return raw.reshape(mb_count, stride)
