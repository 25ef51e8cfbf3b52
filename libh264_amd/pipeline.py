"""BASELINE config 4 as a pipeline over one or N GPUs:

  A  the rank that holds the Annex-B byte stream scans it (device) and reads
     the head of every NAL unit back;
  B  its host parses the parameter sets in stream order (the drop-in library's
     own host parser) and lists the slice NAL units with the parameter-set
     epoch current at their position;
  C  that plan -- parameter-set NAL bytes in order + slice descriptors, KBs --
     is broadcast; every rank replays the parameter sets into its own snapshot
     table (same order => same epochs) and computes the same size-balanced
     partition (shard.balance);
  D  the slice bytes travel: one packed buffer per peer, grouped send / recv
     (point-to-point: on xGMI every peer has its own link, a ring collective
     would be bound by one);
  E  every rank decodes its slices;
  F  the 64-byte result records are all-gathered into stream order.

The device work goes through an `ops` object (Engine-backed on the GPU box, see
EngineOps; the CPU tests plug in the oracle as the checker's stand-in), the
traffic through torch.distributed (backend nccl = RCCL, or gloo).  With one
rank (dist None) stages C, D and F degenerate to local copies.
"""
import ctypes as C
import os
import time

import numpy as np

from . import shard

DESC_DTYPE = np.dtype([("offset", "<u8"), ("size", "<u4"), ("epoch", "<u4")])
ALIGN = 16  # pieces of the gather kernel (h264_amd_pack) start 16-byte aligned
MIN_SLICES_PER_RANK = 1024  # a quarter of one chip's wave slots


def partition(sizes, world, min_per_rank=MIN_SLICES_PER_RANK):
    """The engine's own plan (h264_amd_group_plan_ex, the code the in-process
    engine group runs): ranks that get work -- min(world, ceil(n /
    min_per_rank)), a stream that fills less than a quarter of one chip is not
    scattered -- and the size-balanced assignment, longest slice first to the
    least loaded rank.  Returns (list of index arrays, one per rank of the
    WORLD, ascending; ranks_used)."""
    from . import capi
    sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
    n = len(sizes)
    if n == 0:
        return [np.zeros(0, dtype=np.int64) for _ in range(world)], 0
    L = capi.engine_lib()
    L.h264_amd_group_plan_ex.restype = C.c_uint32
    L.h264_amd_group_plan_ex.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_void_p]
    owner = np.zeros(n, dtype=np.uint32)
    used = L.h264_amd_group_plan_ex(sizes.ctypes.data, n, world, min_per_rank,
                                    owner.ctypes.data)
    return [np.flatnonzero(owner == r).astype(np.int64)
            for r in range(world)], int(used)


class Plan:
    """What stage B produces and stage C broadcasts."""

    def __init__(self, ps_nals, descs, n_nals, consumed):
        self.ps_nals = ps_nals  # [(nal_unit_type, bytes)] in stream order
        self.descs = descs  # DESC_DTYPE array, stream order
        self.n_nals = n_nals
        self.consumed = consumed
        self.ranks_used = 1  # set by run_step: ranks that got slices


def _host_lib():
    from . import capi  # noqa: F401  (engine library first: one HIP runtime)
    from .ctypes_abi import Sps, Pps
    here = os.path.dirname(os.path.abspath(__file__))
    capi.engine_lib()
    H = C.CDLL(os.path.join(here, "lib", "libh264.so"))
    H.h264_parse_sps.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Sps)]
    H.h264_parse_pps.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Sps),
                                 C.POINTER(Pps)]
    return H


class ParamReplay:
    """Parameter-set NAL units -> h264_amd_params snapshots, in stream order.
    Every rank runs this on the same list and ends with the same epochs."""

    def __init__(self):
        from . import Params
        from .ctypes_abi import Sps
        self.H = _host_lib()
        self.params = Params()
        self.sps = Sps()

    def put(self, nal_type, raw):
        from .ctypes_abi import Sps, Pps
        if nal_type == 7:
            sps = Sps()
            if self.H.h264_parse_sps(raw, len(raw), C.byref(sps)) == 0:
                self.sps = sps
                self.params.put_sps(sps)
        elif nal_type == 8:
            pps = Pps()
            if self.H.h264_parse_pps(raw, len(raw), C.byref(self.sps),
                                     C.byref(pps)) == 0:
                self.params.put_pps(pps)
        return self.params.epoch


def make_plan(ops, stream, length, cap):
    """Stages A + B on the rank that holds the stream."""
    tab, count, consumed, heads = ops.scan_heads(stream, length, cap, 64)
    replay = ParamReplay()
    ps_nals, rows = [], []
    for i in range(count):
        t = int(heads[i, 0]) & 0x1f
        if t not in (1, 5, 7, 8):
            continue
        a, b = int(tab[i, 0]), int(tab[i, 1])
        if t in (7, 8):
            raw = heads[i, :b - a].tobytes() if b - a <= heads.shape[1] else \
                ops.read_bytes(stream, a, b - a)
            ps_nals.append((t, raw))
            replay.put(t, raw)
        else:
            rows.append((a, b - a, replay.params.epoch))
    descs = np.array(rows, dtype=DESC_DTYPE) if rows else \
        np.zeros(0, dtype=DESC_DTYPE)
    return Plan(ps_nals, descs, count, consumed), replay.params


def broadcast_plan(dist, plan, src=0):
    """Stage C.  Small: pickled object broadcast."""
    if dist is None:
        return plan
    box = [plan if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def local_layout(sizes):
    """Packed layout of a rank's slices (offsets, total bytes)."""
    sizes = np.asarray(sizes, dtype=np.int64)
    padded = (sizes + ALIGN - 1) // ALIGN * ALIGN
    offs = np.concatenate([[0], np.cumsum(padded)[:-1]]) if len(sizes) else \
        np.zeros(0, dtype=np.int64)
    return offs.astype(np.int64), int(padded.sum())


def scatter_slices(dist, ops, stream, plan, parts, src=0):
    """Stage D.  Returns (buffer, descriptors) this rank decodes from.  The
    rank that holds the stream decodes its own part in place; every other rank
    receives its slices packed back to back."""
    rank = 0 if dist is None else dist.get_rank()
    world = 1 if dist is None else dist.get_world_size()
    mine = parts[rank]
    if rank == src:
        if world > 1:
            import torch.distributed as td
            ops_list, keep = [], []
            for r in range(world):
                if r == src or len(parts[r]) == 0:
                    continue
                idx = parts[r]
                o, tot = local_layout(plan.descs["size"][idx])
                b = ops.pack(stream, plan.descs["offset"][idx],
                             plan.descs["size"][idx], o, tot)
                b = ops.tensor(b)
                keep.append(b)
                ops_list.append(td.P2POp(td.isend, b, r))
            if ops_list:
                for q in td.batch_isend_irecv(ops_list):
                    q.wait()
        return stream, plan.descs[mine].copy()
    loffs, total = local_layout(plan.descs["size"][mine])
    ldescs = np.zeros(len(mine), dtype=DESC_DTYPE)
    ldescs["offset"] = loffs
    ldescs["size"] = plan.descs["size"][mine]
    ldescs["epoch"] = plan.descs["epoch"][mine]
    local = ops.alloc(total)
    if len(mine):
        import torch.distributed as td
        view = ops.tensor(local)
        for q in td.batch_isend_irecv([td.P2POp(td.irecv, view, src)]):
            q.wait()
        if hasattr(ops, "received"):
            ops.received(local, view)
    return local, ldescs


def run_step(dist, ops, stream, length, cap, src=0, timing=None,
             min_per_rank=MIN_SLICES_PER_RANK):
    """One pass of the whole pipeline.  Returns (result table in stream order
    as uint8 [n, 64], plan).  `timing`: dict receiving wall seconds per stage
    (device work synchronised at the stage boundaries when given)."""
    rank = 0 if dist is None else dist.get_rank()
    world = 1 if dist is None else dist.get_world_size()

    def mark(name, t0):
        if timing is not None:
            ops.sync()
            timing[name] = timing.get(name, 0.0) + time.perf_counter() - t0
        return time.perf_counter()

    t = time.perf_counter()
    plan = None
    if rank == src:
        plan, _params = make_plan(ops, stream, length, cap)
    t = mark("scan_plan", t)
    plan = broadcast_plan(dist, plan, src)
    replay = ParamReplay()
    epochs_after = [replay.put(kind, raw) for kind, raw in plan.ps_nals]
    if hasattr(ops, "note_plan"):
        ops.note_plan(plan, epochs_after)
    parts, plan.ranks_used = partition(plan.descs["size"], world, min_per_rank)
    t = mark("broadcast", t)
    local, ldescs = scatter_slices(dist, ops, stream, plan, parts, src)
    t = mark("scatter", t)
    res = ops.decode(replay.params, local, ldescs)
    t = mark("decode", t)
    table = shard.gather_records(dist, res, parts[rank], len(plan.descs),
                                 device=ops.device)
    mark("gather", t)
    return table, plan


class EngineOpsPlain:
    """The device side of a one-GPU run on an existing Engine, no torch."""

    device = "cpu"  # gather_records with dist None never looks at it

    def __init__(self, eng, mb_capacity=8160):
        self.eng = eng
        self.mb_capacity = mb_capacity
        self.last_decode_ms = 0.0
        self._out = None

    def sync(self):
        self.eng.sync()

    def scan_heads(self, stream, length, cap, head_bytes):
        tab, count, consumed, d_nals = self.eng.scan(stream, length, cap,
                                                     keep=True)
        heads = self.eng.gather_nal_heads(stream, d_nals, count, head_bytes)
        d_nals.free()
        return tab, count, consumed, heads

    def read_bytes(self, stream, off, n):
        return self.eng.download(stream, n, offset=off).tobytes()

    def decode(self, params, local, ldescs):
        from .ctypes_abi import TIER_T0
        n = len(ldescs)
        if n == 0:
            return np.zeros((0, 64), dtype=np.uint8)
        key = (n, ldescs.tobytes())
        if getattr(self, "_desc_key", None) != key:
            self._desc_key = key
            self._d_descs = self.eng.upload(ldescs)  # once per plan
            self._out = None
        out = self.eng.decode_slices(params, local, ldescs, reader_flags=1,
                                     tier=TIER_T0, mb_capacity=self.mb_capacity,
                                     d_descs=self._d_descs, outputs=self._out)
        self._out = out
        self.last_decode_ms = self.eng.last_kernel_ms("decode")
        return self.eng.download(out["results"], n * 64).reshape(n, 64)


class EngineOps:
    """The device side on a GPU: libh264_amd.Engine on torch's current stream,
    buffers owned by torch (so that RCCL can send them)."""

    def __init__(self, local_rank, mb_capacity=8160, host_staging=False):
        import torch
        from . import Engine
        self.torch = torch
        self.gpu = torch.device("cuda", local_rank)
        # host_staging: the transport cannot take device memory (gloo, the
        # one-GPU rehearsal of the N > 1 path): traffic goes through the host
        self.staging = host_staging
        self.device = torch.device("cpu") if host_staging else self.gpu
        torch.cuda.set_device(local_rank)
        self.eng = Engine(local_rank,
                          stream=torch.cuda.current_stream().cuda_stream)
        self.mb_capacity = mb_capacity
        self.last_decode_ms = 0.0
        self._out = None
        self._desc_key = None
        self._pack_cache = {}

    def sync(self):
        self.eng.sync()
        self.torch.cuda.synchronize()

    def alloc(self, nbytes):
        return self.torch.empty(int(nbytes) + 64, dtype=self.torch.uint8,
                                device=self.gpu)

    def upload(self, array):
        t = self.torch.from_numpy(np.array(array, copy=True))
        buf = self.alloc(t.numel())
        buf[:t.numel()].copy_(t)
        buf[t.numel():].fill_(0xff)
        return buf

    def tensor(self, buf):
        return buf.cpu() if self.staging else buf

    def received(self, buf, view):
        if self.staging:
            buf.copy_(view)

    def _ext(self, buf):
        from .capi import ExtBuf
        return ExtBuf(buf.data_ptr(), buf.numel())

    def scan_heads(self, stream, length, cap, head_bytes):
        tab, count, consumed, d_nals = self.eng.scan(self._ext(stream), length,
                                                     cap, keep=True)
        heads = self.eng.gather_nal_heads(self._ext(stream), d_nals, count,
                                          head_bytes)
        d_nals.free()
        return tab, count, consumed, heads

    def read_bytes(self, stream, off, n):
        return bytes(stream[off:off + n].cpu().numpy())

    def pack(self, stream, offs, sizes, loffs, total):
        """slices of the stream -> one packed buffer, by the engine's gather
        kernel (h264_amd_pack); the index arrays are uploaded once per plan"""
        if len(offs) == 0:
            return self.alloc(0)
        key = (len(offs), int(np.asarray(offs, dtype=np.uint64).sum()),
               int(total))
        idx = self._pack_cache.get(key)
        if idx is None:
            idx = (self.eng.upload(np.ascontiguousarray(offs, dtype=np.uint64)),
                   self.eng.upload(np.ascontiguousarray(loffs, dtype=np.uint64)),
                   self.eng.upload(np.ascontiguousarray(sizes, dtype=np.uint32)))
            self._pack_cache = {key: idx}
        out = self.alloc(total)
        from .capi import engine_lib, _check
        L = engine_lib()
        L.h264_amd_pack.argtypes = [C.c_void_p] * 6 + [C.c_uint32]
        _check(L.h264_amd_pack(self.eng.h, out.data_ptr(), stream.data_ptr(),
                               idx[0].ptr, idx[1].ptr, idx[2].ptr, len(offs)),
               "pack")
        self.eng.sync()
        return out

    def decode(self, params, local, ldescs):
        from .ctypes_abi import TIER_T0
        from .capi import ExtBuf
        n = len(ldescs)
        if n == 0:
            return np.zeros((0, 64), dtype=np.uint8)
        # descriptors go up once per plan, result records stay on the device
        # (a torch tensor, so that RCCL can all-gather them from where they are)
        key = (n, ldescs.tobytes())
        if self._desc_key != key:
            self._desc_key = key
            self._d_descs = self.eng.upload(ldescs)
            self._res = self.torch.empty(n * 64, dtype=self.torch.uint8,
                                         device=self.gpu)
            self._out = None
        reuse = self._out
        if reuse is None:
            reuse = {"results": ExtBuf(self._res.data_ptr(), n * 64)}
        out = self.eng.decode_slices(params, self._ext(local), ldescs,
                                     reader_flags=1, tier=TIER_T0,
                                     mb_capacity=self.mb_capacity,
                                     d_descs=self._d_descs, outputs=reuse)
        self._out = out
        self.last_decode_ms = self.eng.last_kernel_ms("decode")
        if self.staging:
            return self._res.cpu().numpy().reshape(n, 64)
        return self._res.view(n, 64)

