"""Rank body of tests/test_dist_gloo.py: the N > 1 path of bench.py --workload
c4 (libh264_amd/pipeline.py + shard.py) on CPU processes, backend gloo.  The
ranks run the PRODUCT's host logic -- rank / device selection, the plan, the
parameter-set replay through the drop-in library's host parser and the
engine's snapshot table, the partition, the point-to-point scatter of real
slice bytes, the all-gather -- and only the two device operations (scan,
slice decode) are stood in for by the oracle, in its role as checker: there is
no GPU here."""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


class OracleOps:
    """CPU stand-in for pipeline.EngineOps: torch CPU tensors as buffers."""

    device = "cpu"

    def __init__(self, local_rank):
        import torch
        self.torch = torch
        self.local_rank = local_rank  # the ordinal an Engine would be made on
        self.ps, self.epochs_after = [], []
        self.bytes_received = 0

    def sync(self):
        pass

    def alloc(self, nbytes):
        return self.torch.zeros(int(nbytes) + 64, dtype=self.torch.uint8)

    def upload(self, array):
        t = self.torch.from_numpy(np.ascontiguousarray(array).copy())
        return self.torch.cat([t, self.torch.full((64,), 0xff,
                                                  dtype=self.torch.uint8)])

    def tensor(self, buf):
        return buf

    def scan_heads(self, stream, length, cap, head_bytes):
        import orc
        data = bytes(stream[:length].numpy())
        rec = orc.parse_stream(data, flags=0, want=0)
        tab = np.array([(n.start, n.end) for n in rec.nals], dtype=np.uint64)
        heads = np.zeros((len(tab), head_bytes), dtype=np.uint8)
        for i, (a, b) in enumerate(tab):
            chunk = data[int(a):min(int(b), int(a) + head_bytes)]
            heads[i, :len(chunk)] = np.frombuffer(chunk, dtype=np.uint8)
        return tab.reshape(-1, 2), len(tab), rec.consumed, heads

    def read_bytes(self, stream, off, n):
        return bytes(stream[off:off + n].numpy())

    def pack(self, stream, offs, sizes, loffs, total):
        out = self.torch.full((int(total) + 64,), 0xff, dtype=self.torch.uint8)
        for o, z, l in zip(offs, sizes, loffs):
            out[int(l):int(l) + int(z)] = stream[int(o):int(o) + int(z)]
        return out

    def note_plan(self, plan, epochs_after):
        self.ps, self.epochs_after = plan.ps_nals, epochs_after

    def decode(self, params, local, ldescs):
        import orc
        out = np.zeros((len(ldescs), 64), dtype=np.uint8)
        for j, d in enumerate(ldescs):
            # the parameter sets current at the slice's position
            k = sum(1 for e in self.epochs_after if e <= int(d["epoch"]))
            nal = bytes(local[int(d["offset"]):int(d["offset"]) +
                              int(d["size"])].numpy())
            rec = orc.parse_nalus([raw for _t, raw in self.ps[:k]] + [nal],
                                  want=0)
            out[j] = np.frombuffer(bytes(rec.results[0]), dtype=np.uint8)
        self.bytes_received += int(ldescs["size"].sum())
        return out


def main():
    import streams
    from libh264_amd import gen, pipeline, shard

    rank, world, local_rank = shard.rank_env(gpus=int(sys.argv[1]))
    dist = shard.init_dist("gloo", local_rank)
    assert (dist.get_rank(), dist.get_world_size()) == (rank, world)
    ops = OracleOps(local_rank)

    # rank 0 alone holds the stream: two picture sizes, so that slices must
    # bind to the parameter sets current at THEIR position on every rank
    data, stream = b"", None
    if rank == 0:
        a = streams.small_cfg()
        b = streams.small_cfg(width_mbs=5, height_mbs=4)
        nals = [gen.sps(a), gen.pps(a), gen.aud(0)]
        nals += [gen.slice_nal(a, streams.rich_slice(a, s, (2, 0, 1)[s % 3]))
                 for s in range(13)]
        nals += [gen.sps(b), gen.pps(b), gen.filler(30)]
        nals += [gen.slice_nal(b, streams.rich_slice(b, s, (2, 0)[s % 2]))
                 for s in range(13, 23)]
        data = streams.annexb(nals)
        stream = ops.upload(np.frombuffer(data, dtype=np.uint8))
    t0 = time.perf_counter()
    # min_per_rank 8: a 23-slice stream is spread over both ranks (the
    # product's default of 1024 keeps anything that small on one)
    table, plan = pipeline.run_step(dist, ops, stream, len(data), 256,
                                    min_per_rank=8)
    dt = time.perf_counter() - t0
    time.sleep(0.05 * (rank + 1))  # ranks finish at different times
    dt2 = time.perf_counter() - t0
    slowest = shard.max_over_ranks(dist, dt2)
    got_bytes, got_slices = shard.sum_over_ranks(
        dist, [ops.bytes_received, 0])
    ordinals = [None] * world
    dist.all_gather_object(ordinals, ops.local_rank)
    seeds = [None] * world
    dist.all_gather_object(seeds, shard.rank_seed(rank))
    received = [None] * world
    dist.all_gather_object(received, ops.bytes_received)
    if rank == 0:
        parts, used = pipeline.partition(plan.descs["size"], world, 8)
        assert used == plan.ranks_used == 2
        assert pipeline.partition(plan.descs["size"], world)[1] == 1
        print(json.dumps({
            "world": world, "slowest": slowest, "own": dt2, "step": dt,
            "slices": len(plan.descs),
            "bytes": int(plan.descs["size"].sum()),
            "bytes_decoded_all_ranks": got_bytes,
            "received": received,
            "loads": [int(plan.descs["size"][p].sum()) for p in parts],
            "max_slice": int(plan.descs["size"].max()),
            "table_sha": hashlib.sha256(table.tobytes()).hexdigest(),
            "stream_sha": hashlib.sha256(data).hexdigest(),
            "ordinals": ordinals, "seeds": seeds,
            "ps_nals": len(plan.ps_nals),
        }), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
