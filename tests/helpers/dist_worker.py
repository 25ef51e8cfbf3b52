"""Rank body of tests/test_dist_gloo.py: the multi-GPU host logic of bench.py /
libh264_amd.shard on CPU processes (backend gloo).  The per-rank "decode" is
the oracle (checker role: there is no GPU here); what is under test is the
partition, the collectives and the ordering."""
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


def main():
    import torch.distributed as dist
    import orc
    import streams
    from libh264_amd import gen, shard

    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()

    # every rank derives the same batch (strong mode: one batch, sharded)
    cfg = streams.small_cfg()
    slices = [streams.rich_slice(cfg, s, (2, 0, 1)[s % 3]) for s in range(23)]
    nals = [gen.slice_nal(cfg, s) for s in slices]
    sizes = [len(n) for n in nals]
    parts = shard.balance(sizes, world)
    mine = parts[rank]

    t0 = time.perf_counter()
    rec = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg)] +
                          [nals[i] for i in mine], want=0)
    local = np.frombuffer(bytes(rec.results), dtype=np.uint8).reshape(-1, 64)
    time.sleep(0.05 * (rank + 1))  # ranks finish at different times
    dt = time.perf_counter() - t0

    dist.barrier()
    slowest = shard.max_over_ranks(dist, dt)
    tot_bytes, tot_slices = shard.sum_over_ranks(
        dist, [sum(sizes[i] for i in mine), len(mine)])
    table = shard.gather_records(dist, local, mine, len(nals))

    # weak mode: distinct seeds per rank
    seeds = [None] * world
    dist.all_gather_object(seeds, shard.rank_seed(rank))

    if rank == 0:
        print(json.dumps({
            "world": world,
            "slowest": slowest,
            "own": dt,
            "bytes": tot_bytes,
            "slices": tot_slices,
            "loads": [int(sum(sizes[i] for i in p)) for p in parts],
            "max_slice": max(sizes),
            "table_sha": hashlib.sha256(table.tobytes()).hexdigest(),
            "seeds": seeds,
        }), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
