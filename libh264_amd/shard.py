"""Multi-GPU sharding of a slice batch.

Slices are independent entropy-coding units, so a batch shards by slice with
no collective on the data path: every rank decodes its own slices on its own
GPU.  What crosses ranks is control-plane sized: the max-over-ranks wall time,
byte / macroblock totals, and -- when a caller wants the whole batch's
per-slice results in one place -- an all-gather of the 64-byte result records.
`torch.distributed` is the transport (backend "nccl" = RCCL on the GPU box,
"gloo" in the CPU tests); nothing here touches HIP.
"""
import heapq

import numpy as np


def rank_seed(rank, base=17):
    """Seed of rank's own synthetic batch (weak scaling: every rank generates
    and decodes the same number of slices, all different)."""
    return 1000003 * rank + base


def rank_env(env=None, gpus=None):
    """(rank, world, local_rank) of this process from the launcher's
    environment (torch.distributed.run sets RANK / WORLD_SIZE / LOCAL_RANK).
    `gpus`: what the command line asked for; a launcher that started another
    number of ranks is a mistake worth stopping for."""
    import os
    env = os.environ if env is None else env
    world = int(env.get("WORLD_SIZE", "1"))
    rank = int(env.get("RANK", "0"))
    local_rank = int(env.get("LOCAL_RANK", str(rank)))
    if not (0 <= rank < world):
        raise RuntimeError("RANK %d outside WORLD_SIZE %d" % (rank, world))
    if gpus is not None and "WORLD_SIZE" in env and int(gpus) != world:
        raise RuntimeError("--gpus %d but the launcher started %d ranks" %
                           (int(gpus), world))
    return rank, world, local_rank


def init_dist(backend, local_rank, env=None):
    """Join the job's process group.  ORDER MATTERS on the GPU box: torch ships
    its own ROCm runtime under the same soname as the system one the engine
    library was linked against; with torch initialised first the engine
    resolves to that already-loaded copy and the process has ONE HIP runtime
    (measured both ways, tools/dist_probe.py).  So this runs before the first
    libh264_amd.Engine() of the process; the engine library loads lazily."""
    import os
    import torch.distributed as dist
    env = os.environ if env is None else env
    if "MASTER_ADDR" not in env:
        env["MASTER_ADDR"] = "127.0.0.1"
        env.setdefault("MASTER_PORT", "29573")
        env.setdefault("RANK", "0")
        env.setdefault("WORLD_SIZE", "1")
    if backend == "nccl":
        import torch
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl",
                                device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=backend)
    return dist


def balance(sizes, world):
    """Greedy longest-first partition of slices over `world` ranks by bytes.
    Returns a list of index arrays (ascending inside each rank), covering
    every slice exactly once; rank loads differ by at most the largest slice.
    """
    sizes = np.asarray(sizes, dtype=np.int64)
    order = np.argsort(-sizes, kind="stable")
    heap = [(0, r) for r in range(world)]
    heapq.heapify(heap)
    parts = [[] for _ in range(world)]
    for i in order:
        load, r = heapq.heappop(heap)
        parts[r].append(int(i))
        heapq.heappush(heap, (load + int(sizes[i]), r))
    return [np.array(sorted(p), dtype=np.int64) for p in parts]


def longest_first(sizes):
    """Launch order inside one rank: largest slices first, so that the last
    wavefronts to start are the shortest."""
    return np.argsort(-np.asarray(sizes, dtype=np.int64), kind="stable")


def max_over_ranks(dist, seconds, device="cpu"):
    """The job's step time is the slowest rank's."""
    if dist is None:
        return float(seconds)
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, values, device="cpu"):
    if dist is None:
        return [float(v) for v in values]
    import torch
    t = torch.tensor([float(v) for v in values], dtype=torch.float64,
                     device=device)
    dist.all_reduce(t)
    return [float(v) for v in t.tolist()]


def gather_records(dist, records, index, total, device="cpu"):
    """All-gather fixed-size per-slice records (uint8 array [n_local, width])
    into the batch's original order.  `index` holds the global slice index of
    every local record (from balance()).  Ranks may hold different counts."""
    on_device = not isinstance(records, np.ndarray)  # a torch tensor
    if on_device and dist is None:
        records = records.cpu().numpy()
        on_device = False
    if not on_device:
        records = np.ascontiguousarray(records, dtype=np.uint8)
    width = records.shape[1]
    if dist is None:
        out = np.zeros((total, width), dtype=np.uint8)
        out[np.asarray(index)] = records
        return out
    import torch
    world = dist.get_world_size()
    n_local = torch.tensor([records.shape[0]], dtype=torch.int64,
                           device=device)
    counts = [torch.zeros(1, dtype=torch.int64, device=device)
              for _ in range(world)]
    dist.all_gather(counts, n_local)
    counts = [int(c.item()) for c in counts]
    cap = max(counts) if counts else 0
    pad_rec = torch.zeros((cap, width), dtype=torch.uint8, device=device)
    pad_idx = torch.full((cap,), -1, dtype=torch.int64, device=device)
    if records.shape[0]:
        # records decoded on the GPU stay there: no host round trip in front
        # of the all-gather
        pad_rec[:records.shape[0]] = records if on_device else \
            torch.from_numpy(records).to(device)
        pad_idx[:records.shape[0]] = torch.from_numpy(
            np.asarray(index, dtype=np.int64)).to(device)
    all_rec = [torch.zeros_like(pad_rec) for _ in range(world)]
    all_idx = [torch.zeros_like(pad_idx) for _ in range(world)]
    dist.all_gather(all_rec, pad_rec)
    dist.all_gather(all_idx, pad_idx)
    out = np.zeros((total, width), dtype=np.uint8)
    seen = np.zeros(total, dtype=bool)
    for r in range(world):
        idx = all_idx[r][:counts[r]].cpu().numpy()
        out[idx] = all_rec[r][:counts[r]].cpu().numpy()
        if seen[idx].any():
            raise RuntimeError("slice assigned to two ranks")
        seen[idx] = True
    if not seen.all():
        raise RuntimeError("%d slices assigned to no rank" % (~seen).sum())
    return out
