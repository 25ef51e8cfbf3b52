"""N > 1 host logic on CPU: world_size 2 over gloo (torch.distributed.run), the
same helpers bench.py uses with nccl/RCCL on the GPU box."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

import orc
import streams
from libh264_amd import gen, shard

ROOT = orc.ROOT


def test_balance_properties():
    rnd = np.random.default_rng(1)
    for world in (1, 2, 3, 8):
        sizes = rnd.integers(1, 300000, size=200)
        parts = shard.balance(sizes, world)
        allidx = np.concatenate(parts)
        assert sorted(allidx.tolist()) == list(range(200))
        loads = [int(sizes[p].sum()) for p in parts]
        assert max(loads) - min(loads) <= int(sizes.max())
    assert [p.tolist() for p in shard.balance([5], 4)] == [[0], [], [], []]
    order = shard.longest_first([3, 9, 1, 9])
    assert order.tolist() == [1, 3, 0, 2]


def test_gather_single_process():
    rec = np.arange(12, dtype=np.uint8).reshape(3, 4)
    out = shard.gather_records(None, rec, [2, 0, 1], 3)
    assert out[2].tolist() == [0, 1, 2, 3] and out[0].tolist() == [4, 5, 6, 7]


def test_world_size_2_gloo():
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    env.pop("RANK", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29571",
           os.path.join(ROOT, "tests", "helpers", "dist_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    got = json.loads(line)
    assert got["world"] == 2 and got["slices"] == 23

    # the same batch in one process
    cfg = streams.small_cfg()
    nals = [gen.slice_nal(cfg, streams.rich_slice(cfg, s, (2, 0, 1)[s % 3]))
            for s in range(23)]
    ref = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg)] + nals, want=0)
    table = np.frombuffer(bytes(ref.results), dtype=np.uint8)
    assert got["table_sha"] == hashlib.sha256(table.tobytes()).hexdigest()
    assert got["bytes"] == sum(len(n) for n in nals)
    assert abs(got["loads"][0] - got["loads"][1]) <= got["max_slice"]
    # the job is as slow as its slowest rank (rank 1 sleeps longer)
    assert got["slowest"] >= got["own"]
    assert len(set(got["seeds"])) == 2
