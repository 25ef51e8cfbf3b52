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


def test_rank_env():
    assert shard.rank_env({}, gpus=1) == (0, 1, 0)
    assert shard.rank_env({"WORLD_SIZE": "4", "RANK": "2",
                           "LOCAL_RANK": "2"}, gpus=4) == (2, 4, 2)
    import pytest
    with pytest.raises(RuntimeError):
        shard.rank_env({"WORLD_SIZE": "2", "RANK": "0"}, gpus=8)
    with pytest.raises(RuntimeError):
        shard.rank_env({"WORLD_SIZE": "2", "RANK": "5"})


def test_bench_spawns_ranks_itself(monkeypatch):
    """`python bench.py --gpus N` without a launcher starts N ranks with
    torch.distributed.run before anything touches a GPU (the command is
    checked, not run: there is no GPU here)."""
    import bench
    seen = {}
    monkeypatch.setattr(bench.subprocess, "call",
                        lambda cmd, env=None: seen.update(cmd=cmd) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps",
                                      "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    import pytest
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert "torch" not in [m for m in ("torch",) if m in sys.modules and
                           getattr(sys.modules[m], "cuda", None) and
                           sys.modules[m].cuda.is_initialized()]


def test_world_size_2_gloo():
    """One stream on rank 0 -> plan -> broadcast -> point-to-point scatter of
    the slice bytes -> decode on both ranks -> all-gather: the table must be
    the single-process one, byte for byte."""
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29571",
           os.path.join(ROOT, "tests", "helpers", "dist_worker.py"), "2"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    got = json.loads(line)
    assert got["world"] == 2 and got["slices"] == 23 and got["ps_nals"] == 4

    # the same stream in one process
    a = streams.small_cfg()
    b = streams.small_cfg(width_mbs=5, height_mbs=4)
    nals = [gen.sps(a), gen.pps(a), gen.aud(0)]
    nals += [gen.slice_nal(a, streams.rich_slice(a, s, (2, 0, 1)[s % 3]))
             for s in range(13)]
    nals += [gen.sps(b), gen.pps(b), gen.filler(30)]
    nals += [gen.slice_nal(b, streams.rich_slice(b, s, (2, 0)[s % 2]))
             for s in range(13, 23)]
    data = streams.annexb(nals)
    assert got["stream_sha"] == hashlib.sha256(data).hexdigest()
    ref = orc.parse_stream(data, want=0)
    assert all(r.status == 0 for r in ref.results)
    table = np.frombuffer(bytes(ref.results), dtype=np.uint8)
    assert got["table_sha"] == hashlib.sha256(table.tobytes()).hexdigest()
    slice_bytes = sum(len(n) for n in nals if n[0] & 0x1f in (1, 5))
    assert got["bytes"] == slice_bytes
    # every slice byte was decoded by exactly one rank, and rank 1's share
    # reached it over the wire
    assert got["bytes_decoded_all_ranks"] == slice_bytes
    assert got["received"] == got["loads"] and min(got["loads"]) > 0
    assert abs(got["loads"][0] - got["loads"][1]) <= got["max_slice"]
    # one engine ordinal per rank, taken from LOCAL_RANK
    assert got["ordinals"] == [0, 1]
    # the job is as slow as its slowest rank (rank 1 sleeps longer)
    assert got["slowest"] >= got["own"]
    assert len(set(got["seeds"])) == 2
