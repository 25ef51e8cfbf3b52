#!/usr/bin/env python3
"""A/B timing of decode-kernel builds on the bench batch (BASELINE config 2).

    python tools/ab_decode.py [--slices 4096,512] [--steps 10] lib1.so lib2.so ...

Every engine library given (default: the in-tree build) decodes the same seeded
batch in a process of its own; a line of JSON per (library, batch size) carries
the median kernel time and a digest over all 64-byte slice results (status,
macroblock count, end bit, checksum over the full macroblock syntax), so that
two builds can be compared bit for bit as well as in time.  Variants are built
with tools/build_variant.sh.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(args):
    import bench
    import libh264_amd
    from libh264_amd import gen, Params, shard
    from libh264_amd.ctypes_abi import TIER_T0
    import orc
    cores = min(len(os.sched_getaffinity(0)), 16)
    eng = libh264_amd.Engine(0)
    for n in [int(x) for x in args.slices.split(",")]:
        cfg, buf, offs, sizes = bench.make_workload(n, 1, cores)
        rec = orc.parse_nalus([gen.sps(cfg), gen.pps(cfg)], flags=0, want=0)
        params = Params()
        params.put_sps(rec.sps[0])
        epoch = params.put_pps(rec.pps[0])
        order = (shard.longest_first(sizes) if args.host_sorted else
                 np.arange(n))
        descs = np.zeros(n, dtype=[("offset", "<u8"), ("size", "<u4"),
                                   ("epoch", "<u4")])
        descs["offset"], descs["size"], descs["epoch"] = \
            offs[order], sizes[order], epoch
        d_buf, d_descs = eng.upload(buf), eng.upload(descs)
        out = eng.decode_slices(params, d_buf, descs, reader_flags=1,
                                tier=TIER_T0, mb_capacity=8160,
                                d_descs=d_descs)
        ms = []
        for _ in range(args.steps):
            eng.decode_slices(params, d_buf, descs, reader_flags=1,
                              tier=TIER_T0, mb_capacity=8160, d_descs=d_descs,
                              outputs=out)
            ms.append(eng.last_kernel_ms("decode"))
        raw = eng.download(out["results"])
        res = eng.fetch_results(out)
        bad = sum(1 for i in range(n)
                  if res[i].status != 0 or res[i].mb_count != 8160)
        mbs = 8160.0 * n
        print(json.dumps({
            "lib": os.path.basename(os.environ.get("H264_AMD_ENGINE_LIB",
                                                   "in-tree")),
            "slices": n, "kernel_ms_median": round(float(np.median(ms)), 3),
            "kernel_ms_min": round(float(np.min(ms)), 3),
            "mmb_per_s": round(mbs / float(np.median(ms)) / 1e3, 1),
            "gbit_per_s": round(8e-6 * float(sizes.sum()) /
                                float(np.median(ms)), 2),
            "bad_slices": bad, "host_sorted": args.host_sorted,
            "engine_order": os.environ.get("H264_AMD_ORDER", "1") != "0",
            "results_sha1": hashlib.sha1(raw.tobytes()).hexdigest()[:16],
        }), flush=True)
        del d_buf, d_descs, out
    eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slices", default="4096")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--host-sorted", action="store_true",
                    help="descriptors longest first already on the host (the "
                         "engine orders its launches itself, H264_AMD_ORDER=0 "
                         "turns that off)")
    ap.add_argument("libs", nargs="*")
    args = ap.parse_args()
    if args.child:
        return child(args)
    for lib in args.libs or [""]:
        env = dict(os.environ)
        if lib:
            env["H264_AMD_ENGINE_LIB"] = os.path.abspath(lib)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child",
                        "--slices", args.slices, "--steps", str(args.steps)] +
                       (["--host-sorted"] if args.host_sorted else []),
                       env=env, check=False)


if __name__ == "__main__":
    main()
