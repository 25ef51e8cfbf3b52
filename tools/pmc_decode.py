#!/usr/bin/env python3
"""Instruction mix of the decode kernel per macroblock from rocprofv3 PMC
passes (run on the GPU box):

    python tools/pmc_decode.py [--slices 1024] [--lib path/to/libengine_x.so]

Runs `rocprofv3 --kernel-trace --pmc ...` over tools/ab_decode.py (one batch of
bench slices, two launches) once per counter group and prints, per counter, the
value per macroblock summed over the chip.  SQ counters count per wave
instruction; one macroblock = one wave's work, so "per macroblock" is also "per
wave and macroblock"."""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [
    ["SQ_INSTS_SALU", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH",
     "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES"],
    ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
     "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
     "SQ_BUSY_CYCLES"],
]
KERNEL = "h264_amd_k_decode_slices"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slices", type=int, default=1024)
    ap.add_argument("--lib", default="")
    ap.add_argument("--out", default="/tmp/pmc_decode")
    ap.add_argument("--cabac", action="store_true",
                    help="the CABAC kernel on tools/bench_cabac.py's slices; "
                         "values per BIN instead of per macroblock")
    ap.add_argument("--counters", default="",
                    help="comma-separated counters of one extra pass instead "
                         "of the two standard groups (e.g. SQC_ICACHE_REQ,"
                         "SQC_ICACHE_HITS,SQC_ICACHE_MISSES)")
    ap.add_argument("--mix", default="",
                    help="a macroblock mix of tools/bench_mix.py instead of the "
                         "bench batch")
    args = ap.parse_args()
    env = dict(os.environ, TMPDIR="/tmp")
    if args.lib:
        env["H264_AMD_ENGINE_LIB"] = os.path.abspath(args.lib)
    res = {}
    groups = [args.counters.split(",")] if args.counters else GROUPS
    for gi, grp in enumerate(groups):
        d = "%s_%d" % (args.out, gi)
        subprocess.run(["rm", "-rf", d])
        prog = [os.path.join(ROOT, "tools", "ab_decode.py"), "--child",
                "--slices", str(args.slices), "--steps", "1"]
        if args.mix:
            prog = [os.path.join(ROOT, "tools", "bench_mix.py"), "--only",
                    args.mix, "--slices", str(args.slices), "--reps", "1"]
        if args.cabac:
            prog = [os.path.join(ROOT, "tools", "bench_cabac.py"), "--slices",
                    str(args.slices), "--reps", "1"]
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + grp + \
            ["--output-format", "csv", "-d", d, "--", sys.executable] + prog
        p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True)
        if p.returncode != 0:
            print(p.stdout[-2000:])
            raise SystemExit("rocprofv3 failed")
        n = {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"),
                           recursive=True):
            for row in csv.DictReader(open(f)):
                if not row["Kernel_Name"].startswith(
                        "h264_amd_k_decode_cabac" if args.cabac else KERNEL):
                    continue
                c = row["Counter_Name"]
                res[c] = res.get(c, 0.0) + float(row["Counter_Value"])
                n[c] = n.get(c, 0) + 1
        for c in grp:
            if c in res and n.get(c):
                res[c] /= n[c]  # per launch
    mbs = 8160.0 * args.slices
    out = {"slices": args.slices, "lib": os.path.basename(args.lib) or
           "in-tree", "mix": args.mix or "bench batch"}
    unit = "_per_mb"
    if args.cabac:
        # the bins of the batch: from the generator, as bench_cabac.py counts them
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_cabac
        mbs = float(bench_cabac.workload(args.slices, 16)[4])
        unit = "_per_bin"
        out["mix"] = "C3 CABAC P-slices"
    for c, v in res.items():
        out[c + unit] = round(v / mbs, 3 if args.cabac else 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
