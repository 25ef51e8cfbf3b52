#!/usr/bin/env python3
"""Copies the summaries of one tools/collect_profiles.sh run from
gpurun_out/<tag>/ into profiles/rNN_* and rebuilds the two history files.

    python tools/publish_profiles.py r02f r02
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json(path):
    lines = [x for x in open(path) if x.startswith("{")]
    return lines[-1] if lines else None


def main():
    tag, rnd = sys.argv[1], sys.argv[2]
    g = os.path.join(ROOT, "gpurun_out", tag)
    p = os.path.join(ROOT, "profiles")
    for src, dst in [("bench_kernel_stats.csv", "bench_kernel_stats.csv"),
                     ("side_kernels_stats.csv", "side_kernels_stats.csv"),
                     ("decode_hbm_traffic.json", "decode_hbm_traffic.json"),
                     ("issue_probe.log", "issue_probe.txt"),
                     ("mix.log", "decode_mix.jsonl"),
                     ("dropin_per_nal.json", "dropin_per_nal.json"),
                     ("cabac_bench.json", "cabac_bench.json"),
                     ("cabac_kernel_stats.csv", "cabac_kernel_stats.csv"),
                     ("cabac_hbm_traffic.json", "cabac_hbm_traffic.json"),
                     ("pmc_cabac.log", "cabac_pmc.json"),
                     ("bench_c1.json", "bench_c1.json"),
                     ("bench_c4_1gib.json", "bench_c4_1gib.json"),
                     ("phase_512.log", "decode_phase_timers_final.txt")]:
        if os.path.exists(os.path.join(g, src)):
            shutil.copy(os.path.join(g, src),
                        os.path.join(p, "%s_%s" % (rnd, dst)))
    line = last_json(os.path.join(g, "bench.json"))
    open(os.path.join(p, "%s_bench.json" % rnd), "w").write(line)
    d = json.loads(line)
    s = d["side"]
    print("bench %.1f Mbit/s, %.3f ms/step, kernel %.3f ms, %.0f MB/s, frac "
          "%.5f; cpu %.1f Mbit/s %.0f MB/s; ceiling %.1f, scan %.1f (%.4f), "
          "write %.1f (%.4f), c4 %.3f ms, cabac dec %.4f enc %.4f" %
          (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"],
           d["macroblocks_per_s"], d["roofline"]["frac"],
           d["cpu_baseline"]["value"], d["cpu_baseline"]["mb_per_s"],
           s["read_ceiling"]["GBps"], s["scan_1gib"]["achieved_GBps"],
           s["scan_1gib"]["frac_measured"], s["write_back"]["achieved_GBps"],
           s["write_back"]["frac_measured"], s["c4_64mib"]["ms_per_step"],
           s["cabac_skip_decode"]["ms"], s["cabac_encode"]["ms"]))
    pm = json.loads(last_json(os.path.join(g, "pmc_decode.log")))
    print("per MB: SALU %.0f VALU %.0f branch %.0f LDS %.1f" %
          (pm["SQ_INSTS_SALU_per_mb"], pm["SQ_INSTS_VALU_per_mb"],
           pm["SQ_INSTS_BRANCH_per_mb"], pm["SQ_INSTS_LDS_per_mb"]))
    for x in open(os.path.join(g, "ab.log")):
        if x.startswith("{"):
            a = json.loads(x)
            print("ab", a["slices"], a["kernel_ms_median"])
    print(open(os.path.join(p, "%s_bench_kernel_stats.csv" % rnd))
          .read().split("\n")[1][:90])


if __name__ == "__main__":
    main()
