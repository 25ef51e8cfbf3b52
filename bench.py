#!/usr/bin/env python3
"""bench.py -- BASELINE.json headline metric on MI355X.

Workload (config.workload "C2"): a batch of independent 1080p Baseline-profile
CAVLC I-slices (1920x1088, 8160 macroblocks each), one slice per wavefront,
decoded by the hand-written gfx950 kernel h264_amd_k_decode_slices through the
C-ABI (h264_amd_decode_slices).  A "step" is one launch over the whole batch:
slice header + complete CAVLC slice_data walk of every slice, T0 macroblock
records and the per-slice checksum over the full macroblock syntax written
back.  Inputs are resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--slices S]

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), the batch
is sharded by slice with no data-path collective (slices are independent,
SURVEY.md 8(e)); weak scaling: every rank decodes S slices.  value = bits of
all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0).  roofline.achieved = slice bytes read per launch
/ average kernel time from HIP events on the launch stream; cpu_baseline = the
oracle's reader (kind "port") on a bounded sample, all host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def make_workload(n_slices, seed0, threads):
    from libh264_amd import gen
    from libh264_amd.ctypes_abi import GenSlice
    cfg = gen.cfg_1080p()
    slices = []
    for i in range(n_slices):
        s = GenSlice(slice_type=7, idr=1, nal_ref_idc=3, first_mb=0,
                     num_mbs=8160, frame_num=0, idr_pic_id=(seed0 + i) & 0xff,
                     pm_i16x16=280, pm_pcm=2, density=5, pm_escape=5,
                     pm_cbp_zero=50, pm_zero_pcm=100, seed=seed0 + i)
        slices.append(s)
    # in chunks: the generator reserves a worst-case slot per slice (all I_PCM,
    # 13 MB at 1080p) before it compacts, 53 GB of address space for 4096
    # slices at once -- times eight ranks on one host
    bufs, offs, sizes, pos = [], [], [], 0
    for k in range(0, n_slices, 512):
        b, o, z = gen.batch(cfg, slices[k:k + 512], threads=threads)
        pad = (-len(b)) % 16
        bufs.append(np.concatenate([b, np.zeros(pad, dtype=np.uint8)]))
        offs.append(o + np.uint64(pos))
        sizes.append(z)
        pos += len(b) + pad
    return (cfg, np.concatenate(bufs), np.concatenate(offs),
            np.concatenate(sizes))


def run_c4(args, eng, dist, rank, world, cores):
    """BASELINE config 4: a mixed Annex-B byte stream resident in HBM goes
    through the whole device-side pipeline per step -- scan, NAL heads to the
    host, parameter sets parsed by the drop-in library's host code, slice
    descriptors built, every slice decoded in one launch.  N > 1: every rank
    has its own stream (weak scaling, no data-path collective)."""
    import libh264_amd
    from libh264_amd import Params, shard
    from libh264_amd.ctypes_abi import Sps, Pps, TIER_T0
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_kernels
    H = C.CDLL(os.path.join(ROOT, "libh264_amd", "lib", "libh264.so"))
    H.h264_parse_sps.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Sps)]
    H.h264_parse_pps.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Sps),
                                 C.POINTER(Pps)]
    t0 = time.time()
    data = bench_kernels.mixed_stream(args.mib << 20, cores,
                                      seed0=shard.rank_seed(rank))
    gen_s = time.time() - t0
    d_buf = eng.upload(np.frombuffer(data + b"\xff" * 64, dtype=np.uint8))
    cap = len(data) // 4096 + 1024

    def step(out=None):
        tab, count, consumed, d_nals = eng.scan(d_buf, len(data), cap,
                                                keep=True)
        heads = eng.gather_nal_heads(d_buf, d_nals, count, 64)
        d_nals.free()
        params = Params()
        sps = Sps()
        descs = []
        for i in range(count):
            t = int(heads[i, 0]) & 0x1f
            if t not in (1, 5, 7, 8):
                continue
            a, b = int(tab[i, 0]), int(tab[i, 1])
            if t in (7, 8):
                raw = heads[i, :b - a].tobytes() if b - a <= 64 else \
                    eng.download(d_buf, b - a, offset=a).tobytes()
                if t == 7:
                    sps = Sps()
                    if H.h264_parse_sps(raw, len(raw), C.byref(sps)) == 0:
                        params.put_sps(sps)
                else:
                    pps = Pps()
                    if H.h264_parse_pps(raw, len(raw), C.byref(sps),
                                        C.byref(pps)) == 0:
                        params.put_pps(pps)
            else:
                descs.append((a, b - a, params.epoch))
        out = eng.decode_slices(params, d_buf, descs, reader_flags=1,
                                tier=TIER_T0, mb_capacity=8160, outputs=out)
        return tab, count, consumed, descs, out, params

    tab, count, consumed, descs, out, params = step()
    for _ in range(args.warmup):
        step(out)
    eng.sync()
    if dist is not None:
        dist.barrier()
    k_dec, k_scan = [], []
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step(out)
        k_dec.append(eng.last_kernel_ms("decode"))
        k_scan.append(eng.last_kernel_ms("scan_all"))
    eng.sync()
    if dist is not None:
        dist.barrier()
    elapsed = shard.max_over_ranks(dist, time.perf_counter() - t_start,
                                   device="cuda")
    res = eng.fetch_results(out)
    n = len(descs)
    bad = [i for i in range(n) if res[i].status != 0 or res[i].mb_count != 8160]
    if bad or consumed != len(data):
        raise SystemExit("rank %d: C4 pipeline failed (%d bad slices, consumed "
                         "%d of %d)" % (rank, len(bad), consumed, len(data)))
    verified = 0
    if rank == 0:
        import orc
        head = [data[int(tab[i, 0]):int(tab[i, 1])] for i in range(count)
                if int(data[int(tab[i, 0])]) & 0x1f in (7, 8)][:2]
        for k in range(0, n, max(1, n // max(1, args.verify))):
            a, sz, _ = descs[k]
            r = orc.parse_nalus(head + [data[a:a + sz]], want=0).results[0]
            g = res[k]
            if (r.status, r.mb_count, r.checksum, r.end_bit) != \
                    (g.status, g.mb_count, g.checksum, g.end_bit):
                raise SystemExit("parity failure on slice %d" % k)
            verified += 1
    slice_bytes = sum(d[1] for d in descs)
    all_bytes, all_mbs = shard.sum_over_ranks(
        dist, [len(data), 8160 * n], device="cuda")
    if rank == 0:
        dec_ms, scan_ms = float(np.mean(k_dec)), float(np.mean(k_scan))
        print(json.dumps({
            "metric": "bitstream_parsed_mbit_per_s",
            "value": round(8.0 * all_bytes * args.steps / elapsed / 1e6, 1),
            "unit": "Mbit/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "macroblocks_per_s": round(all_mbs * args.steps / elapsed, 1),
            "config": {
                "workload": "C4: %.1f MiB mixed Annex-B stream per GPU (%d NAL "
                            "units, %d 1080p I/P slices): scan -> NAL heads -> "
                            "host parameter sets -> one decode launch" %
                            (len(data) / 2 ** 20, count, n),
                "parallelism": "one stream per GPU, %d GPU(s), no data-path "
                               "collective" % world},
            "roofline": {
                "bound": "hbm", "kernel": "h264_amd_k_decode_slices",
                "achieved": round(slice_bytes / dec_ms / 1e6, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(slice_bytes / dec_ms / 1e6 / HBM_PEAK_GBS, 5),
                "traffic": None, "kernel_ms": round(dec_ms, 3),
                "note": ("%d slices fill %.1f of 16 wave slots per CU: the "
                         "step is bound by the ~58 ms one wavefront needs for "
                         "one 1080p slice" % (n, n / 256.0)) if n < 4096 else
                        ("%d slices = %.1f rounds of 16 waves per CU; CAVLC is "
                         "a serial chain per slice: latency / scalar-issue "
                         "bound, not HBM bound" % (n, n / 4096.0))},
            "scan": {"kernels": "mark + prefix + place",
                     "ms": round(scan_ms, 4),
                     "achieved_GBps": round(len(data) / scan_ms / 1e6, 1)},
            "cpu_baseline": None,
            "verified_slices_vs_oracle": verified,
            "gen_seconds": round(gen_s, 1)}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2", choices=["c2", "c4"],
                    help="c2: batch of I-slices (BASELINE config 2, the "
                         "headline); c4: mixed Annex-B stream through scan + "
                         "decode (config 4)")
    ap.add_argument("--mib", type=int, default=64,
                    help="c4: stream size per GPU in MiB")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--slices", type=int, default=4096,
                    help="slices per GPU (BASELINE config 2: 4096)")
    ap.add_argument("--tier", type=int, default=1, help="0 none, 1 T0, 2 T1")
    ap.add_argument("--verify", type=int, default=16,
                    help="slices re-parsed by the oracle after the run")
    ap.add_argument("--cpu-sample", type=int, default=0,
                    help="slices in the CPU baseline sample (0 = auto)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ORDER MATTERS: torch ships its own ROCm runtime (same soname as the
    # system one the engine was linked against).  With torch initialised
    # first the engine library resolves to that already-loaded copy and the
    # process has ONE HIP runtime; the other way round torch finds no GPU
    # (measured, tools/dist_probe.py).  libh264_amd loads its library lazily.
    dist = None
    if world > 1 or os.environ.get("H264_AMD_FORCE_DIST"):
        # H264_AMD_FORCE_DIST: rehearse the N > 1 path on a one-GPU box
        import torch
        import torch.distributed as dist_mod
        dist = dist_mod
        torch.cuda.set_device(local_rank)
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ.setdefault("MASTER_PORT", "29573")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl",
                                device_id=torch.device("cuda", local_rank))
    n_gpus = world

    import libh264_amd
    from libh264_amd import gen, Params, shard

    from libh264_amd.ctypes_abi import Sps, Pps, TIER_NONE, TIER_T0, TIER_T1

    # a GPU box gives each GPU a share of 16 host cores whatever os.cpu_count()
    # says; the CPU baseline and the generator stay inside that share
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, int(os.environ.get("H264_AMD_HOST_CORES", "16"))))
    threads = cores
    if args.workload == "c4":
        eng = libh264_amd.Engine(local_rank)
        run_c4(args, eng, dist, rank, world, cores)
        if dist is not None:
            dist.destroy_process_group()
        return
    t0 = time.time()
    cfg, buf, offs, sizes = make_workload(args.slices, shard.rank_seed(rank),
                                          threads)
    gen_s = time.time() - t0
    total_bytes = int(sizes.sum())
    total_mbs = 8160 * args.slices

    eng = libh264_amd.Engine(local_rank)
    # parameter sets: parse the generator's SPS/PPS bytes with the oracle-free
    # host parser of the product when available, else fill from the config
    sps_nal, pps_nal = gen.sps(cfg), gen.pps(cfg)
    sps, pps = Sps(), Pps()
    sps.profile_idc = cfg.profile_idc
    sps.level_idc = cfg.level_idc
    sps.chroma_format_idc = 1
    sps.pic_order_cnt_type = cfg.pic_order_cnt_type
    sps.max_num_ref_frames = cfg.max_num_ref_frames
    sps.pic_width_in_mbs_minus1 = cfg.width_mbs - 1
    sps.pic_height_in_map_units_minus1 = cfg.height_mbs - 1
    sps.frame_mbs_only_flag = 1
    sps.direct_8x8_inference_flag = cfg.direct_8x8_inference_flag
    sps.frame_cropping_flag = 1
    sps.frame_crop_bottom_offset = cfg.crop_bottom
    pps.deblocking_filter_control_present_flag = \
        cfg.deblocking_filter_control_present_flag
    params = Params()
    params.put_sps(sps)
    epoch = params.put_pps(pps)

    # largest slices first: the last wavefronts to start are the shortest
    order = shard.longest_first(sizes)
    descs = np.zeros(args.slices, dtype=[("offset", "<u8"), ("size", "<u4"),
                                         ("epoch", "<u4")])
    descs["offset"] = offs[order]
    descs["size"] = sizes[order]
    descs["epoch"] = epoch
    d_buf = eng.upload(buf)
    d_descs = eng.upload(descs)
    tier = {0: TIER_NONE, 1: TIER_T0, 2: TIER_T1}[args.tier]
    out = eng.decode_slices(params, d_buf, descs, reader_flags=1, tier=tier,
                            mb_capacity=8160, d_descs=d_descs)

    def barrier():
        eng.sync()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    for _ in range(args.warmup):
        eng.decode_slices(params, d_buf, descs, reader_flags=1, tier=tier,
                          mb_capacity=8160, d_descs=d_descs, outputs=out)
    barrier()
    kernel_ms = []
    t_start = time.perf_counter()
    for _ in range(args.steps):
        eng.decode_slices(params, d_buf, descs, reader_flags=1, tier=tier,
                          mb_capacity=8160, d_descs=d_descs, outputs=out,
                          sync=False)
        kernel_ms.append(eng.last_kernel_ms("decode"))
    barrier()
    elapsed = time.perf_counter() - t_start
    elapsed = shard.max_over_ranks(dist, elapsed, device="cuda")

    res = eng.fetch_results(out)
    bad = [i for i in range(args.slices)
           if res[i].status != 0 or res[i].mb_count != 8160]
    if bad:
        raise SystemExit("rank %d: %d slices failed on the GPU (first %d: "
                         "status %d, %d MBs)" %
                         (rank, len(bad), bad[0], res[bad[0]].status,
                          res[bad[0]].mb_count))

    # parity at full size: the oracle re-parses a sample and must reproduce the
    # checksum over every macroblock's full syntax (checker only, not timed)
    verified = 0
    cpu = None
    if rank == 0:
        import orc
        for j in range(min(args.verify, args.slices)):
            k = j * (args.slices // max(1, min(args.verify, args.slices)))
            i = int(order[k])  # result k belongs to generated slice i
            nal = bytes(buf[int(offs[i]):int(offs[i]) + int(sizes[i])])
            ref = orc.parse_nalus([sps_nal, pps_nal, nal], want=0)
            r = ref.results[0]
            g = res[k]
            if (r.status, r.mb_count, r.checksum, r.end_bit, r.hdr_len_bits,
                    r.rawdata_off) != (g.status, g.mb_count, g.checksum,
                                       g.end_bit, g.hdr_len_bits,
                                       g.rawdata_off):
                raise SystemExit("parity failure on slice %d" % i)
            verified += 1
        if world == 1:
            # ~1 M macroblocks/s per core: 128 slices per core is 15-20 s
            # of CPU work, about a second of wall time on the box's share
            n_s = args.cpu_sample or min(args.slices, cores * 128)
            sel = np.arange(n_s)
            o = np.ascontiguousarray(offs[sel])
            z = np.ascontiguousarray(sizes[sel])
            secs, mbs, byts = C.c_double(), C.c_uint64(), C.c_uint64()
            rc = orc.lib().orc_bench_slices(
                buf.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p),
                z.ctypes.data_as(C.c_void_p), n_s, sps_nal, len(sps_nal),
                pps_nal, len(pps_nal), 1, cores, C.byref(secs),
                C.byref(mbs), C.byref(byts))
            if rc != 0 or mbs.value != 8160 * n_s:
                raise SystemExit("cpu baseline failed: rc %d" % rc)
            # one thread on a smaller sample of the same slices (BASELINE.md 3:
            # single-thread and all-cores numbers side by side)
            n_1 = min(n_s, 96)
            s1, m1, b1 = C.c_double(), C.c_uint64(), C.c_uint64()
            rc = orc.lib().orc_bench_slices(
                buf.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p),
                z.ctypes.data_as(C.c_void_p), n_1, sps_nal, len(sps_nal),
                pps_nal, len(pps_nal), 1, 1, C.byref(s1), C.byref(m1),
                C.byref(b1))
            if rc != 0 or m1.value != 8160 * n_1:
                raise SystemExit("cpu baseline (1 thread) failed: rc %d" % rc)
            cpu = {
                "value": round(8.0 * byts.value / secs.value / 1e6, 1),
                "unit": "Mbit/s",
                "cores": cores,
                "kind": "port",
                "mb_per_s": round(mbs.value / secs.value, 1),
                "sample": "%d of the same 1080p I-slices (%.1f MB), oracle "
                          "reader with slice_data, one reader per thread, "
                          "%.1f s wall" % (n_s, byts.value / 1e6, secs.value),
                "single_thread": {
                    "value": round(8.0 * b1.value / s1.value / 1e6, 1),
                    "mb_per_s": round(m1.value / s1.value, 1),
                    "sample": "%d slices, %.1f s" % (n_1, s1.value)},
            }

    ms_per_step = 1e3 * elapsed / args.steps
    all_bytes, all_mbs = shard.sum_over_ranks(dist, [total_bytes, total_mbs],
                                              device="cuda")
    mbit_s = 8.0 * all_bytes * args.steps / elapsed / 1e6
    k_ms = float(np.mean(kernel_ms))
    achieved = total_bytes / (k_ms * 1e-3) / 1e9
    out_bytes = args.slices * (64 + 8160 * out["stride"])

    # HBM traffic per launch comes from PMC passes of this very command
    # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass; a
    # profiler cannot run inside the timed process).  The summary is committed
    # under profiles/ and only used when it was taken on the identical
    # workload (same seeded bytes, slice count and tier).
    traffic, traffic_src = None, None
    try:
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles",
                                               "r*_decode_hbm_traffic.json"))):
            t = json.load(open(f))
            if (t.get("bytes_per_gpu") == total_bytes and
                    t.get("slices") == args.slices and
                    t.get("tier") == ["none", "T0", "T1"][args.tier]):
                traffic = t["hbm_bytes_per_launch"]
                traffic_src = os.path.relpath(f, ROOT)
    except (OSError, ValueError, KeyError):
        traffic = None

    if rank == 0:
        line = {
            "metric": "bitstream_parsed_mbit_per_s",
            "value": round(mbit_s, 1),
            "unit": "Mbit/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "macroblocks_per_s": round(all_mbs * args.steps / elapsed, 1),
            "config": {
                "workload": "C2: batch of %d independent 1080p Baseline CAVLC "
                            "I-slices per GPU (8160 MBs each, mean %.0f "
                            "bits/MB), one slice per wavefront" %
                            (args.slices, 8.0 * total_bytes / total_mbs),
                "slices_per_gpu": args.slices,
                "bytes_per_gpu": total_bytes,
                "output_tier": ["none", "T0", "T1"][args.tier],
                "parallelism": "slices sharded over %d GPU(s), no data-path "
                               "collective" % n_gpus,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "h264_amd_k_decode_slices",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel_ms": round(k_ms, 3),
                "algorithmic_bytes_read": total_bytes,
                "algorithmic_bytes_written": out_bytes,
                # a serial-dependency kernel is described by
                # its waves in flight and what one wave gets done per cycle
                "waves_in_flight": min(args.slices, 4096),
                "waves_per_cu": min(args.slices, 4096) / 256.0,
                "bits_per_wave_cycle": round(
                    8.0 * total_bytes / args.slices /
                    (k_ms * 1e-3 * 2.4e9 / max(1.0, args.slices / 4096.0)), 5),
                "clock_ghz_assumed": 2.4,
                "note": "CAVLC is a serial dependency chain per slice: the "
                        "kernel is latency bound, not HBM bound",
            },
            "cpu_baseline": cpu,
            "verified_slices_vs_oracle": verified,
            "gen_seconds": round(gen_s, 1),
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
