#!/bin/bash
# Everything profiles/rNN_* is made from, in one GPU-box call:
#
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh r02'
#
# writes gpurun_out/<tag>/...; the summaries are then copied into profiles/ by
# hand (see profiles/README.md for which file is which).  Counter passes run on
# their own (rocprofv3 --pmc with --kernel-trace only), the program itself after
# `--` (python3, never a launcher).
set -o pipefail
tag=${1:-r02}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
step() { echo "== $1 $(date +%T)"; }

step "bench line"
python3 bench.py > "$out/bench.json" 2> "$out/bench.err" || exit 1
tail -c 600 "$out/bench.json"

step "rocprofv3 kernel stats of the bench command"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- \
	python3 bench.py --steps 5 --warmup 2 --verify 0 --cpu-sample 16 --no-extras \
	> "$out/stats.log" 2>&1 || exit 1
find "$out/stats" -name "*kernel_stats.csv" -exec cp {} "$out/bench_kernel_stats.csv" \;

step "HBM traffic, one counter per pass"
for c in FETCH_SIZE WRITE_SIZE; do
	rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc_$c" -- \
		python3 bench.py --steps 2 --warmup 0 --verify 0 --cpu-sample 16 --no-extras \
		> "$out/pmc_$c.log" 2>&1 || exit 1
done
nbytes=$(python3 -c "import json,sys; print([json.loads(l) for l in open('$out/bench.json') if l.startswith('{')][-1]['config']['bytes_per_gpu'])")
python3 tools/pmc_traffic.py "$out/pmc_FETCH_SIZE" "$out/pmc_WRITE_SIZE" "$nbytes" \
	"$out/decode_hbm_traffic.json" > "$out/pmc_traffic.log" 2>&1 || cat "$out/pmc_traffic.log"

step "instruction mix of the decode kernel (PMC)"
python3 tools/pmc_decode.py --slices 1024 --out /tmp/pmc_decode_$tag > "$out/pmc_decode.log" 2>&1 || exit 1
tail -n 30 "$out/pmc_decode.log"

step "decode kernel by macroblock mix"
python3 tools/bench_mix.py > "$out/mix.log" 2>&1 || exit 1

step "full chip / one wave per SIMD"
python3 tools/ab_decode.py --slices 4096,1024,512 --steps 10 > "$out/ab.log" 2>&1 || exit 1
cat "$out/ab.log"

step "side kernels"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/side" -- \
	python3 tools/bench_kernels.py --mib 1024 --slices 4096 > "$out/side.log" 2>&1 || exit 1
find "$out/side" -name "*kernel_stats.csv" -exec cp {} "$out/side_kernels_stats.csv" \;

step "issue probe"
if [ -f tools/probe/issue_probe.hip ]; then
	/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_probe tools/probe/issue_probe.hip \
		> "$out/issue_probe_build.log" 2>&1 && /tmp/issue_probe > "$out/issue_probe.log" 2>&1
fi

step "CABAC kernel on config 3 with real content"
python3 tools/bench_cabac.py --slices 4096 --reps 5 > "$out/cabac_bench.json" 2> "$out/cabac_bench.err" || exit 1
tail -c 400 "$out/cabac_bench.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/cabac_stats" -- \
	python3 tools/bench_cabac.py --slices 4096 --reps 5 > "$out/cabac_stats.log" 2>&1 || exit 1
find "$out/cabac_stats" -name "*kernel_stats.csv" -exec cp {} "$out/cabac_kernel_stats.csv" \;
for c in FETCH_SIZE WRITE_SIZE; do
	rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/cpmc_$c" -- \
		python3 tools/bench_cabac.py --slices 4096 --reps 1 > "$out/cpmc_$c.log" 2>&1 || exit 1
done
cbytes=$(python3 -c "import json; d=json.loads(open('$out/cabac_bench.json').read().strip().splitlines()[-1]); print(int(round(d['mbit_per_s']*1e6/8*d['ms']/1e3)))")
python3 tools/pmc_traffic.py "$out/cpmc_FETCH_SIZE" "$out/cpmc_WRITE_SIZE" "$cbytes" \
	"$out/cabac_hbm_traffic.json" h264_amd_k_decode_cabac > "$out/cpmc_traffic.log" 2>&1 || cat "$out/cpmc_traffic.log"
python3 tools/pmc_decode.py --cabac --slices 4096 --out /tmp/pmc_cabac_$tag > "$out/pmc_cabac.log" 2>&1 || exit 1
tail -n 2 "$out/pmc_cabac.log"
rm -rf "$out/cabac_stats" "$out"/cpmc_FETCH_SIZE "$out"/cpmc_WRITE_SIZE

step "config 1 (dump tool) and config 4 at 1 GiB"
python3 bench.py --workload c1 > "$out/bench_c1.json" 2> "$out/bench_c1.err" || exit 1
python3 bench.py --workload c4 --mib 1024 --steps 3 --warmup 1 > "$out/bench_c4_1gib.json" 2> "$out/bench_c4_1gib.err" || exit 1

step "drop-in latency per NAL unit"
python3 tools/bench_dropin.py --per-nal > "$out/dropin_per_nal.json" 2> "$out/dropin_per_nal.err" || true

# raw traces are large: keep the summaries only
rm -rf "$out/stats" "$out/side" "$out"/pmc_FETCH_SIZE "$out"/pmc_WRITE_SIZE
step done
