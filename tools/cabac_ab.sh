#!/bin/bash
# A/B of CABAC kernel builds: tools/cabac_ab.sh variant1 variant2 ... (names of
# libh264_amd/lib/variants/libengine_NAME.so; "in-tree" for the product build)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/cab
mkdir -p $O
cd $R
: > $O/ab.jsonl
for v in "$@"; do
	if [ "$v" = in-tree ]; then unset H264_AMD_ENGINE_LIB; else export H264_AMD_ENGINE_LIB=$R/libh264_amd/lib/variants/libengine_$v.so; fi
	timeout -k 10 300 python3 tools/bench_cabac.py --slices 4096 --reps 5 | sed "s/\"lib\": \"[^\"]*\"/\"lib\": \"$v\"/" >> $O/ab.jsonl || exit 1
done
