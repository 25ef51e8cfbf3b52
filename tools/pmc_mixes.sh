#!/bin/bash
# instruction mix of the CAVLC kernel per macroblock for several macroblock
# mixes (where do the branches come from): run on the GPU box
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/mix
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/pmc_mixes.jsonl
for m in inxn_empty i16_empty inxn_sparse inxn_mid inxn_dense bench; do
	timeout -k 10 200 python3 $R/tools/pmc_decode.py --slices 1024 --mix "$m" --counters SQ_INSTS_SALU,SQ_INSTS_VALU,SQ_INSTS_BRANCH,SQ_INSTS_LDS,SQ_WAVE_CYCLES >> $O/pmc_mixes.jsonl 2>> $O/pmc_mixes.err || exit 1
done
