#!/bin/bash
# Instruction-cache counters of the two slice kernels (one rocprofv3 --pmc pass
# each, through tools/pmc_decode.py --counters): run on the GPU box.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/ic
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C=SQC_ICACHE_REQ,SQC_ICACHE_HITS,SQC_ICACHE_MISSES,SQC_ICACHE_MISSES_DUPLICATE,SQC_TC_INST_REQ,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES
python3 $R/tools/pmc_decode.py --cabac --slices 4096 --counters $C > $O/cabac.json 2> $O/cabac.err
python3 $R/tools/pmc_decode.py --slices 4096 --counters $C > $O/cavlc.json 2> $O/cavlc.err
python3 $R/tools/pmc_decode.py --slices 512 --counters $C > $O/cavlc512.json 2> $O/cavlc512.err
