#!/bin/bash
# CABAC kernel after a change: GPU parity tests, timing with result digest,
# instruction mix per bin.  Run on the GPU box; output under gpurun_out/cab/.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/cab
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_cabac_full.py tests/test_gpu_cabac.py -x -q -m gpu > $O/tests.log 2>&1
timeout -k 10 300 python3 tools/bench_cabac.py --slices 4096 --reps 5 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/tools/pmc_decode.py --cabac --slices 4096 > $O/pmc.json 2> $O/pmc.err
