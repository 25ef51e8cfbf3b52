#!/bin/bash
# One pass of every GPU fuzzer with a time box each (run on the GPU box):
#   tools/fuzz_round.sh SEED_BASE SECONDS_EACH
# Logs under gpurun_out/fuzz/.  A fuzzer that is still running at its time box
# is stopped there; what it printed up to then is the result.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/fuzz
mkdir -p $O
cd $R
base=${1:-50000}
secs=${2:-150}
run() { # name, command...
	local name=$1; shift
	timeout -k 10 $secs "$@" > $O/$name.log 2>&1
	local rc=$?
	echo "$name rc=$rc $(tail -1 $O/$name.log | cut -c1-160)"
	# 124: the time box; anything else that is not 0 is a failure
	[ $rc -eq 0 ] || [ $rc -eq 124 ]
}
run campaign python3 tools/fuzz_campaign.py $base 100000 &&
run dropin python3 tools/fuzz_dropin.py $base 100000 &&
run dropin_params python3 tools/fuzz_dropin.py $base 100000 params &&
run params python3 tools/fuzz_params.py $base 100000 &&
run cabac_full python3 tools/fuzz_cabac_full.py $base 100000 &&
run cabac_real python3 tools/fuzz_cabac_real.py $base 100000 &&
run scan python3 tools/fuzz_scan.py $base 100000
