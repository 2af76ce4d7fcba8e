#!/bin/bash
# A/B of one environment switch at the three reference points (run on the GPU box from the repository root):
#   bash tools/ab_env.sh <VAR> "<value> <value> ..." [out file]      e.g.  bash tools/ab_env.sh BFHIP_FFT_TWO_PASS "0 1"
# Two rounds per value, interleaved, so box drift shows up as disagreement between the rounds. Prints "VAR=value point: ms mean min sha".
VAR=$1; VALUES=$2; OUT=${3:-/dev/stdout}
ROOT=$(pwd)
for round in 1 2; do for v in $VALUES; do for w in 20 22 fib19; do
  line=$(env $VAR=$v python3 $ROOT/tools/point.py $w --steps 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_proof'], d['ms_min'], d.get('proof_sha256','')[:12])")
  echo "$VAR=$v $w: $line" >> $OUT
done; done; done
