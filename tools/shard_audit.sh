#!/bin/bash
# Which kernels does every rank of a shard group repeat? (run through gpurun from the repository root)
#   gpurun -- 'bash tools/shard_audit.sh <tag> [N] [what] [extra shard_kernels.py flags]'
# Kernel traces of one rank and of N ranks time-sharing the GPU -> gpurun_out/<tag>/shard_redundancy_<what>_N<N>.txt
# AUDIT_PMC=1: the same under `--pmc SQ_INSTS_VALU` — counter collection serialises the dispatches, so the durations are free of the
# stretch that co-running launches of different ranks cause in the plain trace (file suffix _serialised).
set -u
ROOT=$(pwd); TAG=${1:-audit}; N=${2:-8}; WHAT=${3:-fib19}; shift 3 || true
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SUF=""; PMC=""
if [ "${AUDIT_PMC:-0}" = "1" ]; then SUF="_serialised"; PMC="--pmc SQ_INSTS_VALU"; fi
for n in 1 $N; do
  rm -rf /tmp/sk_$n
  timeout 420 rocprofv3 --kernel-trace $PMC --output-format csv -d /tmp/sk_$n -- python3 $ROOT/tools/shard_kernels.py $n $WHAT --steps 3 "$@" > $OUT/sk_${WHAT}_$n$SUF.json 2> $OUT/sk_${WHAT}_$n$SUF.err
  tail -1 $OUT/sk_${WHAT}_$n$SUF.json | cut -c1-300
done
python3 $ROOT/tools/shard_redundancy.py $(ls /tmp/sk_1/*/*kernel_trace.csv | head -1) $OUT/sk_${WHAT}_1$SUF.json $(ls /tmp/sk_$N/*/*kernel_trace.csv | head -1) $OUT/sk_${WHAT}_$N$SUF.json > $OUT/shard_redundancy_${WHAT}_N$N$SUF.txt 2>&1
cat $OUT/shard_redundancy_${WHAT}_N$N$SUF.txt
if [ -z "$SUF" ]; then gzip -c $(ls /tmp/sk_$N/*/*kernel_trace.csv | head -1) > $OUT/sk_${WHAT}_$N.kernel_trace.csv.gz; python3 $ROOT/tools/shard_timeline.py $(ls /tmp/sk_$N/*/*kernel_trace.csv | head -1) $OUT/sk_${WHAT}_$N.json 30 > $OUT/shard_timeline_${WHAT}_N$N.txt 2>&1; cat $OUT/shard_timeline_${WHAT}_N$N.txt; fi
