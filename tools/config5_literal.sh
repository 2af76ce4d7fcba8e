#!/bin/bash
# BASELINE config 5 in its literal shape on a one-GPU box: the 2^26-row synthetic trace, Poseidon252 MerkleChannel, proved ONCE by 8 ranks of an
# in-process shard group (all on this GPU), next to the same proof by one rank. Memory is checked on the 2^24-row trace first: the 2^26-row
# run is started only if four times that footprint fits the device. (gpurun -- 'bash tools/config5_literal.sh <tag>')
set -u
ROOT=$(pwd); TAG=${1:-config5}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp
timeout 600 python3 $ROOT/tools/shard_kernels.py 8 24 --steps 1 --warmup 1 --poseidon 2> $OUT/p24_8.err | tail -1 > $OUT/p24_8.json
python3 - "$OUT/p24_8.json" <<'PY' || exit 0
import json, sys
d = json.load(open(sys.argv[1]))
need = 4 * d["device_GB_reserved_all_ranks"]
print("2^24, 8 ranks:", d["ms_per_proof_wall"], "ms; reserved", d["device_GB_reserved_all_ranks"], "GB; arena peaks", d["arena_peak_GB_per_rank"], "-> 2^26 needs about", need, "GB")
sys.exit(0 if need < 235 else 1)
PY
for n in 8 1; do
  timeout 900 python3 $ROOT/tools/shard_kernels.py $n 26 --steps 1 --warmup 1 --poseidon 2> $OUT/p26_$n.err | tail -1 > $OUT/p26_$n.json
  python3 -c "
import json,sys
d=json.load(open('$OUT/p26_$n.json')); print('2^26,', d['ranks_on_one_gpu'], 'ranks:', d['ms_per_proof_wall'], 'ms', d['proof_sha256'], 'reserved GB', d['device_GB_reserved_all_ranks'], d['rank0_phase_ms_last_proof'])"
done
