set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-ab}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for ov in 0 1 2 3; do for w in 20 22 fib19; do
echo -n "overlap=$ov $w: "; BFHIP_OVERLAP=$ov python3 $ROOT/tools/point.py $w --steps 30 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_proof'], d['ms_min'], d['proof_sha256'][:12])"
done; done; done | tee $OUT/ab.txt
