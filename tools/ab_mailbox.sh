set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/mb; mkdir -p $OUT
cd /tmp
for m in 1 0 1 0; do
BFHIP_MAILBOX=$m timeout 300 python3 $ROOT/tools/point.py 20 --steps 400 --warmup 2 --times 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); t=d['ms_sorted']; print('$m 2^20 x400', d['ms_per_proof'], 'min', d['ms_min'], 'median', t[len(t)//2], 'p90', t[int(len(t)*0.9)], 'max', t[-1])"
done
for m in 1 0 1 0; do
BFHIP_MAILBOX=$m timeout 300 python3 $ROOT/tools/point.py 22 --steps 100 --warmup 2 --times 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); t=d['ms_sorted']; print('$m 2^22 x100', d['ms_per_proof'], 'min', d['ms_min'], 'median', t[len(t)//2], 'p90', t[int(len(t)*0.9)], 'max', t[-1])"
done
