set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03a; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in 20 22 fib19; do python3 $ROOT/tools/point.py $w --steps 20 > $OUT/point_$w.json 2>$OUT/point_$w.err; done
for w in 20 22; do
rm -rf /tmp/tl_$w; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$w -- python3 $ROOT/tools/point.py $w --steps 3 --warmup 1 > /dev/null 2>&1
F=$(ls /tmp/tl_$w/*/*kernel_trace.csv | head -1)
python3 $ROOT/tools/timeline_dump.py $F > $OUT/tl_$w.txt
python3 $ROOT/tools/timeline_dump.py $F --summary > $OUT/tl_${w}_summary.txt
python3 $ROOT/tools/timeline_gaps.py $F 15 > $OUT/tl_${w}_gaps.txt
done
cat $OUT/point_*.json
