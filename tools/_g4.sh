# like _g2.sh plus a fib19 kernel trace
set -u
ROOT=$(pwd); TAG=${1:-x}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
for w in 20 22 fib19; do python3 $ROOT/tools/point.py $w --steps 20 > $OUT/point_$w.json 2>$OUT/point_$w.err; cat $OUT/point_$w.json | cut -c1-420; done
for w in 22 fib19; do
rm -rf /tmp/tl_$w; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$w -- python3 $ROOT/tools/point.py $w --steps 3 --warmup 1 > /dev/null 2>&1
F=$(ls /tmp/tl_$w/*/*kernel_trace.csv | head -1)
python3 $ROOT/tools/timeline_dump.py $F > $OUT/tl_$w.txt
python3 $ROOT/tools/timeline_dump.py $F --summary > $OUT/tl_${w}_summary.txt
python3 $ROOT/tools/timeline_gaps.py $F 15 > $OUT/tl_${w}_gaps.txt
done
