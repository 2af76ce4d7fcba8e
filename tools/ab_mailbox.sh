set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/cols; mkdir -p $OUT
timeout 1200 python3 -m pytest tests/test_gpu_prove.py tests/test_gpu_ops.py tests/test_gpu_components.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
cd /tmp; export TMPDIR=/tmp
for w in 20 22 fib19; do python3 $ROOT/tools/point.py $w --steps 40 | cut -c1-200; done
python3 $ROOT/tools/merkle_shapes.py 2>&1 | tail -16
for w in 22; do
rm -rf /tmp/tl_$w; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$w -- python3 $ROOT/tools/point.py $w --steps 3 --warmup 1 > /dev/null 2>&1
F=$(ls /tmp/tl_$w/*/*kernel_trace.csv | head -1)
python3 $ROOT/tools/timeline_dump.py $F > $OUT/tl_$w.txt
python3 $ROOT/tools/timeline_dump.py $F --summary > $OUT/tl_${w}_summary.txt
python3 $ROOT/tools/timeline_gaps.py $F 15 > $OUT/tl_${w}_gaps.txt
done
