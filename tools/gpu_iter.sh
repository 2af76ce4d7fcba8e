#!/bin/bash
# One iteration on the GPU box while working on the kernels (run through gpurun from the repository root):
#   gpurun --timeout 1500 -- 'bash tools/gpu_iter.sh <tag> [trace]'
# the full -m gpu suite, then ms per proof at the three reference points (2^20 rows, 2^22 rows = BASELINE's metric, fib19) and, with `trace`,
# a rocprofv3 kernel trace of the two small points reduced to per-kernel summaries, idle gaps and the launch list (gpurun_out/<tag>/).
set -u
ROOT=$(pwd); TAG=${1:-iter}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
for w in 20 22 fib19; do python3 $ROOT/tools/point.py $w --steps 20 > $OUT/point_$w.json 2>$OUT/point_$w.err; cut -c1-420 $OUT/point_$w.json; done
if [ "${2:-}" = "trace" ]; then
for w in 20 22; do
rm -rf /tmp/tl_$w; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$w -- python3 $ROOT/tools/point.py $w --steps 3 --warmup 1 > /dev/null 2>&1
F=$(ls /tmp/tl_$w/*/*kernel_trace.csv | head -1)
python3 $ROOT/tools/timeline_dump.py $F > $OUT/tl_$w.txt
python3 $ROOT/tools/timeline_dump.py $F --summary > $OUT/tl_${w}_summary.txt
python3 $ROOT/tools/timeline_gaps.py $F 15 > $OUT/tl_${w}_gaps.txt
done
fi
