set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/b1; mkdir -p $OUT
python3 -m pytest tests/test_gpu_components.py tests/test_gpu_fuzz.py tests/test_gpu_prove.py tests/test_gpu_shard.py tests/test_gpu_pool.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for p in 1 0; do for w in fib19 22; do echo -n "pairs=$p $w: "; BFHIP_CONSTRAINT_PAIRS=$p python3 $ROOT/tools/point.py $w --steps 20 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_proof'], d['ms_min'], d['proof_sha256'][:12])"; done; done; done > $OUT/constraint_pairs_ab.txt 2>&1
for p in 1 0; do
rm -rf /tmp/cp_$p; BFHIP_CONSTRAINT_PAIRS=$p timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp_$p -- python3 $ROOT/tools/point.py fib19 --steps 10 --warmup 2 > /dev/null 2>&1
grep -i "constraints\|quotients\|eval_at" $(ls /tmp/cp_$p/*/*kernel_stats.csv | head -1) > $OUT/constraint_pairs_${p}_stats.txt
done
for w in 20; do
rm -rf /tmp/tl_$w; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$w -- python3 $ROOT/tools/point.py $w --steps 3 --warmup 1 > /dev/null 2>&1
F=$(ls /tmp/tl_$w/*/*kernel_trace.csv | head -1)
python3 $ROOT/tools/timeline_dump.py $F > $OUT/tl_$w.txt
python3 $ROOT/tools/timeline_dump.py $F --summary > $OUT/tl_${w}_summary.txt
python3 $ROOT/tools/timeline_gaps.py $F 15 > $OUT/tl_${w}_gaps.txt
done
cat $OUT/constraint_pairs_ab.txt; cat $OUT/constraint_pairs_1_stats.txt $OUT/constraint_pairs_0_stats.txt
