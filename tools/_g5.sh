set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-x}; mkdir -p $OUT
python3 -m pytest tests/test_gpu_fft.py tests/test_gpu_prove.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
for tp in 0 1; do for w in 20 22 fib19; do
echo -n "two_pass=$tp $w: "; BFHIP_FFT_TWO_PASS=$tp python3 $ROOT/tools/point.py $w --steps 30 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_proof'], d['ms_min'], d['proof_sha256'][:12])"
done; done | tee $OUT/ab.txt
python3 $ROOT/tools/fft_roofline.py > $OUT/fft_roofline.json 2>$OUT/fft_roofline.err; head -c 1500 $OUT/fft_roofline.json
