set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-x}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for tp in 0 1; do for lg in 20 21 22; do
echo "== two_pass=$tp log=$lg"; BFHIP_FFT_TWO_PASS=$tp python3 $ROOT/tools/fft_roofline.py $lg 4 32 | python3 -c "
import sys,json
for r in json.load(sys.stdin):
    print(r['columns'], r['ifft_plus_lde_plus_fft_ms'], {k:(v['avg_us'],v['GB/s_moved']) for k,v in r.items() if isinstance(v,dict)})"
done; done | tee $OUT/fft_ab.txt
