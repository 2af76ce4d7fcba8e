set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04; mkdir -p $OUT; cd /tmp
timeout 1100 python3 $ROOT/tools/fuzz_campaign.py 900 47000 persistent > $OUT/r04_fuzz_final_persistent_long.json 2> $OUT/fz1.err; echo "rc=$?"
timeout 800 python3 $ROOT/tools/fuzz_campaign.py 600 48000 fresh 3 > $OUT/r04_fuzz_final_xl.json 2> $OUT/fz2.err; echo "rc=$?"
