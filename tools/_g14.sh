set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03fuzz; mkdir -p $OUT
timeout 900 python3 tools/fuzz_campaign.py 420 30000 fresh > $OUT/r03_fuzz_campaign.json 2> $OUT/fuzz_fresh.err; echo "fresh rc=$?"; tail -c 900 $OUT/r03_fuzz_campaign.json
timeout 900 python3 tools/fuzz_campaign.py 420 40000 persistent > $OUT/r03_fuzz_persistent.json 2> $OUT/fuzz_persistent.err; echo "persistent rc=$?"; tail -c 900 $OUT/r03_fuzz_persistent.json
