set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04; mkdir -p $OUT
timeout 2400 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_final.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_final.log
bash tools/profile_round.sh r04 bench trace latency misc shard > $ROOT/gpurun_out/r04_round2.log 2>&1; echo "round rc=$?"
cd /tmp
timeout 700 python3 $ROOT/tools/fuzz_campaign.py 420 41000 fresh > $OUT/r04_fuzz_final_fresh.json 2> $OUT/fuzz_fresh.err; echo "fuzz fresh rc=$?"
timeout 700 python3 $ROOT/tools/fuzz_campaign.py 420 42000 persistent > $OUT/r04_fuzz_final_persistent.json 2> $OUT/fuzz_pers.err; echo "fuzz persistent rc=$?"
BFHIP_MAILBOX=1 timeout 500 python3 $ROOT/tools/fuzz_campaign.py 240 43000 fresh > $OUT/r04_fuzz_final_mailbox_on.json 2> $OUT/fuzz_mb.err; echo "fuzz mailbox rc=$?"
