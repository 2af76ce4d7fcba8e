set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03fuzz2; mkdir -p $OUT
python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 1100 python3 tools/fuzz_campaign.py 600 50000 persistent > $OUT/r03_fuzz_persistent.json 2> $OUT/fuzz_persistent.err; echo "persistent rc=$?"; python3 -c "
import json; d=json.load(open('$OUT/r03_fuzz_persistent.json')); print({k:(v if not isinstance(v,list) else len(v)) for k,v in d.items()})"
timeout 700 python3 tools/fuzz_campaign.py 300 60000 fresh 5 > $OUT/r03_fuzz_campaign_xl.json 2> $OUT/fuzz_xl.err; echo "xl rc=$?"; python3 -c "
import json; d=json.load(open('$OUT/r03_fuzz_campaign_xl.json')); print({k:(v if not isinstance(v,list) else len(v)) for k,v in d.items()})"
