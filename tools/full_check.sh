set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/full; mkdir -p $OUT
timeout 2400 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
cd /tmp
timeout 900 python3 $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python3 -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['metric_point'], d['roofline']['frac'], d['roofline'].get('traffic'), d.get('parity_checked'))"
