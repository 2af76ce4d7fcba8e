#!/bin/bash
# Last check of a round on the GPU box: the whole GPU suite, smoke(), then randomized parity campaigns on the final build (gpurun_out/<round>/).
set -u
R=${1:-r06}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$R; mkdir -p $OUT
timeout 2400 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_final.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest_final.log | tail -1
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
cd /tmp
timeout 600 python3 $ROOT/tools/fuzz_campaign.py 360 44000 fresh > $OUT/${R}_fuzz_final_fresh.json 2> $OUT/fuzz_fresh.err; echo "fuzz fresh rc=$?"
timeout 600 python3 $ROOT/tools/fuzz_campaign.py 360 45000 persistent > $OUT/${R}_fuzz_final_persistent.json 2> $OUT/fuzz_pers.err; echo "fuzz persistent rc=$?"
BFHIP_MAILBOX=1 timeout 400 python3 $ROOT/tools/fuzz_campaign.py 180 46000 fresh > $OUT/${R}_fuzz_final_mailbox_on.json 2> $OUT/fuzz_mb.err; echo "fuzz mailbox rc=$?"
timeout 500 python3 $ROOT/tools/fuzz_campaign.py 300 47000 pool > $OUT/${R}_fuzz_final_pool.json 2> $OUT/fuzz_pool.err; echo "fuzz pool rc=$?"
