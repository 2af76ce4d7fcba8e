#!/bin/bash
# Regenerates the measurement artefacts of a round on the GPU box (run through gpurun from the repository root):
#   gpurun -- 'bash tools/profile_round.sh r02'
# Writes into gpurun_out/<round>/ ; copy what should be judged into profiles/.
set -u
R=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# 1. headline bench line (roofline measured live with HIP events, CPU baseline on the host cores)
python3 "$ROOT/bench.py" > "$OUT/bench.log" 2>&1; grep '^{"metric"' "$OUT/bench.log" | tail -1 > "$OUT/${R}_bench.json"
# 2. same command under rocprofv3 --kernel-trace --stats (per-kernel average durations must agree with the roofline object)
rm -rf /tmp/prof_stats; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-sweep > "$OUT/bench_under_rocprof.log" 2>&1
grep '^{"metric"' "$OUT/bench_under_rocprof.log" | tail -1 > "$OUT/${R}_bench_under_rocprof.json"
cp $(ls /tmp/prof_stats/*/*kernel_stats.csv | head -1) "$OUT/${R}_bench_kernel_stats.csv"
# 2b. idle-gap analysis of one proof without the event instrumentation
rm -rf /tmp/prof_tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-kernel-events --no-sweep --steps 3 --warmup 1 > /dev/null 2>&1
python3 "$ROOT/tools/timeline_gaps.py" $(ls /tmp/prof_tl/*/*kernel_trace.csv | head -1) 12 > "$OUT/${R}_timeline_gaps.txt" 2>&1
python3 "$ROOT/tools/fft_launches.py" $(ls /tmp/prof_tl/*/*kernel_trace.csv | head -1) > "$OUT/${R}_fft_launches.txt" 2>&1
for K in k_quotients k_constraints k_fold k_eval; do python3 "$ROOT/tools/fft_launches.py" $(ls /tmp/prof_tl/*/*kernel_trace.csv | head -1) $K | head -12; done > "$OUT/${R}_field_kernel_launches.txt" 2>&1
# 3. HBM traffic counters, one pass each, kernel trace only
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$C; rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/prof_$C -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-sweep > "$OUT/pmc_$C.log" 2>&1
done
python3 "$ROOT/tools/pmc_traffic.py" /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE > "$OUT/${R}_pmc_traffic.json"
# 3b. effective clock and issue-slot accounting of the Merkle kernel and of the register-only Blake2s micro-benchmark (one counter pass each)
CNT="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"
rm -rf /tmp/prof_clock; rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d /tmp/prof_clock -- python3 "$ROOT/tools/merkle_shapes.py" > /dev/null 2>&1
python3 "$ROOT/tools/merkle_clock.py" /tmp/prof_clock > "$OUT/${R}_merkle_clock.json" 2>&1
hipcc -O3 --offload-arch=gfx950 -o /tmp/ubench_blake "$ROOT/tools/ubench_blake.hip" 2>/dev/null && { rm -rf /tmp/prof_ub; rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d /tmp/prof_ub -- /tmp/ubench_blake > /dev/null 2>&1; python3 "$ROOT/tools/merkle_clock.py" /tmp/prof_ub k_bench > "$OUT/${R}_ubench_blake_clock.json" 2>&1; }
# 4. FFT kernel run (BASELINE config 3 (i)), Merkle shapes, one-call end-to-end, large synthetic trace
python3 "$ROOT/tools/fft_roofline.py" > "$OUT/${R}_fft_roofline.json" 2> "$OUT/fft_roofline.err"
python3 "$ROOT/tools/merkle_shapes.py" > "$OUT/${R}_merkle_shapes.txt" 2>&1
python3 "$ROOT/tools/trace_time.py" > "$OUT/${R}_one_call.txt" 2>&1
python3 "$ROOT/tools/big_trace.py" > "$OUT/${R}_big_trace.json" 2> "$OUT/big_trace.err"
for n in 2 3; do python3 "$ROOT/bench.py" --no-cpu-baseline --no-kernel-events --no-sweep --inflight $n 2>/dev/null | tail -1 > "$OUT/${R}_bench_inflight$n.json"; done
python3 "$ROOT/bench.py" --no-cpu-baseline --no-kernel-events --no-sweep 2>/dev/null | tail -1 > "$OUT/${R}_bench_no_events.json"
python3 "$ROOT/bench.py" --no-cpu-baseline --no-kernel-events --no-sweep --reuse-preprocessed 2>/dev/null | tail -1 > "$OUT/${R}_bench_reuse_preprocessed.json"
# 5. one proof over N ranks of this one GPU (local shard group): replicated vs divided work; Poseidon252 variant (config 5) and its bound
python3 "$ROOT/tools/shard_local.py" 5 > "$OUT/${R}_shard_local_one_gpu.json" 2> "$OUT/shard_local.err"
python3 "$ROOT/tools/poseidon_trace.py" 24 2 > "$OUT/${R}_poseidon_trace_2p24.json" 2> "$OUT/poseidon_trace.err"
hipcc -O3 -std=c++17 --offload-arch=gfx950 -I "$ROOT/stwo-brainfuck_amd/csrc" -o /tmp/ubench_poseidon "$ROOT/tools/ubench_poseidon.hip" 2>/dev/null && /tmp/ubench_poseidon > "$OUT/${R}_ubench_poseidon.txt"
rm -rf /tmp/prof_pos; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pos -- python3 "$ROOT/tools/poseidon_trace.py" 22 1 > /dev/null 2>&1
cp $(ls /tmp/prof_pos/*/*kernel_stats.csv | head -1) "$OUT/${R}_poseidon_2p22_kernel_stats.csv"
ls -la "$OUT"
