#!/bin/bash
# Regenerates the measurement artefacts of a round on the GPU box (run through gpurun from the repository root):
#   gpurun -- 'bash tools/profile_round.sh r04 [part ...]'      parts: bench roofline trace inflight pmc clock fft poseidon shard latency misc   (default: all)
# Writes into gpurun_out/<round>/ ; copy what should be judged into profiles/.
set -u
R=${1:-r06}; shift || true
PARTS=${*:-bench roofline trace inflight pmc clock fft poseidon shard latency misc}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
kt() { ls $1/*/*kernel_trace.csv | head -1; }
# every profiled command is bounded: rocprofv3 with a multi-threaded program can hang (it did, once, with eight rank threads) and a call that
# runs into gpurun's own limit costs the whole budget of that call
RP="timeout 900 rocprofv3"

if has bench; then
# 1. headline bench line (roofline measured live with HIP events, CPU baseline on the host cores)
python3 "$ROOT/bench.py" > "$OUT/bench.log" 2>&1; grep '^{"metric"' "$OUT/bench.log" | tail -1 > "$OUT/${R}_bench.json"
# 2. same command under rocprofv3 --kernel-trace --stats (per-kernel average durations must agree with the roofline object)
rm -rf /tmp/prof_stats; $RP --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-sweep --no-poseidon --no-clock-probe > "$OUT/bench_under_rocprof.log" 2>&1
grep '^{"metric"' "$OUT/bench_under_rocprof.log" | tail -1 > "$OUT/${R}_bench_under_rocprof.json"
cp $(ls /tmp/prof_stats/*/*kernel_stats.csv | head -1) "$OUT/${R}_bench_kernel_stats.csv"
fi

if has roofline; then
# 2a. the headline roofline, reproducible: the SAME command (20 steps, 5 warm-up, nothing else) un-profiled and under rocprofv3 --kernel-trace --stats, on ONE
#     stream (BFHIP_SINGLE_STREAM=1: the preprocessed tree's Blake2s does not co-run with the main-trace transforms, so a launch's duration does not
#     depend on how the profiler lets two queues share the GPU) and, beside it, with the default two streams. tools/recompute_from_profiles.py reads them.
for ss in 1 0; do
  tag=$([ $ss = 1 ] && echo single_stream || echo two_streams)
  BFHIP_SINGLE_STREAM=$ss python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-sweep --no-poseidon --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > "$OUT/${R}_roofline_${tag}_events.json"
  rm -rf /tmp/prof_rf; BFHIP_SINGLE_STREAM=$ss $RP --kernel-trace --stats --output-format csv -d /tmp/prof_rf -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-sweep --no-poseidon --no-cpu-baseline --no-clock-probe > "$OUT/roofline_${tag}_under_rocprof.log" 2>&1      # no probe: its k_merkle_layer launches would join the CSV's average
  grep '^{"metric"' "$OUT/roofline_${tag}_under_rocprof.log" | tail -1 > "$OUT/${R}_roofline_${tag}_under_rocprof.json"
  cp $(ls /tmp/prof_rf/*/*kernel_stats.csv | head -1) "$OUT/${R}_roofline_${tag}_kernel_stats.csv"
  python3 "$ROOT/tools/merkle_launches.py" $(kt /tmp/prof_rf) > "$OUT/${R}_roofline_${tag}_merkle_launches.txt" 2>&1
done
fi

if has inflight; then
# 2c. proofs in flight behind ONE caller thread (r06: the library's pool, bfhip_prove_batch): un-profiled ms per proof for pools of 1 / 2 / 3 sub-contexts and
#     the three ways the pool treats the preprocessed tree, at the metric's size, 2^20 rows and fib19; beside it the round-5 pattern (k Python threads over k
#     full contexts) at 2^22 rows; then the kernel trace of a pool of 2: how much of the time launches of BOTH proofs are in flight
( python3 "$ROOT/tools/pool_rate.py" 22 --in-flight 1,2,3 --preprocessed 0,1,2; python3 "$ROOT/tools/pool_rate.py" 20 --in-flight 2,3 --preprocessed 0,1 --batch 24;
  python3 "$ROOT/tools/pool_rate.py" fib19 --in-flight 2,3 --preprocessed 0,1 --batch 6; python3 "$ROOT/tools/pool_rate.py" 24 --in-flight 2,3 --preprocessed 1 --batch 6 ) > "$OUT/${R}_pool_rate.jsonl" 2>/dev/null
for k in 1 2 3; do python3 "$ROOT/tools/inflight_profile.py" 22 $k --rounds 8; done > "$OUT/${R}_inflight_threads.jsonl" 2>/dev/null
# host-inclusive: batches of PROGRAMS (VM + tables + upload inside the workers) — fib19 and the 2^22-row synthetic program, pools of 1 / 2 / 3
( POOL_RATE_PROGRAMS=1 python3 "$ROOT/tools/pool_rate.py" fib19 --in-flight 1,2,3 --preprocessed 1 --batch 6 --batches 3; POOL_RATE_PROGRAMS=1 python3 "$ROOT/tools/pool_rate.py" 22 --in-flight 1,3 --preprocessed 1 --batch 12 --batches 3 ) > "$OUT/${R}_pool_rate_programs.jsonl" 2>/dev/null
rm -rf /tmp/prof_pool; $RP --kernel-trace --output-format csv -d /tmp/prof_pool -- python3 "$ROOT/tools/pool_rate.py" 22 --child 2,1 --batch 12 --batches 3 > "$OUT/pool2_under_rocprof.json" 2>/dev/null
python3 "$ROOT/tools/timeline_gaps.py" $(kt /tmp/prof_pool) --window 0.55:0.90 > "$OUT/${R}_2p22_pool2_timeline_gaps.txt" 2>&1
fi

if has trace; then
# 2b. idle-gap analysis of one proof without the event instrumentation: the bench workload and the metric's own size (2^22 rows), 2^20
rm -rf /tmp/prof_tl; $RP --kernel-trace --output-format csv -d /tmp/prof_tl -- python3 "$ROOT/tools/point.py" fib19 --steps 3 --warmup 1 > /dev/null 2>&1
python3 "$ROOT/tools/timeline_gaps.py" $(kt /tmp/prof_tl) 12 > "$OUT/${R}_timeline_gaps.txt" 2>&1
python3 "$ROOT/tools/timeline_dump.py" $(kt /tmp/prof_tl) > "$OUT/${R}_fib19_timeline.txt" 2>&1
python3 "$ROOT/tools/fft_launches.py" $(kt /tmp/prof_tl) > "$OUT/${R}_fft_launches.txt" 2>&1
for K in k_quotients k_constraints k_fold k_eval; do python3 "$ROOT/tools/fft_launches.py" $(kt /tmp/prof_tl) $K | head -12; done > "$OUT/${R}_field_kernel_launches.txt" 2>&1
for w in 22 20; do
  rm -rf /tmp/prof_$w; $RP --kernel-trace --stats --output-format csv -d /tmp/prof_$w -- python3 "$ROOT/tools/point.py" $w --steps 10 --warmup 2 > "$OUT/point_${w}_under_rocprof.json" 2>/dev/null
  cp $(ls /tmp/prof_$w/*/*kernel_stats.csv | head -1) "$OUT/${R}_2p${w}_kernel_stats.csv"
  python3 "$ROOT/tools/timeline_gaps.py" $(kt /tmp/prof_$w) 15 > "$OUT/${R}_2p${w}_timeline_gaps.txt" 2>&1
  python3 "$ROOT/tools/timeline_dump.py" $(kt /tmp/prof_$w) --summary > "$OUT/${R}_2p${w}_timeline_summary.txt" 2>&1
  python3 "$ROOT/tools/timeline_dump.py" $(kt /tmp/prof_$w) > "$OUT/${R}_2p${w}_timeline.txt" 2>&1
  # the same with the mailbox order forced on and off (the default is on for LOG_MAX_ROWS <= 21): under the profiler the host is ~3x slower at
  # launching, so the mailbox kernels wait for it far longer than they do in an un-profiled run (r04_host_round_trips.txt has those waits)
  for mbx in 1 0; do
    export BFHIP_MAILBOX=$mbx; tag=$([ $mbx = 1 ] && echo on || echo off)
    rm -rf /tmp/prof_${w}_mb; $RP --kernel-trace --output-format csv -d /tmp/prof_${w}_mb -- python3 "$ROOT/tools/point.py" $w --steps 3 --warmup 1 > /dev/null 2>&1
    python3 "$ROOT/tools/timeline_gaps.py" $(kt /tmp/prof_${w}_mb) 15 > "$OUT/${R}_2p${w}_mailbox_${tag}_timeline_gaps.txt" 2>&1
  done
  unset BFHIP_MAILBOX
done
for w in 20 21 22 23 24 25 26 fib19; do python3 "$ROOT/tools/point.py" $w --steps 10; done > "$OUT/${R}_points.jsonl" 2>/dev/null
# overlap switches on this box (A/B), concurrent k_merkle_layer + k_quotients visible in the timeline with bit 1
for rep in 1 2; do for ov in 0 1 2 3; do for w in 22 fib19; do echo -n "overlap=$ov $w: "; BFHIP_OVERLAP=$ov python3 "$ROOT/tools/point.py" $w --steps 20 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_proof'], d['ms_min'], d['proof_sha256'][:12])"; done; done; done > "$OUT/${R}_overlap_ab.txt" 2>&1
fi

if has pmc; then
# 3. HBM traffic counters, one pass each, kernel trace only
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$C; $RP --kernel-trace --pmc $C --output-format csv -d /tmp/prof_$C -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-sweep --no-poseidon > "$OUT/pmc_$C.log" 2>&1
done
python3 "$ROOT/tools/pmc_traffic.py" /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE > "$OUT/${R}_pmc_traffic.json"
fi

CNT="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"
if has clock; then
# 3b. effective clock and issue-slot accounting of the Merkle kernel and of the register-only Blake2s micro-benchmark (one counter pass each)
rm -rf /tmp/prof_clock; $RP --kernel-trace --pmc $CNT --output-format csv -d /tmp/prof_clock -- python3 "$ROOT/tools/merkle_shapes.py" > /dev/null 2>&1
python3 "$ROOT/tools/merkle_clock.py" /tmp/prof_clock > "$OUT/${R}_merkle_clock.json" 2>&1
python3 "$ROOT/tools/merkle_shapes.py" > "$OUT/${R}_merkle_shapes.txt" 2>&1
fi

if has fft; then
# 4. FFT kernel run (BASELINE config 3 (i)): HIP-event profile, rocprofv3 kernel stats of the same command, FETCH/WRITE counter passes
python3 "$ROOT/tools/fft_roofline.py" > "$OUT/${R}_fft_roofline.json" 2> "$OUT/fft_roofline.err"
rm -rf /tmp/prof_fft; $RP --kernel-trace --stats --output-format csv -d /tmp/prof_fft -- python3 "$ROOT/tools/fft_roofline.py" 24 128 > "$OUT/fft_roofline_under_rocprof.json" 2>/dev/null
cp $(ls /tmp/prof_fft/*/*kernel_stats.csv | head -1) "$OUT/${R}_fft_kernel_stats.csv"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_fft_$C; $RP --kernel-trace --pmc $C --output-format csv -d /tmp/prof_fft_$C -- python3 "$ROOT/tools/fft_roofline.py" 24 128 > /dev/null 2>&1
done
python3 "$ROOT/tools/pmc_traffic.py" /tmp/prof_fft_FETCH_SIZE /tmp/prof_fft_WRITE_SIZE > "$OUT/${R}_fft_pmc_traffic.json"
# issue-slot / LDS counters of the FFT kernels (ISA audit): one counter pass over the same command
rm -rf /tmp/prof_fft_sq; $RP --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d /tmp/prof_fft_sq -- python3 "$ROOT/tools/fft_roofline.py" 24 128 > /dev/null 2>&1
python3 "$ROOT/tools/merkle_clock.py" /tmp/prof_fft_sq k_fft > "$OUT/${R}_fft_clock.json" 2>&1
( cd "$ROOT" && python3 tools/isa_mix.py stwo-brainfuck_amd/csrc/fft.hip k_fft_tile12 k_fft_strided7 k_fft_stridedK > "$OUT/${R}_fft_isa_mix.txt" 2>&1 )
for tp in 0 1; do for lg in 20 21 22; do echo "== two_pass=$tp log=$lg"; BFHIP_FFT_TWO_PASS=$tp python3 "$ROOT/tools/fft_roofline.py" $lg 4 32 | python3 -c "
import sys,json
for r in json.load(sys.stdin):
    print(r['columns'], 'columns:', r['ifft_plus_lde_plus_fft_ms'], 'ms', {k:(v['avg_us'],v['GB/s_moved']) for k,v in r.items() if isinstance(v,dict)})"; done; done > "$OUT/${R}_fft_two_pass_ab.txt" 2>&1
fi

if has poseidon; then
# 5. Poseidon252 variant (config 5): proof times, kernel stats, issue-slot counters of the layer kernel, register-only micro-benchmark
python3 "$ROOT/tools/poseidon_trace.py" 24 2 > "$OUT/${R}_poseidon_trace_2p24.json" 2> "$OUT/poseidon_trace.err"
hipcc -O3 -std=c++17 --offload-arch=gfx950 -I "$ROOT/stwo-brainfuck_amd/csrc" -o /tmp/ubench_poseidon "$ROOT/tools/ubench_poseidon.hip" 2>/dev/null && /tmp/ubench_poseidon > "$OUT/${R}_ubench_poseidon.txt"
rm -rf /tmp/prof_pos; $RP --kernel-trace --stats --output-format csv -d /tmp/prof_pos -- python3 "$ROOT/tools/poseidon_trace.py" 22 1 > /dev/null 2>&1
cp $(ls /tmp/prof_pos/*/*kernel_stats.csv | head -1) "$OUT/${R}_poseidon_2p22_kernel_stats.csv"
rm -rf /tmp/prof_posc; $RP --kernel-trace --pmc $CNT --output-format csv -d /tmp/prof_posc -- python3 "$ROOT/tools/poseidon_trace.py" 22 1 > /dev/null 2>&1
python3 "$ROOT/tools/merkle_clock.py" /tmp/prof_posc k_merkle_layer_poseidon > "$OUT/${R}_poseidon_clock.json" 2>&1
rm -rf /tmp/prof_posu; $RP --kernel-trace --pmc $CNT --output-format csv -d /tmp/prof_posu -- /tmp/ubench_poseidon > /dev/null 2>&1
python3 "$ROOT/tools/merkle_clock.py" /tmp/prof_posu k_hades > "$OUT/${R}_ubench_poseidon_clock.json" 2>&1
fi

if has shard; then
# 6. one proof over N ranks of this one GPU (local shard group): replicated vs divided work, per-collective GPU time; which kernels every rank
#    repeats (kernel traces of 1 and 8 ranks: plain, and serialised by a counter pass), idle time of the 8-rank proof; BASELINE config 5 in its
#    literal shape (8 ranks x 2^26 rows x Poseidon252) beside the same proof by one rank
python3 "$ROOT/tools/shard_local.py" 10 > "$OUT/${R}_shard_local_one_gpu.json" 2> "$OUT/shard_local.err"
SHARD_LOCAL_POSEIDON=1 SHARD_LOCAL_LOG=24 python3 "$ROOT/tools/shard_local.py" 2 > "$OUT/${R}_shard_local_poseidon_2p24.json" 2> "$OUT/shard_local_poseidon.err"
( cd "$ROOT" && bash tools/shard_audit.sh $R 8 fib19 > "$OUT/shard_audit_plain.log" 2>&1; AUDIT_PMC=1 bash tools/shard_audit.sh $R 8 fib19 > "$OUT/shard_audit_pmc.log" 2>&1; bash tools/config5_literal.sh $R > "$OUT/config5_literal.log" 2>&1 )
cd /tmp
fi

if has latency; then
# 7. the latency chains: host-side marks of the Fiat-Shamir round trips (BFHIP_TRACE_HOST), the small end of a tree in isolation
# (means over 20 proofs; the last five lines of the mailbox order: microseconds each mailbox kernel waited for the host, on the GPU's clock)
for w in 20 22 fib19; do for m in 0 1; do echo "== $w BFHIP_MAILBOX=$m"; BFHIP_MAILBOX=$m BFHIP_TRACE_HOST=2 python3 "$ROOT/tools/point.py" $w --steps 41 --warmup 2 2>&1 >/dev/null | grep "mean of 20" | tail -32; done; done > "$OUT/${R}_host_round_trips.txt" 2>&1
# mailboxes on/off on this box, alternating (ms per proof: loop mean, fastest proof)
for rep in 1 2 3; do for m in 0 1; do for w in 20 22 fib19; do echo -n "BFHIP_MAILBOX=$m $w: "; BFHIP_MAILBOX=$m python3 "$ROOT/tools/point.py" $w --steps 40 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_proof'], d['ms_min'], d['proof_sha256'][:12])"; done; done; done > "$OUT/${R}_mailbox_ab.txt" 2>&1
hipcc -O3 -std=c++17 --offload-arch=gfx950 -o /tmp/ubench_mailbox "$ROOT/tools/ubench_mailbox.hip" 2>/dev/null && timeout 120 /tmp/ubench_mailbox > "$OUT/${R}_ubench_mailbox.txt" 2>&1
hipcc -O3 -std=c++17 --offload-arch=gfx950 -I "$ROOT/stwo-brainfuck_amd/csrc" -o /tmp/ubench_tree_top "$ROOT/tools/ubench_tree_top.hip" 2>/dev/null && /tmp/ubench_tree_top > "$OUT/${R}_tree_small_end_ubench.txt" 2>&1
fi

if has misc; then
python3 "$ROOT/tools/trace_time.py" > "$OUT/${R}_one_call.txt" 2>&1
python3 "$ROOT/tools/big_trace.py" > "$OUT/${R}_big_trace.json" 2> "$OUT/big_trace.err"
for n in 2 3; do python3 "$ROOT/bench.py" --no-cpu-baseline --no-kernel-events --no-sweep --no-poseidon --inflight $n 2>/dev/null | tail -1 > "$OUT/${R}_bench_inflight$n.json"; done
python3 "$ROOT/bench.py" --no-cpu-baseline --no-kernel-events --no-sweep --no-poseidon 2>/dev/null | tail -1 > "$OUT/${R}_bench_no_events.json"
python3 "$ROOT/bench.py" --no-cpu-baseline --no-kernel-events --no-sweep --no-poseidon --reuse-preprocessed 2>/dev/null | tail -1 > "$OUT/${R}_bench_reuse_preprocessed.json"
fi
ls -la "$OUT"
