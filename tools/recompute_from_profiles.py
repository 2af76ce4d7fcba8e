#!/usr/bin/env python3
"""Recomputes the headline figures of a round from the raw files under profiles/ — the arithmetic a reviewer would do by hand:
  * k_merkle_layer: compressions x 977 lane-ops / kernel time, from the bench line's HIP events AND from the rocprofv3 summary of the same command;
    counter traffic vs algorithmic bytes per launch;
  * circle-FFT kernel run (128 x 2^24): rocprofv3 average duration / counter traffic per kernel = TB/s and fraction of 8 TB/s; algorithmic rate;
  * the 2^22-row point: ms per proof, k_merkle_layer share, busy fraction of the traced proof;
  * one proof over N ranks time-sharing one GPU: T(N) = N S + P, projection S + P / N (mean and fastest proofs); BASELINE config 5 literal shape;
  * Poseidon252: permutations per second.
  * (r05) the headline roofline two ways from ONE command on ONE stream: HIP events of the un-profiled run vs rocprofv3 --stats of the profiled run
    (must agree within 3 %), and the same pair with the default two streams (they do not: the measured reason is printed);
  * (r05) proofs in flight at the metric's size.
Usage: python3 tools/recompute_from_profiles.py [r05] > profiles/r05_recomputed.txt      (CPU only; reads nothing but profiles/)"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r05"
P = lambda name: os.path.join(ROOT, "profiles", f"{R}_{name}")
VALU_PEAK = 256 * 4 * 16 * 2.4e9          # lane-ops/s
OPS = 977
HBM = 8000.0                                # GB/s


def last_json_line(path):
    return json.loads(open(path).read().strip().split("\n")[-1])


def stats(path):
    out = {}
    for r in csv.DictReader(open(path)):
        n = r["Name"].split("(")[0].replace("void ", "").replace("bf::", "")
        out[n] = (int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]))
    return out


def main():
    b = last_json_line(P("bench.json"))
    ro = b["roofline"]
    comp = ro["compressions_per_proof"]
    print(f"== {R}: bench line ({b['config'].get('workload', '')[:60]})")
    print(f"value {b['value']:.4e} {b['unit']}, {b['ms_per_step']:.2f} ms per proof, parity_checked {b.get('parity_checked')}")
    ms = ro["kernels_ms_per_step"]["k_merkle_layer"]
    print(f"k_merkle_layer by HIP events: {ms:.2f} ms per proof, {comp} compressions -> {comp * OPS / (ms * 1e-3) / 1e12:.2f} T lane-ops/s = {comp * OPS / (ms * 1e-3) / VALU_PEAK:.3f} of {VALU_PEAK / 1e12:.2f} T (line says {ro['frac']})")
    if not os.path.exists(P("bench_kernel_stats.csv")):
        print("(no rocprofv3 summary of the bench command in this round's files)")
        return
    st = stats(P("bench_kernel_stats.csv"))
    c, tot, avg = st["k_merkle_layer"]
    launches_per_proof = ro["launches"] / b["steps"]
    proofs = c / launches_per_proof
    ms_p = tot / 1e6 / proofs
    print(f"k_merkle_layer by rocprofv3 --stats of the same command: {c} calls = {proofs:.1f} proofs x {launches_per_proof:.0f}, {ms_p:.2f} ms per proof, average launch {avg / 1e3:.1f} us "
          f"(un-profiled line: {ro['avg_launch_us']}) -> {comp * OPS / (ms_p * 1e-3) / VALU_PEAK:.3f} of the VALU peak under the profiler")
    pr = last_json_line(P("bench_under_rocprof.json"))["roofline"]
    print(f"   the profiled run's own HIP events: average launch {pr['avg_launch_us']} us, frac {pr['frac']}  (must agree with the CSV)")
    traffic = ro.get("traffic")
    if traffic is None and os.path.exists(P("pmc_traffic.json")):      # the line was printed before the round's counter passes were committed: read them directly
        traffic = json.load(open(P("pmc_traffic.json"))).get(ro["kernel"], {}).get("hbm_bytes_per_launch")
    print(f"HBM traffic per launch by the counters {traffic} B vs algorithmic {ro['hbm']['algorithmic_bytes_per_launch']} B = "
          f"{(traffic or 0) / ro['hbm']['algorithmic_bytes_per_launch']:.3f}x")
    mp = b["metric_point"]
    print(f"metric point (2^22 rows): {mp['ms_per_proof']} ms per proof = {mp['value']:.3e} cells/s; sweep {[(r['log_domain_rows'], r['ms_per_proof']) for r in b['sweep']]}")
    pl = b.get("pipelined") or {}
    print(f"Poseidon252 2^26 rows: {b['poseidon252']['ms_per_proof']} ms")
    for w in ("fib19", "2^22_rows", "2^20_rows"):
        if w in pl:
            print(f"proofs in flight, {w}: " + ", ".join((f"{k[-1]} in flight {pl[w][k]['ms_per_proof']} ms/proof" + (f" (x{pl[w][k]['gain_vs_1']}, same bytes {pl[w][k]['same_proof_as_1']})" if 'gain_vs_1' in pl[w][k] else ""))
                                                             if "ms_per_proof" in pl[w][k] else f"{k[-1]} in flight: {pl[w][k].get('error', '?')[:80]}" for k in ("in_flight_1", "in_flight_2", "in_flight_3")))
    cb = b["cpu_baseline"]
    print(f"cpu_baseline ({cb['kind']}): {cb['value']:.3e} cells/s with {cb.get('threads')} threads on {cb.get('cores_effective')} effective cores (quota {cb.get('quota_cores')}, affinity {cb.get('host_threads_in_affinity_mask')}); "
          f"proof identical to the GPU's: {cb.get('proof_identical_to_gpu')}; scalar port (not timed here): {cb.get('scalar_value', 0):.3e} cells/s")
    print("   north-star 10x: " + str(cb.get("north_star_10x"))[:200])
    if "simd_bound" in cb and "seconds_lower_bound" in cb["simd_bound"]:
        print(f"   simd_bound (floor of hashing + transforms on the vector units): {cb['simd_bound']['seconds_lower_bound']['total']:.3f} s as run on the granted cores")

    if os.path.exists(P("roofline_single_stream_kernel_stats.csv")):
        print(f"\n== {R}: headline roofline, reproducible (bench.py --steps 20 --warmup 5 --no-sweep --no-poseidon --no-cpu-baseline)")
        for tag in ("single_stream", "two_streams"):
            ev = last_json_line(P(f"roofline_{tag}_events.json")); pr = last_json_line(P(f"roofline_{tag}_under_rocprof.json"))
            st = stats(P(f"roofline_{tag}_kernel_stats.csv"))
            c, tot, avg = st["k_merkle_layer"]
            ro = ev["roofline"]
            lpp = ro["launches"] / ev["steps"]
            f_csv = ro["compressions_per_proof"] * OPS / (avg * 1e-9 * lpp) / VALU_PEAK
            all_ms = sum(v[1] for v in st.values()) / 1e6 / (c / lpp)
            print(f"{tag:14s} un-profiled: {ev['ms_per_step']:.2f} ms/proof, HIP events {ro['avg_launch_us']} us/launch -> frac {ro['frac']};  under rocprofv3: {pr['ms_per_step']:.2f} ms/proof, "
                  f"CSV {c} launches avg {avg / 1e3:.2f} us -> frac_rocprof {f_csv:.4f} (that run's own events: {pr['roofline']['frac']});  difference {100 * (f_csv / ro['frac'] - 1):+.1f} %;  "
                  f"sum of all kernels under the profiler {all_ms:.2f} ms per proof")
        print("   one stream: the two methods agree (< 3 %). Two streams: un-profiled, the side stream's kernels stretch the Merkle launches by ~1 %; under rocprofv3 the two queues"
              " co-run launch by launch and k_merkle_layer is stretched by ~12 % (the sum of kernel times exceeds the proof's wall time) — the profiler changes the sharing, not the kernel.")
    if os.path.exists(P("inflight.jsonl")):
        print(f"\n== {R}: proofs in flight (tools/inflight_profile.py, fresh contexts, un-profiled)")
        for line in open(P("inflight.jsonl")):
            d = json.loads(line)
            print(f"  {d['workload']:>6s}  {d['in_flight']} in flight: {d['ms_per_proof']} ms per proof, {d['cells_per_s']:.3e} cells/s, one SHA-256: {len(set(d['proof_sha256'])) == 1}")
        for k in (1, 2):
            if os.path.exists(P(f"2p22_inflight{k}_timeline_gaps.txt")):
                print(f"  2^22 rows, {k} in flight under rocprofv3: " + " | ".join(open(P(f'2p22_inflight{k}_timeline_gaps.txt')).read().split("\n")[:4]))

    if os.path.exists(P("2p22_kernel_stats.csv")):
        print(f"\n== {R}: 2^22-row point under rocprofv3")
        st = stats(P("2p22_kernel_stats.csv"))
        c, tot, _ = st["k_merkle_layer"]
        per = tot / 1e6 / (c / 53.0)
        print(f"k_merkle_layer {c} calls / 53 per proof = {c / 53:.0f} proofs, {per:.2f} ms per proof")
        print(open(P("2p22_timeline_gaps.txt")).read().split("\n")[0])

    if not os.path.exists(P("fft_kernel_stats.csv")):
        return
    print(f"\n== {R}: circle-FFT kernel run, 128 columns of 2^24 (rocprofv3 average duration / counter traffic)")
    st = stats(P("fft_kernel_stats.csv"))
    tr = json.load(open(P("fft_pmc_traffic.json")))
    for k, v in tr.items():
        if k.startswith("_") or k not in st or "fft" not in k:
            continue
        calls, _, avg = st[k]
        gbs = v["hbm_bytes_per_launch"] / avg
        print(f"{k:28s} {calls:3d} launches, {avg / 1e3:8.1f} us, {v['hbm_bytes_per_launch'] / 1e9:6.2f} GB -> {gbs / 1e3:.2f} TB/s = {100 * gbs / HBM:.1f} % of 8 TB/s")
    rf = json.load(open(P("fft_roofline.json")))
    for r in rf:
        print(f"{r['columns']:4d} columns: iFFT + LDE + FFT {r['ifft_plus_lde_plus_fft_ms']} ms, algorithmic {r['algorithmic_GB/s']} GB/s = {100 * r['algorithmic_GB/s'] / HBM:.1f} % (HIP events)")

    if not os.path.exists(P("shard_local_one_gpu.json")):
        return
    print(f"\n== {R}: one proof over N ranks time-sharing ONE GPU (fib19)")
    sl = json.load(open(P("shard_local_one_gpu.json")))["runs"]
    t1, f1 = sl[0]["ms_per_proof"], sl[0].get("ms_fastest_proof", sl[0]["ms_per_proof"])
    for r in sl[1:]:
        n = r["ranks_on_one_gpu"]
        s = (r["ms_per_proof"] - t1) / (n - 1)
        f = r.get("ms_fastest_proof", r["ms_per_proof"])
        sf = (f - f1) / (n - 1)
        print(f"N = {n}: T = {r['ms_per_proof']} ms (fastest {f}); S = {s:.2f} ({sf:.2f}), P = {t1 - s:.2f}; projected with one GPU per rank {s + (t1 - s) / n:.2f} ({sf + (f1 - sf) / n:.2f}) ms")
    log = open(P("config5_literal.log")).read() if os.path.exists(P("config5_literal.log")) else ""
    t = dict((int(m.group(1)), float(m.group(2))) for m in re.finditer(r"2\^26, (\d+) ranks: ([\d.]+) ms", log))
    if 1 in t and 8 in t:
        s = (t[8] - t[1]) / 7
        print(f"config 5 literal (8 ranks x 2^26 rows x Poseidon252): T(8) = {t[8]:.0f} ms, T(1) = {t[1]:.0f} ms -> S = {s:.1f} ms, projected {s + (t[1] - s) / 8:.0f} ms = {t[1] / (s + (t[1] - s) / 8):.2f}x")
        print("   same SHA-256: " + str(len(set(re.findall(r"[0-9a-f]{64}", log))) == 1))

    if not os.path.exists(P("poseidon_trace_2p24.json")):
        return
    print(f"\n== {R}: Poseidon252")
    pt = json.load(open(P("poseidon_trace_2p24.json")))["poseidon252"]
    print(f"2^24 rows: {pt['ms_per_proof']} ms per proof; " + (f"layer kernel {pt['k_merkle_layer_poseidon']['G_units_per_s']} G permutations/s" if 'k_merkle_layer_poseidon' in pt else ""))
    if os.path.exists(P("ubench_poseidon.txt")):
        print(open(P("ubench_poseidon.txt")).read().strip())


if __name__ == "__main__":
    main()
