#!/usr/bin/env python3
"""Recomputes the headline figures of a round from the raw files under profiles/ — the arithmetic a reviewer would do by hand:
  * k_merkle_layer: compressions x 977 lane-ops / kernel time, from the bench line's HIP events AND from the rocprofv3 summary of the same command;
    counter traffic vs algorithmic bytes per launch;
  * circle-FFT kernel run (128 x 2^24): rocprofv3 average duration / counter traffic per kernel = TB/s and fraction of 8 TB/s; algorithmic rate;
  * the 2^22-row point: ms per proof, k_merkle_layer share, busy fraction of the traced proof;
  * one proof over N ranks time-sharing one GPU: T(N) = N S + P, projection S + P / N (mean and fastest proofs); BASELINE config 5 literal shape;
  * Poseidon252: permutations per second.
Usage: python3 tools/recompute_from_profiles.py [r04] > profiles/r04_recomputed.txt      (CPU only; reads nothing but profiles/)"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r04"
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
    st = stats(P("bench_kernel_stats.csv"))
    c, tot, avg = st["k_merkle_layer"]
    launches_per_proof = ro["launches"] / b["steps"]
    proofs = c / launches_per_proof
    ms_p = tot / 1e6 / proofs
    print(f"k_merkle_layer by rocprofv3 --stats of the same command: {c} calls = {proofs:.1f} proofs x {launches_per_proof:.0f}, {ms_p:.2f} ms per proof, average launch {avg / 1e3:.1f} us "
          f"(un-profiled line: {ro['avg_launch_us']}) -> {comp * OPS / (ms_p * 1e-3) / VALU_PEAK:.3f} of the VALU peak under the profiler")
    pr = last_json_line(P("bench_under_rocprof.json"))["roofline"]
    print(f"   the profiled run's own HIP events: average launch {pr['avg_launch_us']} us, frac {pr['frac']}  (must agree with the CSV)")
    print(f"HBM traffic per launch by the counters {ro.get('traffic')} B vs algorithmic {ro['hbm']['algorithmic_bytes_per_launch']} B = "
          f"{(ro.get('traffic') or 0) / ro['hbm']['algorithmic_bytes_per_launch']:.3f}x")
    mp = b["metric_point"]
    print(f"metric point (2^22 rows): {mp['ms_per_proof']} ms per proof = {mp['value']:.3e} cells/s; sweep {[(r['log_domain_rows'], r['ms_per_proof']) for r in b['sweep']]}")
    print(f"Poseidon252 2^26 rows: {b['poseidon252']['ms_per_proof']} ms; two proofs in flight: {b['pipelined']['ms_per_proof']} ms per proof")
    cb = b["cpu_baseline"]
    print(f"cpu_baseline ({cb['kind']}, {cb['cores']} threads): {cb['value']:.3e} cells/s -> GPU / CPU = {b['value'] / cb['value']:.0f}x; simd_bound: {cb['simd_bound']['seconds_lower_bound']['total']:.3f} s "
          f"-> {cb['simd_bound']['gpu_over_simd_bound_as_measured_on_all_threads']}x on the granted cores, {cb['simd_bound']['gpu_over_simd_bound']}x against every physical core")

    print(f"\n== {R}: 2^22-row point under rocprofv3")
    st = stats(P("2p22_kernel_stats.csv"))
    c, tot, _ = st["k_merkle_layer"]
    per = tot / 1e6 / (c / 53.0)
    print(f"k_merkle_layer {c} calls / 53 per proof = {c / 53:.0f} proofs, {per:.2f} ms per proof")
    print(open(P("2p22_timeline_gaps.txt")).read().split("\n")[0])

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

    print(f"\n== {R}: one proof over N ranks time-sharing ONE GPU (fib19)")
    sl = json.load(open(P("shard_local_one_gpu.json")))["runs"]
    t1, f1 = sl[0]["ms_per_proof"], sl[0].get("ms_fastest_proof", sl[0]["ms_per_proof"])
    for r in sl[1:]:
        n = r["ranks_on_one_gpu"]
        s = (r["ms_per_proof"] - t1) / (n - 1)
        f = r.get("ms_fastest_proof", r["ms_per_proof"])
        sf = (f - f1) / (n - 1)
        print(f"N = {n}: T = {r['ms_per_proof']} ms (fastest {f}); S = {s:.2f} ({sf:.2f}), P = {t1 - s:.2f}; projected with one GPU per rank {s + (t1 - s) / n:.2f} ({sf + (f1 - sf) / n:.2f}) ms")
    log = open(P("config5_literal.log")).read()
    t = dict((int(m.group(1)), float(m.group(2))) for m in re.finditer(r"2\^26, (\d+) ranks: ([\d.]+) ms", log))
    if 1 in t and 8 in t:
        s = (t[8] - t[1]) / 7
        print(f"config 5 literal (8 ranks x 2^26 rows x Poseidon252): T(8) = {t[8]:.0f} ms, T(1) = {t[1]:.0f} ms -> S = {s:.1f} ms, projected {s + (t[1] - s) / 8:.0f} ms = {t[1] / (s + (t[1] - s) / 8):.2f}x")
        print("   same SHA-256: " + str(len(set(re.findall(r"[0-9a-f]{64}", log))) == 1))

    print(f"\n== {R}: Poseidon252")
    pt = json.load(open(P("poseidon_trace_2p24.json")))["poseidon252"]
    print(f"2^24 rows: {pt['ms_per_proof']} ms per proof; " + (f"layer kernel {pt['k_merkle_layer_poseidon']['G_units_per_s']} G permutations/s" if 'k_merkle_layer_poseidon' in pt else ""))
    print(open(P("ubench_poseidon.txt")).read().strip())


if __name__ == "__main__":
    main()
