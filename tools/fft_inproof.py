#!/usr/bin/env python3
"""The circle-FFT launches of ONE proof, priced one by one (BFHIP_FFT_PROF_DETAIL=1: the library's profiler records carry the launch shape): per launch
butterflies, bytes moved, time -> VALU fraction (butterflies x 11 lane-ops / time / 39.3 T) and fraction of HBM peak. Where the in-proof transforms lose
against the 128 x 2^24 kernel run (tools/fft_roofline.py).   python3 tools/fft_inproof.py [fib19|22|...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BFHIP_FFT_PROF_DETAIL"] = "1"
import bench  # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "fib19"
    pkg = bench.load_package()
    code, lmr = (bench.FIB19, 24) if what == "fib19" else (bench.sweep_program(int(what)), int(what))
    c = pkg.Context(0, max_log_domain=lmr + 2)
    tr = pkg.Trace(c, code, b"")
    lib = pkg.lib()
    for _ in range(2):
        tr.prove(lmr, want_json=False)
    lib.bfhip_profile_enable(c._h, 1); lib.bfhip_profile_reset(c._h)
    n = 5
    for _ in range(n):
        tr.prove(lmr, want_json=False)
    c.sync()
    rep = bench.profile_report(lib, c)
    lib.bfhip_profile_enable(c._h, 0)
    rows = []
    for k, v in rep.items():
        if not k.startswith("k_fft"):
            continue
        ms = v["total_ms"] / n
        rows.append({"launch": k, "per_proof": v["calls"] / n, "ms_per_proof": round(ms, 4), "butterflies": round(v["aux"] / n), "moved_GB": round(v["bytes"] / n / 1e9, 3),
                     "valu_frac": round(v["aux"] / n * 11 / (ms * 1e-3) / 39.3216e12, 3) if ms else None, "hbm_frac": round(v["bytes"] / n / (ms * 1e-3) / 8e12, 3) if ms else None})
    rows.sort(key=lambda r: -r["ms_per_proof"])
    tot = sum(r["ms_per_proof"] for r in rows); bf = sum(r["butterflies"] for r in rows)
    print(json.dumps({"workload": what, "fft_ms_per_proof": round(tot, 3), "butterflies": bf, "valu_frac": round(bf * 11 / (tot * 1e-3) / 39.3216e12, 3),
                      "ms_if_every_launch_ran_at_0.77": round(bf * 11 / (0.77 * 39.3216e12) * 1e3, 3), "launches": rows}, indent=1))
    tr.close(); c.close()


if __name__ == "__main__":
    main()
