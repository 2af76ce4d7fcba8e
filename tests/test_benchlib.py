"""CPU: the arithmetic of tools/benchlib (what bench.py's line is assembled from) on canned records — the roofline object from the library's HIP-event report, the
sustained-clock fields, the FFT kernels against both of their bounds, config.batch from the proofs-in-flight table. No GPU: fake contexts and fixed numbers that a
reader can redo by hand."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.benchlib import probes, roofline, workloads      # noqa: E402


def test_dominant_roofline_is_the_valu_roofline_of_the_merkle_kernel():
    # 10 timed proofs: 780 launches, 6 697 779 200 compressions in 184.35 ms, 5.219 MB algorithmic per launch
    rep = {"k_merkle_layer": {"calls": 780, "total_ms": 184.35, "bytes": 780 * 521.9e6, "units": 6697779200.0, "aux": 0.0}}
    r = roofline.dominant_roofline(rep, steps=10)
    tops = 6697779200.0 * 977 / 0.18435 / 1e12
    assert r["kernel"] == "k_merkle_layer" and r["bound"] == "valu" and r["compressions_per_proof"] == 669777920 and r["launches"] == 780
    assert abs(r["achieved"] - tops) < 0.01 and abs(r["frac"] - tops / (256 * 4 * 16 * 2.4e9 / 1e12)) < 1e-3 and abs(r["avg_launch_us"] - 236.35) < 0.01
    assert abs(r["hbm"]["achieved"] - 521.9e6 / 236.35e-6 / 1e9) < 1.0 and r["hbm"]["peak"] == 8000.0
    assert len(r["kernel_sources_sha256"]) == 64 and "frac_rocprof" in r and "traffic_source" in r
    # a sharded line never borrows the one-GPU rocprofv3 launch average
    s = roofline.dominant_roofline(rep, steps=10, sharded_world=8)
    assert s["frac_rocprof"] is None and "share" in s["frac_rocprof_source"] and "1 of 8 ranks" in s["scope"]


def test_sustained_clock_fields():
    class Ctx:
        def __init__(self, ghz, mix_ghz=None, mix_rate=35.8, spanned=True):
            self.ghz, self.mix_ghz, self.mix_rate, self.spanned = ghz, mix_ghz or ghz, mix_rate, spanned
        def clock_probe(self, seconds): return {"ghz": self.ghz, "ghz_min": self.ghz - 0.03, "ghz_max": self.ghz + 0.01, "G_compressions_per_s": 39.9 * self.ghz / 2.4, "launches": 32, "ms_per_launch": 27.0}
        def clock_probe_mix(self, seconds, log_nodes=22): return {"ghz": self.mix_ghz, "G_compressions_per_s": self.mix_rate, "launches": 4288.0, "us_per_launch": 117.0, "sampler_spanned_the_window": self.spanned, "sampler_seconds": 0.5}
    base = {"bound": "valu", "achieved": 32.75, "frac": 0.833}
    slow = roofline.add_sustained_clock(dict(base), Ctx(2.19, mix_rate=32.0))            # a device that holds 2.19 GHz under both probes
    assert slow["sustained_clock_ghz"] == 2.19 and abs(slow["frac_at_sustained_clock"] - 0.833 * 2.4 / 2.19) < 2e-3 and slow["clock_probe"]["device_is_slow"] is True
    assert "clock" in slow["clock_probe"]["device_is_slow_because"]
    fast = roofline.add_sustained_clock(dict(base, achieved=35.85, frac=0.9117), Ctx(2.386))
    assert fast["clock_probe"]["device_is_slow"] is False and fast["clock_probe"]["device_is_slow_because"] is None and abs(fast["frac_at_sustained_clock"] - 0.9117 * 2.4 / 2.386) < 2e-3
    assert 0.99 < fast["clock_probe"]["probe_frac_at_its_clock"] < 1.0           # 39.9 G compressions/s x 977 ops at 2.4 GHz = 0.991 of the issue slots
    # round 6's second kind of slow box: both clocks hold, the REAL kernel is 6 % slower on the fixed shape
    mem = roofline.add_sustained_clock(dict(base, achieved=33.6, frac=0.8554), Ctx(2.392, mix_ghz=2.168, mix_rate=33.53))
    assert mem["sustained_clock_ghz"] == 2.168 and mem["clock_probe"]["device_is_slow"] is True and "real HBM traffic" in mem["clock_probe"]["device_is_slow_because"]
    assert mem["clock_probe"]["merkle_kernel"]["vs_builder_boxes"] == round(33.53 / 35.8, 4) and abs(mem["frac_at_sustained_clock"] - 0.8554 * 2.4 / 2.168) < 2e-3
    ok = roofline.add_sustained_clock(dict(base, achieved=36.0, frac=0.9156), Ctx(2.398, mix_ghz=2.317, mix_rate=36.15))      # the fastest box seen: the two fractions at their clocks agree (0.947)
    assert ok["clock_probe"]["device_is_slow"] is False and abs(ok["frac_at_sustained_clock"] - mem["frac_at_sustained_clock"]) < 0.01
    # the sampler that did not span its window is not trusted: the register-only clock stands
    ns = roofline.add_sustained_clock(dict(base), Ctx(2.38, mix_ghz=1.0, spanned=False))
    assert ns["sustained_clock_ghz"] == 2.38
    class Broken:
        def clock_probe(self, seconds): raise RuntimeError("no probe")
    b = roofline.add_sustained_clock(dict(base), Broken())                      # the probe must never cost the line
    assert b["sustained_clock_ghz"] is None and "no probe" in b["clock_probe"]["error"]
    assert roofline.add_sustained_clock(None, Ctx(2.4)) is None and "sustained_clock_ghz" not in roofline.add_sustained_clock({"bound": "hbm"}, Ctx(2.4))


def test_fft_report_prices_every_kernel_against_both_bounds():
    full = {"k_fft_tile12<false>": {"calls": 10, "total_ms": 1.754, "bytes": 1.754e-3 * 2765e9, "units": 1.0e9, "aux": 3293075456.0},
            "k_fft_strided7<true>": {"calls": 6, "total_ms": 0.802, "bytes": 0.802e-3 * 4825e9, "units": 0.5e9, "aux": 1692106752.0},
            "k_quotients": {"calls": 2, "total_ms": 1.7, "bytes": 0.0, "units": 0.0, "aux": 0.0}}
    f = roofline.fft_report(full)
    assert set(f["kernels"]) == {"k_fft_tile12<false>", "k_fft_strided7<true>"}
    t = f["kernels"]["k_fft_tile12<false>"]
    assert abs(t["valu_frac"] - 3293075456 * 11 / 1.754e-3 / 39.3216e12) < 1e-3 and abs(t["moved_frac_of_hbm_peak"] - 2765 / 8000) < 1e-3 and t["butterflies"] == 3293075456
    assert abs(f["valu_frac"] - (3293075456 + 1692106752) * 11 / 2.556e-3 / 39.3216e12) < 1e-3
    assert roofline.fft_report({"k_quotients": full["k_quotients"]}) is None


def test_batch_summary_of_the_metric_point():
    row = {"in_flight_1": {"ms_per_proof": 8.9, "cells_per_s": 1.62e10, "proof_sha256": ["ab"]},
           "in_flight_2": {"ms_per_proof": 7.1, "cells_per_s": 2.03e10, "gain_vs_1": 1.25, "same_proof_as_1": True, "proof_sha256": ["ab", "ab"], "recommitted": {"ms_per_proof": 7.8, "cells_per_s": 1.85e10, "gain_vs_1": 1.14}},
           "in_flight_3": {"ms_per_proof": 6.6, "cells_per_s": 2.18e10, "gain_vs_1": 1.35, "same_proof_as_1": True, "proof_sha256": ["ab"] * 3, "recommitted": {"error": "child exited 1"}}}
    b = probes.batch_summary({"2^22_rows": row, "fib19": {}})
    assert b["ms_per_proof"] == 6.6 and b["cells_per_s"] == 2.18e10 and b["one_at_a_time"]["ms_per_proof"] == 8.9
    assert b["in_flight_2"]["shared_preprocessed"] == {"ms_per_proof": 7.1, "cells_per_s": 2.03e10, "gain_vs_1": 1.25, "same_proof_as_1": True}
    assert b["in_flight_2"]["recommitted_preprocessed"]["ms_per_proof"] == 7.8 and b["in_flight_3"]["recommitted_preprocessed"] == {"error": "child exited 1"}
    assert "proof_sha256" not in str(b["in_flight_2"]) and "one caller thread" in b["what"].lower()
    assert probes.batch_summary(None) is None and probes.batch_summary({"error": "x"}) is None


def test_workloads_and_digests():
    assert workloads.sweep_program(22).count("+") == 14 + (250 << 2) + 1 and workloads.sweep_program(20).startswith("+" * 14 + "[>")
    d = workloads.want_digest((0, 0, 0, 0), 24)
    assert d is not None and len(d["sha256"]) == 64 and workloads.want_digest((9, 9, 9, 9), 24) is None
    assert workloads.pick_device(3, 8) == 3 and workloads.pick_device(3, 1) == 0 and workloads.pick_device(5, 8, override=0) == 0
    assert os.path.samefile(workloads.BENCH, os.path.join(ROOT, "bench.py"))
    import bench
    n = len(open(os.path.join(ROOT, "bench.py")).read().splitlines())
    assert n <= 400, f"bench.py has {n} lines: the contract path stays reviewable (VERDICT r05 item 6)"
    assert bench.FIB19 == workloads.FIB19
