"""Roofline arithmetic of the bench line (see tools/benchlib/__init__.py)."""
import ctypes
import json
import os

from .workloads import HBM_PEAK_GBS, ROOT, VALU_OPS_PER_COMPRESSION, VALU_PEAK_TOPS, kernel_sources_sha256


def roofline_from_committed_rocprof(compressions_per_proof, launches_per_proof):
    """frac_rocprof: k_merkle_layer's VALU fraction recomputed from the latest committed profiles/rNN_roofline_single_stream_kernel_stats.csv (average
    launch duration by rocprofv3) — next to `frac` (HIP events of THIS run). The JSON line of the profiled run lies beside the CSV and carries the
    SHA-256 of the kernel sources; a mismatch reports null with the reason. frac_range_this_round: min..max of `frac` over the round's committed lines."""
    import csv
    import glob
    out = {"frac_rocprof": None, "frac_rocprof_source": None, "frac_range_this_round": None}
    if not compressions_per_proof:
        return out
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_roofline_single_stream_kernel_stats.csv")))
    if not cands:
        out["frac_rocprof_source"] = "no committed rocprofv3 summary (tools/profile_round.sh rNN roofline)"
        return out
    path = cands[-1]
    rnd = os.path.basename(path).split("_")[0]
    try:
        line = json.loads(open(path.replace("_kernel_stats.csv", "_under_rocprof.json")).read().strip().split("\n")[-1])
        if line["roofline"].get("kernel_sources_sha256") != kernel_sources_sha256():
            out["frac_rocprof_source"] = os.path.relpath(path, ROOT) + " is STALE: taken on other kernel sources — rerun tools/profile_round.sh roofline"
            return out
        row = next(r for r in csv.DictReader(open(path)) if r["Name"].split("(")[0].replace("void ", "").replace("bf::", "") == "k_merkle_layer")
        avg_us = float(row["AverageNs"]) / 1e3
        ms_per_proof = avg_us * launches_per_proof / 1e3
        out["frac_rocprof"] = round(compressions_per_proof * VALU_OPS_PER_COMPRESSION / (ms_per_proof * 1e-3) / 1e12 / VALU_PEAK_TOPS, 4)
        out["avg_launch_us_rocprof"] = round(avg_us, 2)
        out["frac_rocprof_source"] = (os.path.relpath(path, ROOT) + f": {row['Calls']} launches, average {avg_us:.1f} us (rocprofv3 --kernel-trace --stats of bench.py --steps 20 --warmup 5 "
                                      "on one stream, same kernel sources; that run's own HIP events: frac " + str(line["roofline"].get("frac")) + ")")
        fr = []
        for f in glob.glob(os.path.join(ROOT, "profiles", rnd + "_*.json")):
            try:
                d = json.loads(open(f).read().strip().split("\n")[-1])
                if isinstance(d, dict) and isinstance(d.get("roofline"), dict) and d["roofline"].get("kernel") == "k_merkle_layer" and d.get("n_gpus") == 1:
                    fr.append(d["roofline"]["frac"])
            except Exception:
                pass
        if fr:
            out["frac_range_this_round"] = {"min": min(fr), "max": max(fr), "lines": len(fr), "what": f"`frac` (HIP events) over the {rnd} bench lines committed under profiles/ (boxes and run modes differ)"}
    except Exception as e:
        out["frac_rocprof_source"] = f"{os.path.relpath(path, ROOT)}: {e!r}"
    return out


def profile_report(lib, ctx):
    js = ctypes.c_void_p()
    lib.bfhip_profile_report(ctx._h, ctypes.byref(js))
    rep = json.loads(ctypes.string_at(js).decode())
    lib.bfhip_free_host(js)
    return rep


def point_roofline(pkg, c, tr, lmr, sec_per_proof):
    """The metric's own size (BASELINE 'at 2^22 rows'): kernel-time split, GPU-busy fraction and the dominant kernel's roofline of that proof, from
    two extra untimed proofs — one with the dominant kernel bracketed by HIP events per run of launches (as in the timed region of the main
    workload), one with every kernel bracketed (time split; the event pairs themselves stretch small proofs, so the busy fraction is the sum of
    the kernel times over the UN-instrumented wall time and is an upper estimate when streams overlap)."""
    lib = pkg.lib()
    lib.bfhip_profile_enable(c._h, 2); lib.bfhip_profile_reset(c._h)
    tr.prove(lmr, want_json=False); c.sync()
    dom = profile_report(lib, c)
    lib.bfhip_profile_enable(c._h, 1); lib.bfhip_profile_reset(c._h)
    tr.prove(lmr, want_json=False); c.sync()
    full = profile_report(lib, c)
    lib.bfhip_profile_enable(c._h, 0)
    out = {"kernels_ms_per_proof_instrumented": {k: round(v["total_ms"], 3) for k, v in sorted(full.items(), key=lambda kv: -kv[1]["total_ms"])}}
    tot = sum(v["total_ms"] for v in full.values())
    out["sum_of_kernel_ms"] = round(tot, 3)
    out["gpu_busy_frac_estimate"] = round(min(1.0, tot / (sec_per_proof * 1e3)), 3)
    d = dom.get("k_merkle_layer")
    if d and d.get("units", 0) > 0 and d["total_ms"] > 0:
        tops = d["units"] * VALU_OPS_PER_COMPRESSION / (d["total_ms"] * 1e-3) / 1e12
        out["roofline"] = {"kernel": "k_merkle_layer", "bound": "valu", "achieved": round(tops, 2), "peak": round(VALU_PEAK_TOPS, 2), "unit": "Tops/s (int32 VALU lane-ops)",
                           "frac": round(tops / VALU_PEAK_TOPS, 4), "launches": d["calls"], "avg_launch_us": round(d["total_ms"] / d["calls"] * 1e3, 2),
                           "compressions_per_proof": round(d["units"]), "kernel_ms_per_proof": round(d["total_ms"], 3),
                           "share_of_proof": round(d["total_ms"] / (sec_per_proof * 1e3), 3),
                           "hbm": {"achieved": round(d["bytes"] / d["total_ms"] / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(d["bytes"] / d["total_ms"] / 1e6 / HBM_PEAK_GBS, 4)}}
    return out


BUTTERFLY_VALU_OPS = 11      # canonical M31 butterfly: product 5 + add 3 + subtract 3 lane-ops (why it stays at 11)


def committed_pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the latest committed counter passes (profiles/rNN_pmc_traffic.json: separate rocprofv3 --pmc passes of the
    same command — the passes serialise dispatches and cannot run inside a timed region), trusted only while the kernel sources it was taken on are the
    ones that run now; otherwise (None, reason)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    pmc = json.load(open(files[-1]))
    rel = os.path.relpath(files[-1], ROOT)
    if pmc.get("_kernel_sources_sha256") == kernel_sources_sha256():
        return pmc.get(kernel, {}).get("hbm_bytes_per_launch"), rel + " (separate rocprofv3 --pmc passes of the same command on the same kernel sources; not collected in this run)"
    return None, rel + " is STALE: taken on other kernel sources (csrc/merkle.hip, csrc/kernels.h changed since) — traffic not reported; rerun tools/profile_round.sh pmc"


def dominant_roofline(rep, steps, sharded_world=0):
    """The `roofline` object of the line from the library's HIP-event records of the TIMED region (bfhip_profile_report): the kernel with the
    largest total time. k_merkle_layer is integer-VALU bound: ~977 lane-ops per Blake2s compression, one compression per 64
    message bytes, compressions counted from the launch shapes of this very run (prof.hip `units`); its HBM figure rides beside it."""
    name, d = max(rep.items(), key=lambda kv: kv[1]["total_ms"])
    avg_ms = d["total_ms"] / d["calls"]
    gbs = d["bytes"] / d["calls"] / (avg_ms * 1e-3) / 1e9
    traffic, traffic_src = committed_pmc_traffic(name)
    hbm = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": round(d["bytes"] / d["calls"])}
    common = {"kernel": name, "traffic": traffic, "traffic_source": traffic_src, "launches": d["calls"], "avg_launch_us": round(avg_ms * 1e3, 2),
              "kernels_ms_per_step": {k: round(v["total_ms"] / steps, 3) for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"])}}
    if name.startswith("k_merkle_layer") and d.get("units", 0) > 0:
        tops = d["units"] * VALU_OPS_PER_COMPRESSION / (d["total_ms"] * 1e-3) / 1e12
        out = {**common, "bound": "valu", "achieved": round(tops, 2), "peak": round(VALU_PEAK_TOPS, 2), "unit": "Tops/s (int32 VALU lane-ops)", "frac": round(tops / VALU_PEAK_TOPS, 4),
               "compressions_per_proof": round(d["units"] / steps), "G_compressions_per_s": round(d["units"] / (d["total_ms"] * 1e-3) / 1e9, 2),
               "valu_ops_per_compression": VALU_OPS_PER_COMPRESSION, "hbm": hbm}
    else:
        out = {**common, **hbm}
    out["kernel_sources_sha256"] = kernel_sources_sha256()
    if sharded_world:
        # a one-GPU rocprofv3 launch average says nothing about a rank's share of the launches: no frac_rocprof for a sharded line
        out.update({"frac_rocprof": None, "frac_rocprof_source": "not applicable: this line's launches are one rank's share of a sharded proof", "frac_range_this_round": None,
                    "scope": f"rank 0's share of the group's proofs (1 of {sharded_world} ranks): launches, compressions and kernel times are this rank's"})
    else:
        out.update(roofline_from_committed_rocprof(out.get("compressions_per_proof"), out["launches"] / steps))
    return out


# The real-kernel probe's shape: an inner layer of 2^24 nodes without columns — 1.5 GB of hashes in flight, i.e. REAL HBM traffic. profiles/r06_clock_probe_boxes.jsonl: on a
# cache-sized shape (2^22 nodes) and on the register-only loop every box of the pool reads the same 2.38-2.40 GHz, fast or slow; from 2^24 nodes up the clock a device holds
# under the kernel drops to 2.26-2.32 GHz on the boxes that prove fib19 in 28.2-28.4 ms and to 2.17 GHz on one that needs 30.05 ms, and the kernel's rate follows
# (35.5-36.2 against 33.5 G compressions/s). Reference = the builder's fast boxes.
MIX_PROBE_LOG = int(os.environ.get("BENCH_MIX_PROBE_LOG", "24"))
MIX_PROBE_REFERENCE_G = 35.8


def add_sustained_clock(roofline, ctx, seconds=0.6):
    """roofline.sustained_clock_ghz / frac_at_sustained_clock: `frac` is priced against the NOMINAL 2.4 GHz, and MI355X devices differ by up to 12 % in what they deliver on a
    compute-bound loop (MI355X_MICROARCH.md, DVFS give-back (5)) — a line at 0.83 can be a slow device or a regression. Two probes of the library, run right behind the timed region:
      register_only   bfhip_clock_probe: a register-only loop of the Merkle kernels' compression, every workgroup stamping s_memtime against the 100 MHz s_memrealtime;
      merkle_kernel   bfhip_clock_probe_mix: k_merkle_layer ITSELF on a fixed shape (2^22 inner nodes over pseudo-random hashes) launched back to back while a one-wave sampler
                      on the other stream stamps the two counters — the clock held under the real mix of VALU and memory traffic, and the kernel's rate on that shape.
    sustained_clock_ghz is the second (the first if the sampler did not span its window). device_is_slow: the real kernel runs the probe's shape more than 3.5 % below the
    builder's fast boxes (same kernel sources, so it is the device) or the register-only clock is below 0.95 x 2.4 GHz. A device can hold 2.38 GHz on the register-only loop
    and still be slow in a proof: what differs between boxes is the clock they hold once HBM traffic is real (r06: 2.32 against 2.17 GHz, proofs 28.2 against 30.1 ms)."""
    if not roofline or roofline.get("bound") != "valu":
        return roofline
    try:
        p = ctx.clock_probe(seconds)
        reg = {"ghz": round(p["ghz"], 3), "ghz_min": round(p["ghz_min"], 3), "ghz_max": round(p["ghz_max"], 3), "G_compressions_per_s": round(p["G_compressions_per_s"], 2),
               "frac_of_nominal_valu_peak": round(p["G_compressions_per_s"] * 1e9 * VALU_OPS_PER_COMPRESSION / 1e12 / VALU_PEAK_TOPS, 4),
               "frac_at_its_clock": round(p["G_compressions_per_s"] * 1e9 * VALU_OPS_PER_COMPRESSION / 1e12 / (VALU_PEAK_TOPS * p["ghz"] / 2.4), 4)}
        ghz, mix = p["ghz"], None
        try:
            m = ctx.clock_probe_mix(0.5, MIX_PROBE_LOG)
            mix = {"ghz": round(m["ghz"], 3), "sampler_spanned_the_window": m["sampler_spanned_the_window"], "shape": f"k_merkle_layer, inner layer of 2^{MIX_PROBE_LOG} nodes, no columns, pseudo-random hashes",
                   "G_compressions_per_s": round(m["G_compressions_per_s"], 2), "us_per_launch": round(m["us_per_launch"], 2),
                   "frac_of_nominal_valu_peak": round(m["G_compressions_per_s"] * 1e9 * VALU_OPS_PER_COMPRESSION / 1e12 / VALU_PEAK_TOPS, 4),
                   "builder_boxes_G_compressions_per_s": MIX_PROBE_REFERENCE_G, "vs_builder_boxes": round(m["G_compressions_per_s"] / MIX_PROBE_REFERENCE_G, 4)}
            if m["sampler_spanned_the_window"] and m["ghz"] > 0:
                ghz = m["ghz"]
        except Exception as e:
            mix = {"error": repr(e)}
        peak_here = VALU_PEAK_TOPS * ghz / 2.4
        roofline["sustained_clock_ghz"] = round(ghz, 3)
        roofline["frac_at_sustained_clock"] = round(roofline["achieved"] / peak_here, 4)
        slow_clock = p["ghz"] < 0.95 * 2.4
        slow_kernel = bool(mix and "vs_builder_boxes" in mix and mix["vs_builder_boxes"] < 0.965)
        roofline["clock_probe"] = {"what": "behind the timed region: (register_only) bfhip_clock_probe, %.1f s; (merkle_kernel) bfhip_clock_probe_mix, 0.5 s: the real kernel on a fixed shape beside a one-wave "
                                           "clock sampler; clock = d(s_memtime) / d(s_memrealtime) x 100 MHz" % seconds,
                                   "nominal_ghz": 2.4, "register_only": reg, "merkle_kernel": mix,
                                   # kept at the top level for readers of round-6 lines
                                   "ghz_min": reg["ghz_min"], "ghz_max": reg["ghz_max"], "G_compressions_per_s": reg["G_compressions_per_s"],
                                   "probe_frac_of_nominal_valu_peak": reg["frac_of_nominal_valu_peak"], "probe_frac_at_its_clock": reg["frac_at_its_clock"],
                                   "device_is_slow": bool(slow_clock or slow_kernel),
                                   "device_is_slow_because": ("the clock it sustains even on a register-only loop" if slow_clock else
                                                              "under real HBM traffic it holds %.2f GHz and runs the real kernel %.1f %% below the builder's fast boxes (2.26-2.32 GHz) at the same kernel sources"
                                                              % (ghz, 100 * (1 - mix["vs_builder_boxes"])) if slow_kernel else None)}
    except Exception as e:      # the probe must never cost the line
        roofline["sustained_clock_ghz"], roofline["frac_at_sustained_clock"], roofline["clock_probe"] = None, None, {"error": repr(e)}
    return roofline


def fft_report(full):
    """The circle-FFT kernels of one fully instrumented (untimed) proof against BOTH of their bounds: HBM bytes moved (north-star: >= 60 % on the FFT
    kernel) and integer VALU — butterflies x 11 lane-ops / time / 39.3 T: the LDS tile pass is issue-bound, not HBM-bound."""
    fk = {k: v for k, v in full.items() if k.startswith("k_fft")}
    if not fk:
        return None
    tot_ms = sum(v["total_ms"] for v in fk.values())
    valu = lambda v: round(v.get("aux", 0) * BUTTERFLY_VALU_OPS / (v["total_ms"] * 1e-3) / 1e12 / VALU_PEAK_TOPS, 4)      # noqa: E731
    return {"kernels": {k: {"ms_per_proof": round(v["total_ms"], 3), "launches": v["calls"], "moved_GBps": round(v["bytes"] / v["total_ms"] / 1e6, 1),
                            "moved_frac_of_hbm_peak": round(v["bytes"] / v["total_ms"] / 1e6 / HBM_PEAK_GBS, 4), "butterflies": round(v.get("aux", 0)), "valu_frac": valu(v)}
                        for k, v in sorted(fk.items())},
            "ms_per_proof": round(tot_ms, 3),
            "moved_GBps": round(sum(v["bytes"] for v in fk.values()) / tot_ms / 1e6, 1),
            "algorithmic_GBps": round(sum(v["units"] for v in fk.values()) / tot_ms / 1e6, 1),
            "algorithmic_frac_of_hbm_peak": round(sum(v["units"] for v in fk.values()) / tot_ms / 1e6 / HBM_PEAK_GBS, 4),
            "valu_frac": round(sum(v.get("aux", 0) for v in fk.values()) * BUTTERFLY_VALU_OPS / (tot_ms * 1e-3) / 1e12 / VALU_PEAK_TOPS, 4),
            "note": "in-proof mix of column sizes (most launches are small); algorithmic bytes = 8N per interpolated, 12N per extended column (SURVEY.md section 8(d)); valu_frac = butterflies x "
                    f"{BUTTERFLY_VALU_OPS} lane-ops / time / {VALU_PEAK_TOPS:.1f} T (nominal 2.4 GHz; addressing and twiddle loads not counted); the 128 x 2^24 kernel run is tools/fft_roofline.py -> profiles/"}
