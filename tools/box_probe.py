#!/usr/bin/env python3
"""One compact line per box: fib19 ms per proof and the dominant kernel's VALU fraction (HIP events, one stream) beside the two clock probes — is a slow box visible in them?
   python3 tools/box_probe.py   (run through gpurun several times: every call lands on another box of the pool; lines collected in profiles/rNN_clock_probe_boxes.jsonl)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, BFHIP_SINGLE_STREAM="1")
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "15", "--warmup", "3", "--no-sweep", "--no-poseidon", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
rf = d["roofline"]
cp = rf["clock_probe"]
print(json.dumps({"ms_per_proof": round(d["ms_per_step"], 3), "frac": rf["frac"], "avg_launch_us": rf["avg_launch_us"], "frac_rocprof": rf.get("frac_rocprof"),
                  "sustained_clock_ghz": rf["sustained_clock_ghz"], "frac_at_sustained_clock": rf["frac_at_sustained_clock"],
                  "register_only": cp.get("register_only"), "merkle_kernel": {k: v for k, v in (cp.get("merkle_kernel") or {}).items() if k != "shape"},
                  "device_is_slow": cp.get("device_is_slow"), "because": cp.get("device_is_slow_because")}))
