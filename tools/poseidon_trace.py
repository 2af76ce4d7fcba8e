#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: a synthetic nested-counter trace of 2^log domain rows proved with the Poseidon252MerkleChannel variant;
ms per proof, Hades permutations per proof (counted from the launch shapes) and their rate; both verifiers must accept.
Usage: python tools/poseidon_trace.py [log=24] [steps=2]"""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package, Oracle


def main():
    log = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    pkg = load_package(); lib = pkg.lib()
    code = "+" * 14 + "[>" + "+" * (250 << (log - 20)) + "[>+<-]<-]"
    out = {"log_domain_rows": log, "log_max_rows": log}
    for name, conv in (("blake2s", (0, 0, 0, 0)), ("poseidon252", (0, 0, 0, 1))):
        pkg.set_default_conventions(*conv)
        c = pkg.Context(0, max_log_domain=log + 2)
        tr = pkg.Trace(c, code, b"")
        proof, _ = tr.prove(log)
        lib.bfhip_profile_enable(c._h, 2); lib.bfhip_profile_reset(c._h)
        c.sync(); t0 = time.perf_counter()
        for _ in range(steps):
            proof, phases = tr.prove(log)
        c.sync(); dt = (time.perf_counter() - t0) / steps
        js = ctypes.c_void_p(); lib.bfhip_profile_report(c._h, ctypes.byref(js)); rep = json.loads(ctypes.string_at(js).decode()); lib.bfhip_free_host(js)
        lib.bfhip_profile_enable(c._h, 0)
        row = {"cells": tr.cells, "ms_per_proof": round(dt * 1e3, 2), "cells_per_s": tr.cells / dt, "proof_bytes": len(proof),
               "own_verifier": pkg.verify_brainfuck(proof, log)[0], "phase_ms": {k: round(v * 1e3, 1) for k, v in phases.items()}}
        for k, v in rep.items():
            row[k] = {"ms_per_proof": round(v["total_ms"] / steps, 2), "hash_units_per_proof": round(v["units"] / steps), "G_units_per_s": round(v["units"] / v["total_ms"] / 1e6, 3)}
        if name == "poseidon252":
            orc = Oracle(); orc.set_conventions(*conv)
            ok, err = orc.verify(proof, log)
            row["oracle_verifier"] = bool(ok)
        out[name] = row
        tr.close(); c.close()
    pkg.set_default_conventions(0, 0, 0, 0)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
