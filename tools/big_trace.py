"""Synthetic large-trace run (BASELINE.json configs 3-5 family): a nested-counter program sized so that the Memory component reaches
2^(LOG_MAX_ROWS-4) table rows. Proves on the GPU with LOG_MAX_ROWS = 26 (needs the raised limit, SURVEY.md §8 d) and checks the proof
with the oracle verifier. Usage: python tools/big_trace.py [a] [b] [log_max_rows]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package, Oracle

def main():
    a = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 16000
    lmr = int(sys.argv[3]) if len(sys.argv) > 3 else 26
    code = "+" * a + "[>" + "+" * b + "[>+<-]<-]"
    pkg = load_package(); orc = Oracle()
    t0 = time.time(); ctx = pkg.Context(0, max_log_domain=lmr + 2); t_ctx = time.time() - t0
    t0 = time.time(); tr = pkg.Trace(ctx, code, b""); t_trace = time.time() - t0
    out = {"program": f"+*{a} [> +*{b} [>+<-]<-]", "vm_rows": tr.n_steps, "component_log_sizes": tr.log_sizes, "cells": tr.cells, "log_max_rows": lmr,
           "ctx_s": round(t_ctx, 3), "vm_tables_upload_s": round(t_trace, 3)}
    if max(tr.log_sizes) > lmr:
        print(json.dumps(out)); raise SystemExit("trace exceeds LOG_MAX_ROWS")
    times = []
    for _ in range(3):
        t0 = time.time(); proof, phases = tr.prove(lmr); times.append(time.time() - t0)
    out["prove_ms"] = [round(t * 1e3, 1) for t in times]
    out["cells_per_s"] = tr.cells / min(times)
    out["phases_ms"] = {k: round(v * 1e3, 1) for k, v in phases.items()}
    out["proof_bytes"] = len(proof)
    t0 = time.time(); ok, err = orc.verify(proof, lmr); out["oracle_verifier"] = [ok, err, round(time.time() - t0, 3)]
    print(json.dumps(out))
    ctx.close()

if __name__ == "__main__":
    main()
