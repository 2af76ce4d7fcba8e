#!/usr/bin/env python3
"""Generates tests/golden/fib19_lmr24_oracle_proof.json: size and SHA-256 of the proof the CPU oracle produces for
tests/golden/programs/fib19.bf at LOG_MAX_ROWS = 24 (BASELINE.json configs[1], the benchmark workload), one entry per convention set
(tests/conftest.py CONVENTIONS; include/bfhip.h `bfhip_conventions`).

The oracle needs ~10 minutes and ~20 GB for this size on 8 cores, so the digests are committed as a fixture; the -m gpu suite and
bench.py compare the device-resident proof against them (tests/test_gpu_prove.py::test_fib19_full_size_proof_matches_oracle_digest).
Run from the repository root:  python tests/golden/make_fib19_proof_digest.py stwo [rfc7693 ...]
"""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from conftest import CONVENTIONS, Oracle  # noqa: E402

if __name__ == "__main__":
    names = sys.argv[1:] or ["stwo"]
    path = os.path.join(HERE, "fib19_lmr24_oracle_proof.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    orc = Oracle()
    orc.L.orc_set_threads(os.cpu_count() or 1)
    code = open(os.path.join(HERE, "programs", "fib19.bf")).read()
    for name in names:
        orc.set_conventions(*CONVENTIONS[name])
        t0 = time.time()
        proof, _, _ = orc.prove(code, b"", log_max_rows=24)
        ok, err = orc.verify(proof, log_max_rows=24)
        assert ok, err
        doc[name] = {"program": "fib19.bf", "input": "", "log_max_rows": 24, "conventions": list(CONVENTIONS[name]), "proof_bytes": len(proof),
                     "sha256": hashlib.sha256(proof).hexdigest(), "generator": "oracle (oracle/libbforacle.so: orc_prove)",
                     "oracle_seconds": round(time.time() - t0, 1)}
        with open(path, "w") as f:
            json.dump(doc, f, indent=1)
            f.write("\n")
        print(name, doc[name])
