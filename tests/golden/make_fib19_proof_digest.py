#!/usr/bin/env python3
"""Generates tests/golden/fib19_lmr24_oracle_proof.json: size and SHA-256 of the proof the CPU oracle produces for
tests/golden/programs/fib19.bf at LOG_MAX_ROWS = 24 (BASELINE.json configs[1], the benchmark workload).

The oracle needs several minutes and ~20 GB for this size on 8 cores, so the digest is committed as a fixture; the -m gpu suite
compares the device-resident proof against it (tests/test_gpu_prove.py::test_fib19_full_size_proof_matches_oracle_digest).
Run from the repository root:  python tests/golden/make_fib19_proof_digest.py
"""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from conftest import Oracle  # noqa: E402

if __name__ == "__main__":
    orc = Oracle()
    orc.L.orc_set_threads(os.cpu_count() or 1)
    code = open(os.path.join(HERE, "programs", "fib19.bf")).read()
    t0 = time.time()
    proof, _, _ = orc.prove(code, b"", log_max_rows=24)
    ok, err = orc.verify(proof, log_max_rows=24)
    assert ok, err
    out = {"program": "fib19.bf", "input": "", "log_max_rows": 24, "proof_bytes": len(proof), "sha256": hashlib.sha256(proof).hexdigest(),
           "generator": "oracle (oracle/libbforacle.so: orc_prove)", "oracle_seconds": round(time.time() - t0, 1)}
    with open(os.path.join(HERE, "fib19_lmr24_oracle_proof.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(out)
