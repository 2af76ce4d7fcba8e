#!/usr/bin/env python3
"""prove | verify over the C ABI — the two sub-commands of the reference's bin/brainfuck_prover.rs (prove: :79-139, verify: :141-151):
  bfprove.py prove  (--file prog.bf | --code '++>,<[>+.<-]') [--input-file in.bin] [--ram-size N] [--output proof.json] [--log-max-rows 24]
  bfprove.py verify proof.json [--log-max-rows 24]
The proof file is the serde_json form of BrainfuckProof (mod.rs:71-76), as the reference writes it (:129-131) and reads it (:146-151).
stdin supplies the program input when --input-file is absent (the reference's VM reads stdin)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    p = sub.add_parser("prove")
    p.add_argument("--file"); p.add_argument("--code"); p.add_argument("--input-file"); p.add_argument("--ram-size", type=int, default=0)
    p.add_argument("--output"); p.add_argument("--log-max-rows", type=int, default=24); p.add_argument("--poseidon252", action="store_true")
    v = sub.add_parser("verify")
    v.add_argument("proof"); v.add_argument("--log-max-rows", type=int, default=24); v.add_argument("--poseidon252", action="store_true")
    a = ap.parse_args()
    pkg = load_package()
    conv = (0, 0, 0, 1 if a.poseidon252 else 0)
    if a.cmd == "prove":
        code = open(a.file).read() if a.file else a.code
        if code is None:
            ap.error("prove needs --file or --code")
        inp = open(a.input_file, "rb").read() if a.input_file else (b"" if sys.stdin.isatty() else sys.stdin.buffer.read())
        pkg.set_default_conventions(*conv)
        ctx = pkg.Context(0, max_log_domain=a.log_max_rows + 2)
        t0 = time.time()
        tr = pkg.Trace(ctx, code, inp, ram_size=a.ram_size)
        t1 = time.time()
        proof, _ = tr.prove(a.log_max_rows)
        t2 = time.time()
        print(f"Steps: {tr.n_steps}; trace preparation {1e3 * (t1 - t0):.1f} ms; proof generation time: {t2 - t1:.3f}s; {len(proof)} bytes", file=sys.stderr)
        if a.output:
            open(a.output, "wb").write(proof)
        else:
            sys.stdout.buffer.write(proof)
        tr.close(); ctx.close()
        return 0
    ok, why = pkg.verify_brainfuck(open(a.proof, "rb").read(), a.log_max_rows, conventions=conv)
    print("Proof verified" if ok else f"Verification failed: {why}")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
