#!/usr/bin/env python3
"""prove | verify over the C ABI — the two sub-commands of the reference's bin/brainfuck_prover.rs (prove: :79-139, verify: :141-151):

  bfprove.py prove  (--file prog.bf | --code '++>,<[>+.<-]') [--input-file in.bin] [--ram-size N] [--output proof.json]
                    [--log-max-rows 24] [--conventions a,b,c,d | --poseidon252] [--all-sets DIR]
  bfprove.py verify proof.json [--log-max-rows 24] [--conventions a,b,c,d | --poseidon252 | --try-all]

The proof file is the serde_json form of BrainfuckProof (mod.rs:71-76), as the reference writes it (:129-131) and reads it (:146-151).
stdin supplies the program input when --input-file is absent (the reference's VM reads stdin).

--conventions merkle_node_hash,mix_u64,logup_mask_order,merkle_channel — the byte-level switches of include/bfhip.h `bfhip_conventions`
(all 0 = the defaults). The two commands below are the one-step pin against the real stwo@31e8dbc for whoever has cargo:

  verify --try-all      a proof written by `brainfuck_prover prove --output` is checked under every switch set (8 Blake2s + 2 Poseidon252);
                        prints the accepting set(s), or for every set the first check that fails. Host only: no GPU needed.
  prove --all-sets DIR  writes this prover's proof of the program under each of the 8 Blake2s switch sets to DIR/proof_<a><b><c>0.json, for
                        `brainfuck_prover verify DIR/proof_*.json`: the file the reference accepts names the conventions of its stwo."""
import argparse
import itertools
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package

NAMES = ("merkle_node_hash", "mix_u64", "logup_mask_order", "merkle_channel")
VALUES = (("stwo-compress", "rfc7693"), ("compress", "hash"), ("[0,-1]", "[-1,0]"), ("blake2s", "poseidon252"))


def parse_conventions(a):
    if a.conventions:
        conv = tuple(int(v) for v in a.conventions.split(","))
        if len(conv) != 4 or any(v not in (0, 1) for v in conv):
            raise SystemExit("--conventions takes four 0/1 values: " + ",".join(NAMES))
        return conv
    return (0, 0, 0, 1 if a.poseidon252 else 0)


def describe(conv):
    return ", ".join(f"{n}={VALUES[i][v]}" for i, (n, v) in enumerate(zip(NAMES, conv)))


def all_sets(channels=(0, 1)):
    """The 8 Blake2s switch sets, and for the Poseidon252 channel the 2 that differ there (the node-hash and mix_u64 switches are Blake2s forms)."""
    sets = []
    for d in channels:
        for a, b, c in itertools.product((0, 1), repeat=3):
            if d == 0 or (a, b) == (0, 0):
                sets.append((a, b, c, d))
    return sets


def try_all(pkg, proof, log_max_rows):
    """Verifies `proof` under every convention set. Returns (accepting sets, {set: first failing check})."""
    accepted, reasons = [], {}
    for conv in all_sets():
        try:
            ok, why = pkg.verify_brainfuck(proof, log_max_rows, conventions=conv)
        except Exception as e:                      # a proof of the other channel's shape does not even parse
            ok, why = False, f"error: {e}"
        if ok:
            accepted.append(conv)
        else:
            reasons[conv] = why
    return accepted, reasons


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    p = sub.add_parser("prove")
    p.add_argument("--file"); p.add_argument("--code"); p.add_argument("--input-file"); p.add_argument("--ram-size", type=int, default=0)
    p.add_argument("--output"); p.add_argument("--log-max-rows", type=int, default=24); p.add_argument("--poseidon252", action="store_true")
    p.add_argument("--conventions"); p.add_argument("--all-sets", metavar="DIR")
    v = sub.add_parser("verify")
    v.add_argument("proof"); v.add_argument("--log-max-rows", type=int, default=24); v.add_argument("--poseidon252", action="store_true")
    v.add_argument("--conventions"); v.add_argument("--try-all", action="store_true")
    a = ap.parse_args()
    pkg = load_package()
    if a.cmd == "prove":
        code = open(a.file).read() if a.file else a.code
        if code is None:
            ap.error("prove needs --file or --code")
        inp = open(a.input_file, "rb").read() if a.input_file else (b"" if sys.stdin.isatty() else sys.stdin.buffer.read())
        sets = all_sets(channels=(0,)) if a.all_sets else [parse_conventions(a)]
        if a.all_sets:
            os.makedirs(a.all_sets, exist_ok=True)
        ctx = pkg.Context(0, max_log_domain=a.log_max_rows + 2)
        t0 = time.time()
        tr = pkg.Trace(ctx, code, inp, ram_size=a.ram_size)
        t_prep = time.time() - t0           # VM run + table build + upload, once, whatever the number of convention sets
        for conv in sets:
            ctx.set_conventions(*conv)
            t_start = time.time()
            proof, _ = tr.prove(a.log_max_rows)
            t_proof = time.time() - t_start
            print(f"Steps: {tr.n_steps}; trace preparation {1e3 * t_prep:.1f} ms; proof generation time: {t_proof:.3f}s; {len(proof)} bytes; {describe(conv)}", file=sys.stderr)
            if a.all_sets:
                path = os.path.join(a.all_sets, "proof_%d%d%d%d.json" % conv)
                open(path, "wb").write(proof)
                print(path)
            elif a.output:
                open(a.output, "wb").write(proof)
            else:
                sys.stdout.buffer.write(proof)
        tr.close(); ctx.close()
        return 0
    proof = open(a.proof, "rb").read()
    if a.try_all:
        accepted, reasons = try_all(pkg, proof, a.log_max_rows)
        for conv in accepted:
            print("Proof verified under --conventions %d,%d,%d,%d  (%s)" % (conv + (describe(conv),)))
        if not accepted:
            print("Verification failed under every convention set; first failing check per set:")
            for conv, why in reasons.items():
                print("  %d,%d,%d,%d  %s" % (conv + (why,)))
        return 0 if accepted else 1
    conv = parse_conventions(a)
    ok, why = pkg.verify_brainfuck(proof, a.log_max_rows, conventions=conv)
    print("Proof verified" if ok else f"Verification failed: {why}")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
