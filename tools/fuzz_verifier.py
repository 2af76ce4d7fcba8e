#!/usr/bin/env python3
"""Differential fuzzing of the two verifiers on the CPU (no GPU needed): the product's host verifier (bfhip_verify_brainfuck_conv) and the
oracle's (orc_verify) must give the same verdict on every mutated proof, and must both accept the unmutated one. Proofs come from the CPU
oracle on seeded random programs; mutations are structural edits of the proof JSON (change / bump / zero a number, swap, drop or duplicate
a list element, move a value between places). A proof both verifiers still ACCEPT after a semantic mutation is reported too.
Usage: python tools/fuzz_verifier.py [seconds=300] [first_seed=1000] [conventions name = stwo]"""
import copy, json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package, Oracle, CONVENTIONS
from bf_fuzz import random_program

P = (1 << 31) - 1


def paths(node, prefix=()):
    """Every (path, kind) in the JSON tree: kind 'num', 'list', 'str', 'dict'."""
    out = []
    if isinstance(node, dict):
        out.append((prefix, "dict"))
        for k, v in node.items():
            out += paths(v, prefix + (k,))
    elif isinstance(node, list):
        out.append((prefix, "list"))
        for i, v in enumerate(node):
            out += paths(v, prefix + (i,))
    elif isinstance(node, bool) or node is None:
        pass
    elif isinstance(node, int):
        out.append((prefix, "num"))
    elif isinstance(node, str):
        out.append((prefix, "str"))
    return out


def get(node, path):
    for k in path:
        node = node[k]
    return node


def set_(node, path, value):
    for k in path[:-1]:
        node = node[k]
    node[path[-1]] = value


def mutate(proof, rng):
    """Returns (mutated copy, description) or None if the draw was a no-op."""
    m = copy.deepcopy(proof)
    ps = paths(m)
    nums = [p for p, k in ps if k == "num"]
    lists = [p for p, k in ps if k == "list" and len(get(m, p)) > 0]
    strs = [p for p, k in ps if k == "str"]
    kind = rng.choice(["bump", "random", "zero", "pminus1", "swap", "drop", "dup", "move", "big", "type", "str"] if strs else ["bump", "random", "zero", "pminus1", "swap", "drop", "dup", "move", "big", "type"])
    if kind == "type":           # a value of another JSON type where a number, a list or an object is expected
        p = rng.choice([q for q, _ in ps if q])
        new = rng.choice([None, "1", [], {}, True, 1.5, [[]], {"a": 1}, 0])
        old = get(m, p)
        if type(old) is type(new) and old == new:
            return None
        set_(m, p, new)
        return m, f"type {'/'.join(map(str, p))}: {type(old).__name__} -> {json.dumps(new)}"
    if kind in ("bump", "random", "zero", "pminus1", "big"):
        p = rng.choice(nums); old = get(m, p)
        new = {"bump": old + 1, "random": rng.randrange(P), "zero": 0, "pminus1": P - 1, "big": rng.choice([P, P + 1, 1 << 32, (1 << 64) - 1, 1 << 64])}[kind]
        if new == old:
            return None
        set_(m, p, new)
        return m, f"{kind} {'/'.join(map(str, p))}: {old} -> {new}"
    if kind == "str":
        p = rng.choice(strs); old = get(m, p)
        i = rng.randrange(2, len(old)) if len(old) > 2 else 0
        c = rng.choice("0123456789abcdef")
        if not old or old[i] == c:
            return None
        set_(m, p, old[:i] + c + old[i + 1:])
        return m, f"str {'/'.join(map(str, p))}[{i}]"
    p = rng.choice(lists); lst = get(m, p)
    if kind == "swap":
        if len(lst) < 2:
            return None
        i, j = rng.sample(range(len(lst)), 2)
        if lst[i] == lst[j]:
            return None
        lst[i], lst[j] = lst[j], lst[i]
        return m, f"swap {'/'.join(map(str, p))}[{i}]<->[{j}]"
    if kind == "drop":
        i = rng.randrange(len(lst)); del lst[i]
        return m, f"drop {'/'.join(map(str, p))}[{i}]"
    if kind == "dup":
        i = rng.randrange(len(lst)); lst.insert(i, copy.deepcopy(lst[i]))
        return m, f"dup {'/'.join(map(str, p))}[{i}]"
    if kind == "move":
        q = rng.choice(lists)
        if q == p or not get(m, q):
            return None
        src = get(m, q); i = rng.randrange(len(src)); j = rng.randrange(len(lst) + 1)
        if type(src[i]) is not type(lst[0]):
            return None
        lst.insert(j, src.pop(i))
        return m, f"move {'/'.join(map(str, q))}[{i}] -> {'/'.join(map(str, p))}[{j}]"
    return None


def text_mutations(js, rng):
    """Edits of the serialised text that a canonical-JSON reader must refuse (or that change nothing): (bytes, description)."""
    import re
    out = []
    nums = list(re.finditer(rb"(?<![0-9.eE\"x])\d+(?![0-9])", js))
    keys = list(re.finditer(rb'"(log_size|commitment|proof_of_work|coeffs|hash_witness|claimed_sum)":', js))
    if keys:                                  # a second occurrence of a key in its object, in front of the real one
        m = rng.choice(keys)
        out.append((js[:m.start()] + m.group() + rng.choice([b"0", b"[]", b"null"]) + b"," + js[m.start():], f"duplicate key {m.group(1).decode()} at byte {m.start()}"))
    for kind in ("leading_zero", "fraction", "exponent", "negative", "plus2p64", "trailing", "space_inside", "null"):
        m = rng.choice(nums)
        t = m.group()
        if kind == "leading_zero": new = b"0" + t
        elif kind == "fraction": new = t + b".0"
        elif kind == "exponent": new = t + b"e0"
        elif kind == "negative": new = b"-" + t
        elif kind == "plus2p64": new = str(int(t) + (1 << 64)).encode()
        elif kind == "space_inside": new = t[:1] + b" " + t[1:] if len(t) > 1 else None
        elif kind == "null": new = b"nuII"
        else: new = None
        if kind == "trailing":
            out.append((js + rng.choice([b"0", b"}", b" x", b",{}"]), "trailing bytes"))
        elif new is not None:
            out.append((js[:m.start()] + new + js[m.end():], f"{kind} at byte {m.start()}"))
    return out


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    cname = sys.argv[3] if len(sys.argv) > 3 else "stwo"
    conv = CONVENTIONS[cname]
    pkg = load_package(); orc = Oracle(); orc.set_conventions(*conv)
    rng = random.Random(seed)
    t_end = time.time() + budget
    summary = {"seconds": budget, "first_seed": seed, "conventions": cname, "proofs": 0, "mutations": 0, "both_reject": 0, "both_accept": [], "disagree": [], "valid_rejected": []}
    while time.time() < t_end:
        code, inp, _ = random_program(seed, 400, min_steps=20)
        lmr = max(max(orc.log_sizes(code, inp)[0]), 8)
        js, _, _ = orc.prove(code, inp, log_max_rows=lmr)
        if not (pkg.verify_brainfuck(js, lmr, conv)[0] and orc.verify(js, lmr)[0]):
            summary["valid_rejected"].append(seed)
        proof = json.loads(js)
        summary["proofs"] += 1
        for _ in range(150):
            mu = mutate(proof, rng)
            if mu is None:
                continue
            m, what = mu
            mjs = json.dumps(m, separators=(",", ":")).encode()
            try:
                a = pkg.verify_brainfuck(mjs, lmr, conv)[0]
            except Exception as e:
                a = f"error: {e!r}"
            b = orc.verify(mjs, lmr)[0]
            summary["mutations"] += 1
            if a is False and b is False:
                summary["both_reject"] += 1
            elif a is True and b is True:
                summary["both_accept"].append({"seed": seed, "mutation": what})
            else:
                summary["disagree"].append({"seed": seed, "mutation": what, "product": a, "oracle": b})
        for mjs, what in text_mutations(js, rng):
            try:
                a = pkg.verify_brainfuck(mjs, lmr, conv)[0]
            except Exception as e:
                a = f"error: {e!r}"
            b = orc.verify(mjs, lmr)[0]
            summary["mutations"] += 1
            if a is False and b is False:
                summary["both_reject"] += 1
            elif a is True and b is True:
                summary["both_accept"].append({"seed": seed, "mutation": what})
            else:
                summary["disagree"].append({"seed": seed, "mutation": what, "product": a, "oracle": b})
        seed += 1
    summary["ok"] = not summary["disagree"] and not summary["valid_rejected"]
    summary["both_accept"] = summary["both_accept"][:40]
    print(json.dumps(summary, indent=1))
    return 0 if summary["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
