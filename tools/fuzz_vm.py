#!/usr/bin/env python3
"""Differential fuzzing of the product's host side (compiler, VM, 13 table builders behind the C ABI: bfhip_host_compile / _run / _table)
against the oracle's (orc_compile / orc_run / orc_table) on the CPU. Programs are UNCONSTRAINED random strings over the eight instructions
(plus comment characters), so most are invalid or fail at run time: both sides must then fail alike (unmatched brackets, pointer below cell 0
or past the RAM, input exhausted), and on the valid ones agree on compiled words, output bytes, the register trace and every table.
Usage: python tools/fuzz_vm.py [seconds=120] [first_seed=1]"""
import ctypes, json, os, random, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package, Oracle

MAX_ROWS = 1 << 16


def gen(rng):
    style = rng.random()
    n = rng.randint(0, 60)
    if style < 0.5:          # anything goes
        alphabet = "+-<>[].,+-<>[].,  x\n"
        code = "".join(rng.choice(alphabet) for _ in range(n))
    else:                    # bracket-balanced, more likely to run for a while
        out, depth = [], 0
        for _ in range(n):
            c = rng.choice("++++---<>>>..,,[]")
            if c == "[":
                depth += 1
            elif c == "]":
                if depth == 0:
                    continue
                depth -= 1
            out.append(c)
        code = "".join(out) + "]" * depth
    inp = bytes(rng.randrange(256) for _ in range(rng.choice([0, 0, 1, 2, 8, 64])))
    return code, inp


def halts_quickly(code, inp, bound=20000):
    """Step-bounded pre-run (the VMs themselves have no step limit, like the reference's): False = skip this program. Errors count as halting."""
    stack, match = [], {}
    for i, ch in enumerate(code):
        if ch == "[":
            stack.append(i)
        elif ch == "]":
            if not stack:
                return True                   # compile error on both sides, no run
            j = stack.pop(); match[i] = j; match[j] = i
    if stack:
        return None                   # unmatched '[': the compilers accept it (as the reference's does) and leave a zero jump target — compile only
    ram, mp, ip, steps, k = {}, 0, 0, 0, 0
    P = (1 << 31) - 1
    while ip < len(code):
        ch = code[ip]
        if ch in "+-<>[].,":
            steps += 1
            if steps > bound:
                return False
        v = ram.get(mp, 0)
        if ch == "+": ram[mp] = (v + 1) % P
        elif ch == "-": ram[mp] = (v - 1) % P
        elif ch == ">": mp += 1
        elif ch == "<":
            mp -= 1
            if mp < 0:
                return True
        elif ch == ",":
            if k >= len(inp):
                return True                   # whatever the VMs do here, they do it in bounded time: the read either fails or yields a value once
            ram[mp] = inp[k]; k += 1
        elif ch == "[" and v == 0: ip = match[ip]
        elif ch == "]" and v != 0: ip = match[ip]
        ip += 1
    return True


class Side:
    """Uniform view of one implementation: compile / run return (ok, payload)."""
    def __init__(self, L, prefix):
        self.compile_f = getattr(L, prefix + "compile"); self.run_f = getattr(L, prefix + "run")

    def compile(self, code):
        out = np.zeros(2 * len(code) + 8, dtype=np.uint32); n = ctypes.c_size_t()
        rc = self.compile_f(code.encode(), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(out.size), ctypes.byref(n))
        return (True, out[: n.value].tolist()) if rc == 0 else (False, None)

    def run(self, code, inp):
        n_out, n_rows = ctypes.c_size_t(), ctypes.c_size_t()
        rc = self.run_f(code.encode(), inp, ctypes.c_size_t(len(inp)), None, ctypes.c_size_t(0), ctypes.byref(n_out), None, ctypes.c_size_t(0), ctypes.byref(n_rows))
        if rc != 0:
            return False, None
        if n_rows.value > MAX_ROWS:
            return True, ("too long", n_rows.value)
        out = (ctypes.c_ubyte * max(1, n_out.value))(); tr = np.zeros((n_rows.value, 7), dtype=np.uint32)
        rc = self.run_f(code.encode(), inp, ctypes.c_size_t(len(inp)), out, ctypes.c_size_t(n_out.value), ctypes.byref(n_out), tr.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n_rows.value), ctypes.byref(n_rows))
        if rc != 0:
            return False, None
        return True, (bytes(out[: n_out.value]), tr)


def table(f, trace, words, comp):
    cw = np.ascontiguousarray(words, dtype=np.uint32); nr, nc = ctypes.c_size_t(), ctypes.c_size_t()
    args = (trace.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(trace.shape[0]), cw.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(cw.size), comp)
    if f(*args, None, ctypes.c_size_t(0), ctypes.byref(nr), ctypes.byref(nc)) != 0:
        return None
    out = np.zeros((nr.value, nc.value), dtype=np.uint32)
    if f(*args, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(out.size), ctypes.byref(nr), ctypes.byref(nc)) != 0:
        return None
    return out


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    pkg = load_package(); orc = Oracle()
    prod, orac = Side(pkg.lib(), "bfhip_host_"), Side(orc.L, "orc_")
    rng = random.Random(seed)
    t_end = time.time() + budget
    s = {"seconds": budget, "first_seed": seed, "programs": 0, "compile_errors": 0, "run_errors": 0, "ran": 0, "tables_compared": 0, "problems": []}
    while time.time() < t_end:
        code, inp = gen(rng)
        h = halts_quickly(code, inp)
        if h is False:
            continue
        s["programs"] += 1
        a, b = prod.compile(code), orac.compile(code)
        if a != b:
            s["problems"].append({"code": code, "what": "compile", "product": a, "oracle": b}); continue
        if not a[0]:
            s["compile_errors"] += 1; continue
        if h is None:
            s["compile_only"] = s.get("compile_only", 0) + 1; continue
        ra, rb = prod.run(code, inp), orac.run(code, inp)
        if ra[0] != rb[0]:
            s["problems"].append({"code": code, "input": inp.hex(), "what": "run status", "product": ra[0], "oracle": rb[0]}); continue
        if not ra[0]:
            s["run_errors"] += 1; continue
        if isinstance(ra[1][0], str) or isinstance(rb[1][0], str):
            if ra[1] != rb[1]:
                s["problems"].append({"code": code, "what": "row count", "product": str(ra[1]), "oracle": str(rb[1])})
            continue
        s["ran"] += 1
        if ra[1][0] != rb[1][0] or ra[1][1].shape != rb[1][1].shape or not np.array_equal(ra[1][1], rb[1][1]):
            s["problems"].append({"code": code, "input": inp.hex(), "what": "output or trace"}); continue
        for comp in range(13):
            ta = table(pkg.lib().bfhip_host_table, ra[1][1], a[1], comp)
            tb = table(orc.L.orc_table_from_registers, rb[1][1], b[1], comp) if hasattr(orc.L, "orc_table_from_registers") else None
            if tb is None:
                continue
            s["tables_compared"] += 1
            if ta is None or ta.shape != tb.shape or not np.array_equal(ta, tb):
                s["problems"].append({"code": code, "input": inp.hex(), "what": f"table {comp}"}); break
    s["ok"] = not s["problems"]
    s["problems"] = s["problems"][:20]
    print(json.dumps(s, indent=1))
    return 0 if s["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
