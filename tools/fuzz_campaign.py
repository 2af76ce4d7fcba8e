#!/usr/bin/env python3
"""Time-bounded randomized parity campaign on one GPU box (test infrastructure; the committed suite runs a fixed subset of this):
seeded random Brainfuck programs of four size classes, every convention set (Poseidon252 on the smaller classes — its CPU oracle is slow),
proved (a) by the CPU oracle, (b) by one context, (c) by a local shard group of 2/4/8 contexts. All proofs of one case must be the
same bytes and both verifiers must accept them. Prints one JSON summary; exit code 1 on any mismatch.
Usage: python tools/fuzz_campaign.py [seconds=600] [first_seed=10000] [fresh|persistent|pool] [xl case every n-th = 11]
persistent: ONE set of 8 long-lived contexts instead of fresh ones per case — between cases they join and leave groups of changing size,
switch conventions, toggle the preprocessed-tree cache and prove with LOG_MAX_ROWS above the trace's need (state carried across proofs:
arena, caches, staging ring, group membership)."""
import json, os, random, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package, Oracle, CONVENTIONS
from bf_fuzz import random_program, _simulate

CLASSES = [("small", 400), ("medium", 6000), ("large", 60000), ("xl", 400000)]


def program(seed, cls, bound):
    """small/medium: tests/bf_fuzz.py as the suite uses it; large/xl: a random body inside a counted outer loop, accepted when it halts within
    the bound, stays right of cell 0 and has enough input."""
    if cls in ("small", "medium"):
        return random_program(seed, bound, min_steps=bound // 20)
    rng = random.Random(seed)
    while True:
        body, _, body_steps = random_program(rng.randrange(1 << 30), 300, min_steps=10)
        n = max(2, min(250, bound // (2 * body_steps + 4)))
        code = "+" * n + "[>" + body + "<-]" + rng.choice(["", ".", ">+."])
        if cls == "xl":
            code = "+" * rng.randint(8, 14) + "[>" + code + "<-]"
        inp = bytes(rng.randrange(256) for _ in range(4096))
        steps = _simulate(code, inp, bound)
        if steps is not None and steps >= bound // 8:
            return code, inp, steps


def prove_sharded(pkg, code, inp, lmr, count, conv, policy=0):
    group = pkg.LocalGroup(count)
    ctxs = [pkg.Context(0, max_log_domain=lmr + 2) for _ in range(count)]
    proofs, errors = [None] * count, []

    def run(rank):
        try:
            ctxs[rank].set_conventions(*conv)
            ctxs[rank].join_local_group(group, rank)
            ctxs[rank].set_shard_policy(policy)
            proofs[rank] = pkg.prove_brainfuck(code, inp, ctx=ctxs[rank], log_max_rows=lmr)
        except Exception as e:
            errors.append(repr(e))

    th = [threading.Thread(target=run, args=(r,)) for r in range(count)]
    [t.start() for t in th]; [t.join() for t in th]
    for c in ctxs:
        c.leave_group(); c.close()
    group.close()
    if errors:
        raise RuntimeError("; ".join(errors))
    return proofs


def persistent(budget, seed):
    pkg = load_package()
    orc = Oracle()
    orc.L.orc_set_threads(min(64, len(os.sched_getaffinity(0))))
    convs = list(CONVENTIONS.items())
    MAXLOG = 21
    ctxs = [pkg.Context(0, max_log_domain=MAXLOG + 2) for _ in range(8)]
    rng = random.Random(seed)
    t_end = time.time() + budget
    summary = {"mode": "persistent", "seconds": budget, "first_seed": seed, "cases": 0, "by_shard_count": {}, "by_conventions": {}, "reuse_preprocessed_on": 0, "lmr_slack": {}, "failures": []}
    while time.time() < t_end:
        cls, bound = CLASSES[rng.randrange(3)]
        cname, conv = convs[rng.randrange(len(convs))]
        if cname == "poseidon" and cls == "large":
            cname, conv = convs[0]
        count = rng.choice((1, 1, 2, 4, 8))
        reuse = rng.random() < 0.4
        slack = rng.choice((0, 0, 1, 2))
        gpu_tables = rng.random() < 0.7
        via_trace = rng.random() < 0.5
        # bfhip_ctx_set_overlap: single contexts take any of the intra-proof modes, a group the same mask on every rank (bit 2 = exchanges on the
        # partner stream changes the number of collectives)
        overlap = rng.choice((0, 0, 1, 2, 3)) if count == 1 else rng.choice((0, 0, 4, 4, 7))
        # bfhip_ctx_set_shard_policy (r06), the same on every rank: 0 exchange columns -> rows, 1 replicate the transforms; it persists on a context, so a later
        # case re-uses contexts whose kept preprocessed tree was built under the other policy (the cache is keyed on it)
        policy = rng.choice((0, 1))
        case = {"seed": seed, "class": cls, "conventions": cname, "shard_count": count, "reuse_preprocessed": reuse, "lmr_slack": slack, "gpu_tables": gpu_tables, "via_trace": via_trace,
                "overlap": overlap, "shard_policy": policy}
        try:
            code, inp, steps = program(seed, cls, bound)
            orc.set_conventions(*conv)
            lmr = max(max(orc.log_sizes(code, inp)[0]), 8) + slack
            if lmr > MAXLOG:
                seed += 1
                continue
            want, _, _ = orc.prove(code, inp, log_max_rows=lmr)
            members = ctxs[:count]
            group = pkg.LocalGroup(count) if count > 1 else None
            proofs, errors = [None] * count, []

            def run(r):
                try:
                    c = members[r]
                    c.set_conventions(*conv)
                    pkg.lib().bfhip_ctx_reuse_preprocessed(c._h, 1 if reuse else 0)
                    if group is not None:
                        c.join_local_group(group, r)
                    c.set_table_builder(gpu_tables)
                    c.set_overlap(overlap)
                    c.set_shard_policy(policy)
                    a = pkg.prove_brainfuck(code, inp, ctx=c, log_max_rows=lmr)
                    if via_trace:                                                    # second proof: the cached tree (if on), a warm arena,
                        tr = pkg.Trace(c, code, inp)                                 # and the resident-trace entry instead of the one-call entry
                        try:
                            b = tr.prove(lmr)[0]
                        finally:
                            tr.close()
                    else:
                        b = pkg.prove_brainfuck(code, inp, ctx=c, log_max_rows=lmr)
                    proofs[r] = (a, b)
                except Exception as e:
                    errors.append(repr(e))

            th = [threading.Thread(target=run, args=(r,)) for r in range(count)]
            [t.start() for t in th]; [t.join() for t in th]
            for c in members:
                if group is not None:
                    c.leave_group()
            if group is not None:
                group.close()
            problems = list(errors)
            for r, pr in enumerate(proofs):
                if pr is not None and (pr[0] != want or pr[1] != want):
                    problems.append(f"rank {r} of {count}: proof differs from the oracle's (first {pr[0] == want}, second {pr[1] == want})")
            if problems:
                case["problems"] = problems; case["code"] = code
                summary["failures"].append(case)
                if errors:                    # a failed group leaves contexts in an unknown state: start over with fresh ones
                    for c in ctxs:
                        c.close()
                    ctxs = [pkg.Context(0, max_log_domain=MAXLOG + 2) for _ in range(8)]
        except Exception as e:
            case["problems"] = [repr(e)]
            summary["failures"].append(case)
        summary["cases"] += 1
        summary["by_shard_count"][str(count)] = summary["by_shard_count"].get(str(count), 0) + 1
        summary["by_conventions"][cname] = summary["by_conventions"].get(cname, 0) + 1
        summary["lmr_slack"][str(slack)] = summary["lmr_slack"].get(str(slack), 0) + 1
        summary["reuse_preprocessed_on"] += int(reuse)
        summary.setdefault("by_overlap_mask", {})[str(overlap)] = summary.setdefault("by_overlap_mask", {}).get(str(overlap), 0) + 1
        seed += 1
    for c in ctxs:
        c.close()
    summary["last_seed"] = seed - 1
    summary["ok"] = not summary["failures"]
    print(json.dumps(summary, indent=1))
    return 0 if summary["ok"] else 1


def pool_campaign(budget, seed):
    """r06: random BATCHES through long-lived pools (bfhip_pool_*): pools of 1..4 sub-contexts that live for the whole campaign, every batch 2..8 random programs of the
    small / medium classes (one LOG_MAX_ROWS per batch = the largest need + slack), handed over as resident traces or as program text, under a convention set and a
    preprocessed mode that change from batch to batch on the SAME pool (shared-tree invalidation, kept trees across batches, workers re-using their sub-contexts).
    Every proof of every batch must be the CPU oracle's bytes."""
    pkg = load_package()
    orc = Oracle()
    orc.L.orc_set_threads(min(64, len(os.sched_getaffinity(0))))
    convs = [c for c in CONVENTIONS.items() if c[0] != "poseidon"] + [("poseidon", CONVENTIONS["poseidon"])]
    MAXLOG = 19
    pools = {k: pkg.Pool(0, n_in_flight=k, max_log_domain=MAXLOG + 2) for k in (1, 2, 3, 4)}
    summary = {"mode": "pool", "seconds": budget, "first_seed": seed, "batches": 0, "proofs": 0, "by_pool_size": {}, "by_preprocessed_mode": {}, "by_conventions": {}, "via_program_text": 0, "failures": []}
    t_end = time.time() + budget
    rng = random.Random(seed)
    while time.time() < t_end:
        k = rng.choice((1, 2, 2, 3, 3, 4))
        mode = rng.choice((0, 1, 1, 2))
        cname, conv = convs[rng.randrange(len(convs) - 1)] if rng.random() < 0.9 else convs[-1]
        via_text = rng.random() < 0.4
        n = rng.randint(2, 8)
        case = {"seed": seed, "pool": k, "preprocessed": mode, "conventions": cname, "n": n, "via_program_text": via_text}
        try:
            progs = []
            for _ in range(n):
                cls, bound = CLASSES[0] if (cname == "poseidon" or rng.random() < 0.6) else CLASSES[1]
                code, inp, _ = program(seed, cls, bound)
                progs.append((code, inp)); seed += 1
            orc.set_conventions(*conv)
            lmr = max(max(max(orc.log_sizes(c, i)[0]) for c, i in progs), 8) + rng.choice((0, 0, 1))
            if lmr > MAXLOG:
                continue
            want = [orc.prove(c, i, log_max_rows=lmr)[0] for c, i in progs]
            pool = pools[k]
            pool.set_conventions(*conv)
            pool.set_preprocessed(mode)
            if via_text:
                got, _ = pool.prove_batch_brainfuck(progs, log_max_rows=lmr)
            else:
                traces = [pkg.Trace(pool.ctx(rng.randrange(k)), c, i) for c, i in progs]
                try:
                    got, _ = pool.prove_batch(traces, log_max_rows=lmr)
                finally:
                    for t in traces:
                        t.close()
            bad = [i for i in range(n) if got[i] != want[i]]
            if bad:
                case["problems"] = [f"proofs {bad} of the batch differ from the oracle's"]; case["codes"] = [progs[i][0] for i in bad]
                summary["failures"].append(case)
            summary["proofs"] += n
        except Exception as e:
            case["problems"] = [repr(e)]
            summary["failures"].append(case)
        summary["batches"] += 1
        summary["via_program_text"] += int(via_text)
        for key, v in (("by_pool_size", str(k)), ("by_preprocessed_mode", str(mode)), ("by_conventions", cname)):
            summary[key][v] = summary[key].get(v, 0) + 1
    for p in pools.values():
        p.close()
    summary["last_seed"] = seed - 1
    summary["ok"] = not summary["failures"]
    print(json.dumps(summary, indent=1))
    return 0 if summary["ok"] else 1


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    if len(sys.argv) > 3 and sys.argv[3] == "persistent":
        return persistent(budget, seed)
    if len(sys.argv) > 3 and sys.argv[3] == "pool":
        return pool_campaign(budget, seed)
    pkg = load_package()
    orc = Oracle()
    orc.L.orc_set_threads(min(64, len(os.sched_getaffinity(0))))
    convs = list(CONVENTIONS.items())
    xl_every = int(sys.argv[4]) if len(sys.argv) > 4 else 11
    t_end = time.time() + budget
    summary = {"seconds": budget, "first_seed": seed, "cases": 0, "by_class": {}, "by_conventions": {}, "by_shard_count": {}, "failures": []}
    k = 0
    while time.time() < t_end:
        cls, bound = CLASSES[(k // 3) % 3 if k % xl_every else 3]     # mostly small..large, an xl case every 11th (argv[4]: every n-th)
        cname, conv = convs[k % len(convs)]
        if cname == "poseidon" and cls in ("large", "xl"):
            cname, conv = convs[0]
        count = (1, 2, 4, 8)[(k // 2) % 4]
        case = {"seed": seed, "class": cls, "conventions": cname, "shard_count": count}
        try:
            code, inp, steps = program(seed, cls, bound)
            case["vm_steps"] = steps
            orc.set_conventions(*conv)
            lmr = max(max(orc.log_sizes(code, inp)[0]), 8)
            if lmr >= 25:                   # the group members share ONE GPU here: eight full-size contexts do not fit its HBM
                count = min(count, 4 if lmr == 25 else 2); case["shard_count"] = count
            want, _, _ = orc.prove(code, inp, log_max_rows=lmr)
            c1 = pkg.Context(0, max_log_domain=lmr + 2)
            c1.set_conventions(*conv)
            got = pkg.prove_brainfuck(code, inp, ctx=c1, log_max_rows=lmr)
            c1.close()
            problems = []
            if got != want:
                problems.append("single-context proof differs from the oracle's")
            if count > 1:
                case["shard_policy"] = seed & 1
                for r, p in enumerate(prove_sharded(pkg, code, inp, lmr, count, conv, policy=seed & 1)):
                    if p != want:
                        problems.append(f"rank {r} of {count} differs from the oracle's proof")
            if pkg.verify_brainfuck(got, lmr, conv) != (True, ""):
                problems.append("own verifier rejects")
            if not orc.verify(got, lmr)[0]:
                problems.append("oracle verifier rejects")
            if problems:
                case["problems"] = problems; case["code"] = code; case["input_hex"] = inp.hex()[:256]
                summary["failures"].append(case)
        except Exception as e:
            case["problems"] = [repr(e)]
            summary["failures"].append(case)
        summary["cases"] += 1
        for key, v in (("by_class", cls), ("by_conventions", cname), ("by_shard_count", str(count))):
            summary[key][v] = summary[key].get(v, 0) + 1
        seed += 1; k += 1
    summary["last_seed"] = seed - 1
    summary["ok"] = not summary["failures"]
    print(json.dumps(summary, indent=1))
    return 0 if summary["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
