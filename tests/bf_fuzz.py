"""Seeded generator of small terminating Brainfuck programs for randomized parity tests (test infrastructure only).

Programs use all eight instructions, nested loops included. A tiny interpreter with the VM's field semantics (cells are M31 values:
`-` on 0 wraps to P - 1, machine.rs:189-192) bounds the step count so that every generated program halts quickly and never reads
past its input or moves left of cell 0."""
import random

P = (1 << 31) - 1


def _simulate(code, inp, max_steps):
    """Returns the number of executed instructions, or None if the program is not acceptable."""
    stack, match = [], {}
    for i, ch in enumerate(code):
        if ch == "[":
            stack.append(i)
        elif ch == "]":
            if not stack:
                return None
            j = stack.pop(); match[i] = j; match[j] = i
    if stack:
        return None
    ram, mp, ip, steps, in_pos = {}, 0, 0, 0, 0
    while ip < len(code):
        ch = code[ip]
        steps += 1
        if steps > max_steps:
            return None
        v = ram.get(mp, 0)
        if ch == "+": ram[mp] = (v + 1) % P
        elif ch == "-": ram[mp] = (v - 1) % P
        elif ch == ">": mp += 1
        elif ch == "<":
            mp -= 1
            if mp < 0:
                return None
        elif ch == ",":
            if in_pos >= len(inp):
                return None
            ram[mp] = inp[in_pos]; in_pos += 1
        elif ch == "[" and v == 0: ip = match[ip]
        elif ch == "]" and v != 0: ip = match[ip]
        ip += 1
    return steps


def _body(rng, depth, scale=1):
    out = []
    for _ in range(rng.randint(1, 6 * scale)):
        r = rng.random()
        if r < 0.18 and depth < 2:
            k = rng.randint(1, 2)
            # counted loop: the counter cell is decremented once per iteration, the body works k cells to the right
            out.append("[" + ">" * k + _body(rng, depth + 1) + "<" * k + "-]")
        elif r < 0.26:
            out.append("[-]")                      # clear (skipped at once on a zero cell: jump-if-zero taken)
        else:
            out.append(rng.choice("++++--><..,") if depth else rng.choice("+++-><.,"))
    return "".join(out)


def random_program(seed, max_steps=400, min_steps=1):
    """Deterministic in (seed, max_steps, min_steps): (code, input bytes, executed steps), min_steps <= steps <= max_steps."""
    rng = random.Random(seed)
    scale = 1 if max_steps <= 1000 else 3
    while True:
        code = "+" * rng.randint(0, 4 * scale) + _body(rng, 0, scale) + rng.choice(["", ".", "+.", ">+"])
        inp = bytes(rng.randrange(256) for _ in range(code.count(",") * 8))
        steps = _simulate(code, inp, max_steps)
        if steps is not None and steps >= min_steps:
            return code, inp[: max(1, len(inp))], steps
