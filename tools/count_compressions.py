"""Counts the Blake2s compressions k_merkle_layer performs for one proof (same layer/shift rules as prover.hip::merkle_commit),
given the 13 component log sizes and LOG_MAX_ROWS. Used for the VALU roofline of the Merkle kernel in bench.py / DESIGN.md."""
import sys

MAIN = [8, 8, 4, 9, 13, 13, 11, 11, 11, 11, 11, 11, 7]
LOGUP = [1, 1, 1, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1]


def tree(cols, fused_top_limit=10):
    """cols: list of (log_size, shift). Returns (compressions in k_merkle_layer, compressions in k_merkle_top)."""
    cols = sorted(cols, key=lambda c: -c[0])
    max_log, min_log = cols[0][0], cols[-1][0]
    shifts = {}
    total = 0
    for log in range(max_log, -1, -1):
        layer = [c for c in cols if c[0] == log]
        sh = 32 if log == max_log else max(shifts[log + 1] - 1, 0)
        for c in layer:
            sh = min(sh, c[1])
        shifts[log] = min(0 if sh == 32 else sh, log)
    fused = min(min_log, fused_top_limit)
    while fused > 0 and shifts[fused] != 0:
        fused -= 1
    top = 0
    for log in range(max_log, -1, -1):
        n_cols = sum(1 for c in cols if c[0] == log)
        msg = (64 if log < max_log else 0) + 4 * n_cols
        blocks = max(1, -(-msg // 64))
        nodes = (1 << log) >> shifts[log]
        if log >= fused:
            total += nodes * blocks
        else:
            top += nodes * blocks
    return total, top


def proof(log_sizes, lmr):
    trees = []
    trees.append([(l + 1, 0) for l in range(lmr, 3, -1)])                                   # preprocessed IsFirst(lmr..4), LDE
    trees.append([(l + 1, 4) for l, m in zip(log_sizes, MAIN) for _ in range(m)])           # main trace: all replicated
    t2 = []
    for l, n in zip(log_sizes, LOGUP):
        t2 += [(l + 1, 4)] * (4 * (n - 1)) + [(l + 1, 0)] * 4
    trees.append(t2)
    comp_log = max(log_sizes) + 1
    trees.append([(comp_log + 1, 0)] * 4)                                                    # composition
    sizes = sorted({c[0] for t in trees for c in t}, reverse=True)
    trees.append([(s, 0) for s in sizes for _ in range(4)])                                  # FRI first layer: 4 coords per quotient
    line = sizes[0] - 1
    while line > 1:
        trees.append([(line, 0)] * 4)                                                        # FRI inner layers
        line -= 1
    a = b = 0
    for t in trees:
        x, y = tree(t)
        a += x; b += y
    return a, b


if __name__ == "__main__":
    ls = [int(x) for x in sys.argv[1:14]] if len(sys.argv) >= 14 else [24, 22, 11, 22, 19, 11, 4, 20, 19, 4, 20, 20, 4]
    lmr = int(sys.argv[14]) if len(sys.argv) > 14 else 24
    a, b = proof(ls, lmr)
    print("k_merkle_layer compressions:", a, " k_merkle_top:", b)
