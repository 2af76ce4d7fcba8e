"""-m gpu: Poseidon252 Merkle variant (BASELINE config 5 hasher). The Hades permutation is pinned by the public known-answer vector;
layers are compared with the pure-Python oracle (oracle/poseidon252.py) on small mixed-degree trees."""
import importlib.util
import os

import numpy as np
import pytest

from conftest import ROOT, splitmix_column

pytestmark = pytest.mark.gpu

spec = importlib.util.spec_from_file_location("poseidon252_oracle", os.path.join(ROOT, "oracle", "poseidon252.py"))
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def test_hades_known_answer(ctx):
    ref.self_test()
    assert ctx.hades_permutation([0, 0, 0]) == ref.HADES_ZERO_KAT


def test_hades_random_states(ctx):
    rng = np.random.default_rng(5)
    for _ in range(5):
        st = [int.from_bytes(rng.bytes(32), "little") % ref.P for _ in range(3)]
        assert ctx.hades_permutation(st) == ref.hades(st)
    assert ctx.hades_permutation([ref.P - 1, 1, ref.P - 2]) == ref.hades([ref.P - 1, 1, ref.P - 2])


def test_hades_limb_boundary_states(ctx):
    """States that put the extreme values into the 29-bit limbs of the kernel's field code (poseidon_dev.h): all-ones limbs, single top bits,
    values just below p and just above the 2^232 / 2^251 boundaries, small values whose upper limbs are all zero, and 200 random states."""
    P = ref.P
    edge = [0, 1, 2, P - 1, P - 2, (1 << 29) - 1, 1 << 29, (1 << 232) - 1, 1 << 232, (1 << 251) - 1, 1 << 251, (1 << 251) + 1, P >> 1,
            sum(((1 << 29) - 1) << (29 * i) for i in range(8)), 17 << 192, (17 << 192) - 1, ((1 << 251) + (17 << 192))]
    rng = np.random.default_rng(29)
    states = [[a, b, c] for a in edge[:6] for b in edge[6:12] for c in edge[12:]][:120]
    states += [[edge[i % len(edge)], edge[(i * 7 + 3) % len(edge)], edge[(i * 5 + 1) % len(edge)]] for i in range(40)]
    states += [[int.from_bytes(rng.bytes(32), "little") % P for _ in range(3)] for _ in range(200)]
    for st in states:
        st = [v % P for v in st]
        assert ctx.hades_permutation(st) == ref.hades(list(st)), st


def _to_int(words):
    return sum(int(w) << (32 * i) for i, w in enumerate(words))


@pytest.mark.parametrize("shape", [{5: 3}, {4: 9, 2: 1}, {6: 8, 5: 16, 3: 2, 0: 1}, {3: 17}])
def test_poseidon_merkle_layers_match_python_oracle(ctx, shape):
    cols = {log: [splitmix_column(1000 + 37 * log + k, 1 << log) for k in range(n)] for log, n in shape.items()}
    want = ref.merkle_layers({log: [c.tolist() for c in cs] for log, cs in cols.items()})
    dev = {log: [ctx.upload(c) for c in cs] for log, cs in cols.items()}
    prev = 0
    for log in range(max(shape), -1, -1):
        out = ctx.malloc(32 << log)
        ctx.merkle_commit_layer_poseidon252(log, prev, dev.get(log, []), out)
        got = ctx.download(out, 8 << log).reshape(-1, 8)
        assert [_to_int(r) for r in got] == want[log], f"layer {log}"
        prev = out


def test_poseidon_leaf_with_replicated_column(ctx):
    log = 5
    full = splitmix_column(7, 1 << log); rows = splitmix_column(8, 1 << (log - 4))
    pf, pr = ctx.upload(full), ctx.upload(rows)
    out = ctx.malloc(32 << log)
    ctx.merkle_commit_layer_poseidon252(log, 0, [pf, pr], out, col_shifts=[0, 4])
    got = ctx.download(out, 8 << log).reshape(-1, 8)
    for i in range(1 << log):
        assert _to_int(got[i]) == ref.hash_node(None, [int(full[i]), int(rows[i >> 4])])
