"""ORACLE — TEST INFRASTRUCTURE ONLY. Pure-Python restatement (small cases only) of the Poseidon252 Merkle hasher of BASELINE.json config 5.

Not used by the reference (no `Poseidon` token in crates/); it is an upstream stwo capability
(`MerkleOps<Poseidon252MerkleHasher>`, stwo@31e8dbc core/vcs/poseidon252_merkle.rs) over starknet-crypto 0.6.2's `poseidon_hash_many`
(Cargo.lock:821-864).

Pinning:
  * the Hades permutation (round constants = sha256("Hades" + str(i)) mod p, MDS [[3,1,1],[1,-1,1],[1,1,-2]], 4 + 83 + 4 rounds, x^3)
    is PINNED by the public known-answer vector hades([0,0,0]) checked in `self_test()`;
  * `hash_node` (children first, then blocks of 8 M31 packed as w = w * 2^31 + v, zero padded) is the recalled stwo layout —
    PARITY UNPINNED (no call site, no vector in the reference).
"""
from hashlib import sha256

P = 2**251 + 17 * 2**192 + 1
M, R_F, R_P = 3, 8, 83
ARK = [[int(sha256(f"Hades{M * i + j}".encode()).hexdigest(), 16) % P for j in range(M)] for i in range(R_F + R_P)]

HADES_ZERO_KAT = [
    3446325744004048536138401612021367625846492093718951375866996507163446763827,
    1590252087433376791875644726012779423683501236913937337746052470473806035332,
    867921192302518434283879514999422690776342565400001269945778456016268852423,
]


def hades(state):
    s = list(state)
    for i in range(R_F + R_P):
        s = [(x + k) % P for x, k in zip(s, ARK[i])]
        full = i < R_F // 2 or i >= R_F // 2 + R_P
        if full:
            s = [pow(x, 3, P) for x in s]
        else:
            s[2] = pow(s[2], 3, P)
        t = (s[0] + s[1] + s[2]) % P
        s = [(t + 2 * s[0]) % P, (t - 2 * s[1]) % P, (t - 3 * s[2]) % P]
    return s


def poseidon_hash_many(values):
    """starknet-crypto poseidon_hash_many: rate-2 sponge, padded with a single 1."""
    s = [0, 0, 0]
    n = len(values)
    for i in range(0, n - n % 2, 2):
        s[0] = (s[0] + values[i]) % P
        s[1] = (s[1] + values[i + 1]) % P
        s = hades(s)
    if n % 2 == 1:
        s[0] = (s[0] + values[-1]) % P
    s[n % 2] = (s[n % 2] + 1) % P
    return hades(s)[0]


def hash_node(children, column_values):
    """Poseidon252MerkleHasher::hash_node. children: None or (left, right) ints; column_values: M31 ints of this layer's columns."""
    values = list(children) if children is not None else []
    vals = list(column_values)
    vals += [0] * (-len(vals) % 8)
    for b in range(0, len(vals), 8):
        w = 0
        for v in vals[b:b + 8]:
            w = (w * 2**31 + v) % P
        values.append(w)
    return poseidon_hash_many(values)


def merkle_layers(columns_by_log):
    """columns_by_log: {log: [column (list of ints), ...]}. Returns {log: [node hashes]} down to the root (log 0)."""
    max_log = max(columns_by_log)
    layers = {}
    prev = None
    for log in range(max_log, -1, -1):
        cols = columns_by_log.get(log, [])
        layer = []
        for i in range(1 << log):
            ch = None if prev is None else (prev[2 * i], prev[2 * i + 1])
            layer.append(hash_node(ch, [c[i] for c in cols]))
        layers[log] = layer
        prev = layer
    return layers


def self_test():
    assert hades([0, 0, 0]) == HADES_ZERO_KAT, "Hades permutation does not match the public known-answer vector"


if __name__ == "__main__":
    self_test()
    print("hades KAT ok")
