"""MI355X-native Circle-STARK prover backend for the Brainfuck zkVM — Python host mirror over the C ABI (include/bfhip.h).

The product path is libbfhip.so (hand-written gfx950 kernels). There is no CPU fallback: if the library or a GPU is missing,
every entry point raises.  Reference interface mirrored: crates/brainfuck_prover/src/brainfuck_air/mod.rs:471 (prove_brainfuck),
:738 (verify_brainfuck) and stwo's PolyOps/MerkleOps/FriOps/QuotientOps trait surface (SURVEY.md §8 b).
"""
import ctypes
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BFHIP_LIBRARY: another build of the library (A/B runs; tests of the late-host and RCCL-double paths name libbfhip_testhooks.so, the
# -DBFHIP_TEST_HOOKS build — the default library has no test hooks)
_LIB_PATH = os.environ.get("BFHIP_LIBRARY") or os.path.join(_HERE, "libbfhip.so")
TESTHOOKS_LIBRARY = os.path.join(_HERE, "libbfhip_testhooks.so")
_lib = None

P = (1 << 31) - 1


class BfhipError(RuntimeError):
    pass


def lib():
    """Load libbfhip.so (built in-tree by `make -C stwo-brainfuck_amd/csrc`). Fails loudly when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise BfhipError(f"{_LIB_PATH} is missing: build it with __graft_entry__.build(); there is no CPU fallback")
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.bfhip_last_error.restype = ctypes.c_char_p
    return _lib


def _check(rc):
    if rc != 0:
        raise BfhipError(lib().bfhip_last_error().decode())


def device_count():
    return lib().bfhip_device_count()


def device_memory(device_id=0):
    """bfhip_device_memory: (free, total) bytes of one GPU right now."""
    f, t = ctypes.c_uint64(), ctypes.c_uint64()
    _check(lib().bfhip_device_memory(device_id, ctypes.byref(f), ctypes.byref(t)))
    return f.value, t.value


def rccl_unique_id() -> bytes:
    """bfhip_rccl_unique_id: the 128-byte id rank 0 creates for a multi-process shard group (loads librccl on first use)."""
    buf = (ctypes.c_uint8 * 128)()
    _check(lib().bfhip_rccl_unique_id(buf))
    return bytes(buf)


class LocalGroup:
    """Rendezvous object of an in-process shard group (bfhip_local_group_create): `count` contexts of this process, one thread each."""

    def __init__(self, count):
        self._h = ctypes.c_void_p()
        _check(lib().bfhip_local_group_create(count, ctypes.byref(self._h)))
        self.count = count

    def close(self):
        if self._h:
            lib().bfhip_local_group_destroy(self._h)
            self._h = ctypes.c_void_p()


class Conventions(ctypes.Structure):
    """include/bfhip.h `bfhip_conventions`: the byte-level stwo conventions that cannot be confirmed offline, one switch each.
    All zero = the defaults (zero-state raw-compress Merkle nodes, raw-compress mix_u64, logUp mask order [0, -1])."""
    _fields_ = [("merkle_node_hash", ctypes.c_uint32), ("mix_u64", ctypes.c_uint32), ("logup_mask_order", ctypes.c_uint32), ("merkle_channel", ctypes.c_uint32),
                ("reserved", ctypes.c_uint32 * 4)]


_default_conventions = (0, 0, 0, 0)
_live_contexts = weakref.WeakSet()


def set_default_conventions(merkle_node_hash=0, mix_u64=0, logup_mask_order=0, merkle_channel=0):
    """Process-wide default of the Python mirror: adopted by every Context created afterwards, applied to the live ones, and used by
    verify_brainfuck(conventions=None). The C ABI itself has no global state: conventions are per context / per verify call."""
    global _default_conventions
    _default_conventions = (int(merkle_node_hash), int(mix_u64), int(logup_mask_order), int(merkle_channel))
    for c in list(_live_contexts):
        if c._h:
            c.set_conventions(*_default_conventions)


CHANNEL_BLAKE2S, CHANNEL_POSEIDON252 = 0, 1
MERKLE_STWO_COMPRESS, MERKLE_RFC7693 = 0, 1
MIX_U64_COMPRESS, MIX_U64_HASH = 0, 1
LOGUP_MASK_CUR_PREV, LOGUP_MASK_PREV_CUR = 0, 1


class Context:
    """One GPU + one HIP stream + the twiddle tree (mod.rs:480-487: twiddles, channel and commitment scheme setup)."""

    def __init__(self, device_id=0, max_log_domain=22):
        self._h = ctypes.c_void_p()
        _check(lib().bfhip_ctx_create(device_id, max_log_domain, ctypes.byref(self._h)))
        self.max_log_domain = max_log_domain
        _live_contexts.add(self)
        if _default_conventions != (0, 0, 0, 0):
            self.set_conventions(*_default_conventions)

    def close(self):
        if self._h:
            lib().bfhip_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _check(lib().bfhip_ctx_sync(self._h))

    def set_conventions(self, merkle_node_hash=0, mix_u64=0, logup_mask_order=0, merkle_channel=0):
        cv = Conventions(merkle_node_hash, mix_u64, logup_mask_order, merkle_channel)
        _check(lib().bfhip_ctx_set_conventions(self._h, ctypes.byref(cv)))

    def get_conventions(self):
        cv = Conventions()
        _check(lib().bfhip_ctx_get_conventions(self._h, ctypes.byref(cv)))
        return cv.merkle_node_hash, cv.mix_u64, cv.logup_mask_order, cv.merkle_channel

    def set_overlap(self, mask=1):
        """bit 0: tree commitment (Merkle beside the transforms of the smaller columns), bit 1: quotients / FRI first-layer tree, bit 2 (shard
        groups): the send-receive of a tree's largest size class on the partner stream beside the transforms of the smaller columns."""
        _check(lib().bfhip_ctx_set_overlap(self._h, int(mask)))

    def set_sync_policy(self, blocking):
        """bfhip_ctx_set_sync_policy: True = the host sleeps in its waits (more waiting contexts than cores), False (default) = polls first."""
        _check(lib().bfhip_ctx_set_sync_policy(self._h, 1 if blocking else 0))

    def set_mailbox(self, mode=-2, timeout_ms=0, test_delay_ms=-1):
        """bfhip_ctx_set_mailbox: mode -2 keeps the mode / -1 automatic / 0 off / 1 on; timeout_ms 0 keeps the timeout; test_delay_ms < 0 keeps
        the (test) delay — a positive delay needs the test-hooks build of the library."""
        _check(lib().bfhip_ctx_set_mailbox(self._h, int(mode), int(timeout_ms), int(test_delay_ms)))

    def last_proof_flags(self):
        """bfhip_ctx_last_proof_flags: {mailbox_order, kept_preprocessed, shared_preprocessed} of the last completed proof."""
        f = ctypes.c_uint32()
        _check(lib().bfhip_ctx_last_proof_flags(self._h, ctypes.byref(f)))
        return {"mailbox_order": bool(f.value & 1), "kept_preprocessed": bool(f.value & 2), "shared_preprocessed": bool(f.value & 4), "replicated_transforms": bool(f.value & 8)}

    def clock_probe(self, seconds=0.6):
        """bfhip_clock_probe: {ghz (median over workgroups), ghz_min, ghz_max, G_compressions_per_s, launches, ms_per_launch} of a register-only
        Blake2s loop run back to back for `seconds` — the clock this device sustains under the dominant kernel's instruction mix."""
        out = (ctypes.c_double * 6)()
        _check(lib().bfhip_clock_probe(self._h, ctypes.c_double(seconds), out))
        return dict(zip(("ghz", "ghz_min", "ghz_max", "G_compressions_per_s", "launches", "ms_per_launch"), [float(v) for v in out]))

    def clock_probe_mix(self, seconds=0.5, log_nodes=22):
        """bfhip_clock_probe_mix: {ghz, G_compressions_per_s, launches, us_per_launch, sampler_spanned_the_window, sampler_seconds} — the clock held under the REAL
        Merkle kernel (sidecar sampler) and that kernel's rate on an inner layer of 2^log_nodes nodes."""
        out = (ctypes.c_double * 6)()
        _check(lib().bfhip_clock_probe_mix(self._h, ctypes.c_double(seconds), int(log_nodes), out))
        d = dict(zip(("ghz", "G_compressions_per_s", "launches", "us_per_launch", "sampler_spanned_the_window", "sampler_seconds"), [float(v) for v in out]))
        d["sampler_spanned_the_window"] = bool(d["sampler_spanned_the_window"])
        return d

    def memory(self):
        """bfhip_ctx_memory: {arena_reserved, arena_peak, twiddles, arena_in_use} in bytes."""
        out = (ctypes.c_uint64 * 4)()
        _check(lib().bfhip_ctx_memory(self._h, out))
        return dict(zip(("arena_reserved", "arena_peak", "twiddles", "arena_in_use"), [int(v) for v in out]))

    def set_table_builder(self, on_gpu=True):
        """Where this context builds the 13 component tables: GPU kernels (default) or the host builders. Identical results."""
        _check(lib().bfhip_ctx_set_table_builder(self._h, int(on_gpu)))

    # -- one proof over several GPUs (shard group: bfhip_ctx_join_*_group) -----------------------------------------------------------------
    def join_local_group(self, group, rank):
        """Makes this context rank `rank` of a LocalGroup: N contexts of this process, one host thread each, prove ONE trace together."""
        _check(lib().bfhip_ctx_join_local_group(self._h, group._h, rank))
        self._group = group          # keep the rendezvous object alive as long as the context uses it

    def join_rccl_group(self, unique_id: bytes, rank, count):
        """One process per GPU: `unique_id` is rccl_unique_id() of rank 0, handed over by the host program (replicas.share_unique_id)."""
        if len(unique_id) != 128:
            raise BfhipError("an RCCL unique id has 128 bytes")
        _check(lib().bfhip_ctx_join_rccl_group(self._h, unique_id, rank, count))

    def set_shard_policy(self, policy=-1):
        """bfhip_ctx_set_shard_policy: -1 automatic, 0 exchange columns -> rows, 1 replicate the transforms (every rank of a group the same)."""
        _check(lib().bfhip_ctx_set_shard_policy(self._h, int(policy)))

    def leave_group(self):
        _check(lib().bfhip_ctx_leave_group(self._h))
        self._group = None

    def group_stats(self):
        """{all_gathers, max_reduces, exchanges, bytes_sent} of this rank since it joined its shard group."""
        out = (ctypes.c_uint64 * 4)()
        _check(lib().bfhip_ctx_group_stats(self._h, out))
        return dict(zip(("all_gathers", "max_reduces", "exchanges", "bytes_sent"), [int(v) for v in out]))

    def group_times(self):
        """{all_gather_ms, max_reduce_ms, exchange_ms}: GPU-side time of this rank's collectives since it joined its shard group."""
        out = (ctypes.c_double * 3)()
        _check(lib().bfhip_ctx_group_times(self._h, out))
        return dict(zip(("all_gather_ms", "max_reduce_ms", "exchange_ms"), [float(v) for v in out]))

    def group_latency(self, reset=False):
        """bfhip_ctx_group_latency: per kind of collective {count, gpu_us: {p50, p90, max}, host_us: {p50, p90, max}} since the join / the last reset."""
        out = (ctypes.c_double * 21)()
        _check(lib().bfhip_ctx_group_latency(self._h, 1 if reset else 0, out))
        res = {}
        for k, name in enumerate(("all_gather", "max_reduce", "exchange")):
            v = [float(x) for x in out[7 * k:7 * k + 7]]
            res[name] = {"count": int(v[0]), "gpu_us": {"p50": round(v[1], 1), "p90": round(v[2], 1), "max": round(v[3], 1)},
                         "host_us": {"p50": round(v[4], 1), "p90": round(v[5], 1), "max": round(v[6], 1)}}
        return res

    def group_info(self):
        r, n, t = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_char_p()
        _check(lib().bfhip_ctx_group_info(self._h, ctypes.byref(r), ctypes.byref(n), ctypes.byref(t)))
        return r.value, n.value, (t.value or b"").decode()

    # -- buffers ---------------------------------------------------------------------------------------------------
    def malloc(self, nbytes):
        p = ctypes.c_void_p()
        _check(lib().bfhip_malloc(self._h, ctypes.c_size_t(nbytes), ctypes.byref(p)))
        return p.value

    def free(self, ptr):
        _check(lib().bfhip_free(self._h, ctypes.c_void_p(ptr)))

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        ptr = self.malloc(arr.nbytes)
        _check(lib().bfhip_upload(self._h, ctypes.c_void_p(ptr), arr.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(arr.nbytes)))
        return ptr

    def download(self, ptr, n, dtype=np.uint32):
        out = np.empty(n, dtype=dtype)
        _check(lib().bfhip_download(self._h, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), ctypes.c_size_t(out.nbytes)))
        return out

    @staticmethod
    def _ptr_array(ptrs):
        return (ctypes.c_void_p * len(ptrs))(*ptrs)

    # -- PolyOps ---------------------------------------------------------------------------------------------------
    def interpolate(self, src_ptrs, dst_ptrs, log_size, replicated=False):
        _check(lib().bfhip_interpolate(self._h, self._ptr_array(src_ptrs), self._ptr_array(dst_ptrs), len(src_ptrs), log_size, int(replicated)))

    def evaluate(self, coeff_ptrs, dst_ptrs, log_size, log_eval, replicated=False):
        _check(lib().bfhip_evaluate(self._h, self._ptr_array(coeff_ptrs), self._ptr_array(dst_ptrs), len(coeff_ptrs), log_size, log_eval, int(replicated)))

    def is_first_coeffs(self, log_min, log_max, dst_ptrs):
        """interpolate(gen_is_first(n)) for n = log_min..log_max in closed form (mod.rs:497); dst_ptrs[n - log_min] may be None."""
        arr = (ctypes.c_void_p * len(dst_ptrs))(*[p or None for p in dst_ptrs])
        _check(lib().bfhip_is_first_coeffs(self._h, log_min, log_max, arr))

    def eval_at_point(self, coeff_ptr, log_size, point8, replicated=False):
        pt = (ctypes.c_uint32 * 8)(*[int(v) for v in point8])
        out = (ctypes.c_uint32 * 4)()
        _check(lib().bfhip_eval_at_point(self._h, ctypes.c_void_p(coeff_ptr), log_size, int(replicated), pt, out))
        return list(out)

    # -- ColumnOps / FieldOps / AccumulationOps ---------------------------------------------------------------------------------
    def broadcast16(self, rows_ptr, dst_ptr, n_rows):
        _check(lib().bfhip_broadcast16(self._h, ctypes.c_void_p(rows_ptr), ctypes.c_void_p(dst_ptr), ctypes.c_size_t(n_rows)))

    def bit_reverse(self, src_ptr, dst_ptr, log_size):
        _check(lib().bfhip_bit_reverse(self._h, ctypes.c_void_p(src_ptr), ctypes.c_void_p(dst_ptr), log_size))

    def batch_inverse_m31(self, src_ptr, dst_ptr, n):
        _check(lib().bfhip_batch_inverse_m31(self._h, ctypes.c_void_p(src_ptr), ctypes.c_void_p(dst_ptr), ctypes.c_size_t(n)))

    def batch_inverse_qm31(self, src_ptrs, dst_ptrs, n):
        _check(lib().bfhip_batch_inverse_qm31(self._h, self._ptr_array(src_ptrs), self._ptr_array(dst_ptrs), ctypes.c_size_t(n)))

    def accumulate(self, dst_ptr, src_ptr, n):
        _check(lib().bfhip_accumulate(self._h, ctypes.c_void_p(dst_ptr), ctypes.c_void_p(src_ptr), ctypes.c_size_t(n)))

    # -- MerkleOps / FriOps / GrindOps -------------------------------------------------------------------------------------------------
    def merkle_commit_layer(self, log_size, prev_ptr, col_ptrs, out_ptr, col_shifts=None):
        sh = None if col_shifts is None else (ctypes.c_uint32 * len(col_shifts))(*col_shifts)
        _check(lib().bfhip_merkle_commit_layer(self._h, log_size, ctypes.c_void_p(prev_ptr) if prev_ptr else None, self._ptr_array(col_ptrs), sh, len(col_ptrs), ctypes.c_void_p(out_ptr)))

    def merkle_commit_layer_poseidon252(self, log_size, prev_ptr, col_ptrs, out_ptr, col_shifts=None):
        sh = None if col_shifts is None else (ctypes.c_uint32 * len(col_shifts))(*col_shifts)
        _check(lib().bfhip_merkle_commit_layer_poseidon252(self._h, log_size, ctypes.c_void_p(prev_ptr) if prev_ptr else None, self._ptr_array(col_ptrs), sh, len(col_ptrs), ctypes.c_void_p(out_ptr)))

    def hades_permutation(self, state3):
        """state3: three Python ints < p. Returns three ints."""
        words = []
        for x in state3:
            words += [(int(x) >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
        out = (ctypes.c_uint32 * 24)()
        _check(lib().bfhip_hades_permutation(self._h, (ctypes.c_uint32 * 24)(*words), out))
        return [sum(int(out[8 * k + i]) << (32 * i) for i in range(8)) for k in range(3)]

    def fold_line(self, src_ptrs, dst_ptrs, log_size, alpha4):
        _check(lib().bfhip_fold_line(self._h, self._ptr_array(src_ptrs), self._ptr_array(dst_ptrs), log_size, (ctypes.c_uint32 * 4)(*[int(v) for v in alpha4])))

    def fold_circle_into_line(self, dst_ptrs, src_ptrs, log_size, alpha4):
        _check(lib().bfhip_fold_circle_into_line(self._h, self._ptr_array(dst_ptrs), self._ptr_array(src_ptrs), log_size, (ctypes.c_uint32 * 4)(*[int(v) for v in alpha4])))

    def grind(self, digest32, pow_bits):
        nonce = ctypes.c_uint64()
        _check(lib().bfhip_grind(self._h, bytes(digest32), pow_bits, ctypes.byref(nonce)))
        return nonce.value

    def gather(self, col_ptr, indices):
        idx = np.ascontiguousarray(indices, dtype=np.uint64)
        out = np.empty(idx.size, dtype=np.uint32)
        _check(lib().bfhip_gather(self._h, ctypes.c_void_p(col_ptr), idx.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(idx.size), out.ctypes.data_as(ctypes.c_void_p)))
        return out

    # -- per-component AIR operations (LogupTraceGenerator / ComponentProver / QuotientOps) --------------------------------------------
    @staticmethod
    def _u32s(values):
        values = [int(v) for v in values]
        return (ctypes.c_uint32 * len(values))(*values)

    def logup_generate(self, component, log_size, main_row_ptrs, lookup24, out_col_ptrs):
        """interaction_trace_evaluation of one component; returns its claimed sum (4 u32)."""
        claimed = (ctypes.c_uint32 * 4)()
        _check(lib().bfhip_logup_generate(self._h, component, log_size, self._ptr_array(main_row_ptrs), self._u32s(lookup24), self._ptr_array(out_col_ptrs), claimed))
        return list(claimed)

    def eval_constraints(self, component, log_size, is_first_ptr, main_lde_ptrs, inter_lde_ptrs, lookup24, claimed4, coeffs, acc_ptrs, main_shifts=None, inter_shifts=None):
        """evaluate_constraint_quotients_on_domain of one component, accumulated into the 4 coordinate columns acc_ptrs."""
        _check(lib().bfhip_eval_constraints(self._h, component, log_size, ctypes.c_void_p(is_first_ptr), self._ptr_array(main_lde_ptrs),
                                            None if main_shifts is None else self._u32s(main_shifts), self._ptr_array(inter_lde_ptrs),
                                            None if inter_shifts is None else self._u32s(inter_shifts), self._u32s(lookup24), self._u32s(claimed4),
                                            self._u32s(coeffs), self._ptr_array(acc_ptrs)))

    def accumulate_quotients(self, log_size, col_ptrs, n_samples, sample_points, sample_values, random_coeff4, out_ptrs, col_shifts=None):
        """QuotientOps::accumulate_quotients for the columns of one LDE size."""
        _check(lib().bfhip_accumulate_quotients(self._h, log_size, self._ptr_array(col_ptrs), None if col_shifts is None else self._u32s(col_shifts), len(col_ptrs),
                                                self._u32s(n_samples), self._u32s(sample_points), self._u32s(sample_values), self._u32s(random_coeff4),
                                                self._ptr_array(out_ptrs)))

    def twiddles(self):
        tw, itw, rl = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint32()
        _check(lib().bfhip_twiddles(self._h, ctypes.byref(tw), ctypes.byref(itw), ctypes.byref(rl)))
        return tw.value, itw.value, rl.value


class Pool:
    """bfhip_pool_create: `n_in_flight` sub-contexts on one GPU behind ONE caller thread — prove_batch() hands the library a batch of resident
    traces and returns when all are proved, n_in_flight at a time on the library's own worker threads. The sub-contexts share one twiddle
    tree and (preprocessed=1, default) one preprocessed commitment per batch; 0 = every proof recommits it like the reference
    (mod.rs:495-500), 2 = kept across batches."""

    def __init__(self, device_id=0, n_in_flight=2, max_log_domain=24, preprocessed=1):
        self._h = ctypes.c_void_p()
        _check(lib().bfhip_pool_create(device_id, n_in_flight, max_log_domain, ctypes.byref(self._h)))
        self.n_in_flight, self.max_log_domain = n_in_flight, max_log_domain
        self._subs = {}
        if preprocessed != 1:
            self.set_preprocessed(preprocessed)
        if _default_conventions != (0, 0, 0, 0):
            self.set_conventions(*_default_conventions)

    def close(self):
        if self._h:
            for c in self._subs.values():
                c._h = ctypes.c_void_p()          # borrowed handles die with the pool
            lib().bfhip_pool_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def ctx(self, i=0):
        """Sub-context i as a (borrowed) Context: for Trace(...) on the pool's device between batches, per-context settings, memory()."""
        if i not in self._subs:
            h = ctypes.c_void_p()
            _check(lib().bfhip_pool_ctx(self._h, i, ctypes.byref(h)))
            c = Context.__new__(Context)
            c._h, c.max_log_domain = h, self.max_log_domain
            c.close = lambda: None                # owned by the pool
            self._subs[i] = c
        return self._subs[i]

    def set_conventions(self, merkle_node_hash=0, mix_u64=0, logup_mask_order=0, merkle_channel=0):
        cv = Conventions(merkle_node_hash, mix_u64, logup_mask_order, merkle_channel)
        _check(lib().bfhip_pool_set_conventions(self._h, ctypes.byref(cv)))

    def set_preprocessed(self, mode):
        _check(lib().bfhip_pool_set_preprocessed(self._h, int(mode)))

    def _outputs(self, n, want_json):
        js = (ctypes.c_void_p * n)() if want_json else None
        return js, (ctypes.c_size_t * n)(), (ctypes.c_int32 * n)(), (ctypes.c_double * (n + 1))()

    def _collect(self, rc, n, js, lens, st, sec, want_json):
        proofs = []
        for i in range(n):
            if want_json and js[i]:
                proofs.append(ctypes.string_at(js[i], lens[i]))
                lib().bfhip_free_host(ctypes.c_void_p(js[i]))
            else:
                proofs.append(None)
        info = {"statuses": [int(v) for v in st], "seconds": [float(v) for v in sec[:n]], "batch_seconds": float(sec[n])}
        if rc != 0:
            err = BfhipError(lib().bfhip_last_error().decode())
            err.proofs, err.info = proofs, info
            raise err
        return proofs, info

    def prove_batch(self, traces, log_max_rows=24, want_json=True):
        """bfhip_prove_batch: returns (proofs, info) — proofs[i] = JSON bytes of traces[i]'s proof; info = per-proof seconds + batch_seconds.
        Raises BfhipError when a proof failed (its .proofs / .info hold what the rest of the batch produced)."""
        n = len(traces)
        arr = (ctypes.c_void_p * max(n, 1))(*[t._h for t in traces])
        js, lens, st, sec = self._outputs(n, want_json)
        rc = lib().bfhip_prove_batch(self._h, arr, n, log_max_rows, js, lens, st, sec)
        return self._collect(rc, n, js, lens, st, sec, want_json)

    def prove_batch_brainfuck(self, programs, log_max_rows=24, want_json=True):
        """bfhip_prove_batch_brainfuck: programs = [(code, input_bytes), ...]; VM, table build and upload run inside the workers."""
        n = len(programs)
        codes = (ctypes.c_char_p * max(n, 1))(*[c.encode() for c, _ in programs])
        bufs = [ctypes.create_string_buffer(bytes(i), max(len(i), 1)) for _, i in programs]
        inputs = (ctypes.c_void_p * max(n, 1))(*[ctypes.addressof(b) for b in bufs])
        nin = (ctypes.c_size_t * max(n, 1))(*[len(i) for _, i in programs])
        js, lens, st, sec = self._outputs(n, want_json)
        rc = lib().bfhip_prove_batch_brainfuck(self._h, codes, inputs, nin, n, log_max_rows, js, lens, st, sec)
        return self._collect(rc, n, js, lens, st, sec, want_json)


PHASES = ("preprocessed", "tables_host", "main_trace", "interaction", "composition", "oods", "quotients", "fri", "decommit", "total")


def prove_brainfuck(code, input_bytes=b"", ctx=None, log_max_rows=24, with_transcript=False, with_timings=False):
    """prove_brainfuck (mod.rs:471): returns the proof as serde_json bytes of BrainfuckProof. GPU only."""
    own = ctx is None
    if own:
        ctx = Context(0, max_log_domain=log_max_rows + 2)
    try:
        js, n, tr = ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_void_p()
        times = (ctypes.c_double * 10)()
        _check(lib().bfhip_prove_brainfuck(ctx._h, code.encode(), input_bytes, ctypes.c_size_t(len(input_bytes)), log_max_rows,
                                           ctypes.byref(js), ctypes.byref(n), ctypes.byref(tr) if with_transcript else None, times))
        proof = ctypes.string_at(js, n.value)
        lib().bfhip_free_host(js)
        out = [proof]
        if with_transcript:
            t = ctypes.string_at(tr).decode()
            lib().bfhip_free_host(tr)
            out.append(dict(line.split(":") for line in t.strip().split("\n")))
        if with_timings:
            out.append(dict(zip(PHASES, list(times))))
        return out[0] if len(out) == 1 else tuple(out)
    finally:
        if own:
            ctx.close()


def verify_brainfuck(proof_json: bytes, log_max_rows=24, conventions=None):
    """verify_brainfuck (mod.rs:738): returns (ok, reason). Host only — no GPU needed, like the reference's verifier.
    conventions: (merkle_node_hash, mix_u64, logup_mask_order) the proof was produced under; None = the defaults."""
    err = ctypes.create_string_buffer(512)
    cv = ctypes.byref(Conventions(*(_default_conventions if conventions is None else conventions)))
    rc = lib().bfhip_verify_brainfuck_conv(proof_json, ctypes.c_size_t(len(proof_json)), log_max_rows, cv, err, ctypes.c_size_t(512))
    if rc < 0:
        raise BfhipError(lib().bfhip_last_error().decode())
    return rc == 0, err.value.decode()


class Trace:
    """Prover input resident in HBM (bfhip_trace_create): VM trace -> 13 component tables -> row-granular device columns."""

    def __init__(self, ctx, code, input_bytes=b"", ram_size=0):
        """ram_size: Machine RAM cells (MachineBuilder::with_ram_size, machine.rs:56-60); 0 = the default 30000."""
        self.ctx = ctx
        self._h = ctypes.c_void_p()
        ls = (ctypes.c_uint32 * 13)()
        steps, mc, ic = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        _check(lib().bfhip_trace_create_ram(ctx._h, code.encode(), input_bytes, ctypes.c_size_t(len(input_bytes)), ctypes.c_size_t(ram_size),
                                            ctypes.byref(self._h), ls, ctypes.byref(steps), ctypes.byref(mc), ctypes.byref(ic)))
        self.log_sizes = list(ls)
        self.n_steps, self.main_cells, self.interaction_cells = steps.value, mc.value, ic.value

    @classmethod
    def from_registers(cls, ctx, trace7, code_words):
        """What prove_brainfuck(&Machine) receives (mod.rs:471-473,508): the executed machine's register trace (n x 7 u32: clk, ip, ci,
        ni, mp, mv, mvi) and its compiled program words. No re-execution."""
        self = cls.__new__(cls)
        self.ctx = ctx
        self._h = ctypes.c_void_p()
        tr = np.ascontiguousarray(trace7, dtype=np.uint32).reshape(-1, 7)
        code = np.ascontiguousarray(code_words, dtype=np.uint32)
        ls = (ctypes.c_uint32 * 13)()
        mc, ic = ctypes.c_uint64(), ctypes.c_uint64()
        _check(lib().bfhip_trace_create_from_registers(ctx._h, tr.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(tr.shape[0]), code.ctypes.data_as(ctypes.c_void_p),
                                                       ctypes.c_size_t(code.size), ctypes.byref(self._h), ls, ctypes.byref(mc), ctypes.byref(ic)))
        self.log_sizes = list(ls)
        self.n_steps, self.main_cells, self.interaction_cells = tr.shape[0], mc.value, ic.value
        return self

    @property
    def cells(self):
        return self.main_cells + self.interaction_cells

    def prove(self, log_max_rows=24, want_json=True):
        js, n = ctypes.c_void_p(), ctypes.c_size_t()
        times = (ctypes.c_double * 10)()
        _check(lib().bfhip_prove_trace(self.ctx._h, self._h, log_max_rows, ctypes.byref(js) if want_json else None, ctypes.byref(n), None, times))
        proof = None
        if want_json:
            proof = ctypes.string_at(js, n.value)
            lib().bfhip_free_host(js)
        return proof, dict(zip(PHASES, list(times)))

    def column(self, component, column):
        n = ctypes.c_size_t()
        _check(lib().bfhip_trace_column(self.ctx._h, self._h, component, column, None, ctypes.c_size_t(0), ctypes.byref(n)))
        out = np.empty(n.value, dtype=np.uint32)
        _check(lib().bfhip_trace_column(self.ctx._h, self._h, component, column, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(out.size), ctypes.byref(n)))
        return out

    def close(self):
        if self._h:
            lib().bfhip_trace_destroy(self.ctx._h, self._h)
            self._h = ctypes.c_void_p()
