//! `HipBackend`: stwo's backend trait surface over `libbfhip.so` — the drop-in for `SimdBackend` at the places the reference fixes it
//! (`crates/brainfuck_prover/src/brainfuck_air/mod.rs:56,399,480,486-487,497,732`, `components/mod.rs:42`).
//!
//! A NON-BUILDING SKETCH, SOURCE ONLY (behind the crate feature `stwo-backend`, off by default): the build image has no Rust toolchain and the `stwo-prover` crate (rev 31e8dbc, `Cargo.toml:41`) is not vendored, so this
//! file has never been compiled. It is written against the trait shapes of that revision as far as the reference's call sites and our
//! recollection pin them; every method body is a thin call into `bfhip_sys` (generated from `include/bfhip.h`).
//! `tests/test_abi_and_replicas.py::test_hip_backend_covers_the_trait_surface` keeps the method list in step with INTEGRATION.md §2.
//!
//! Layout decisions that make `prover::prove::<HipBackend, _>` (`mod.rs:732`) type-check:
//!   * `Col<HipBackend, BaseField>` = `HipColumn<BaseField>`: an owned device buffer of canonical `u32` (M31), bit-reversed circle-domain
//!     order exactly like `BaseColumn`; `Col<HipBackend, SecureField>` = `HipSecureColumn`: 4 coordinate buffers (`SecureColumnByCoords`);
//!     `Col<HipBackend, Blake2sHash>` = `HipColumn<Blake2sHash>`: 32-byte records.
//!   * one process-wide context per GPU (`ctx()`), created on first use with the twiddle tree of `LOG_MAX_ROWS + 2` (`mod.rs:480-484`).
//!   * `ComponentProver<HipBackend>` is implemented for the 13 `FrameworkComponent<XEval>` types through `bfhip_eval_constraints`
//!     (upstream only implements it for `SimdBackend`, SURVEY.md §8(b) caveat).
#![allow(dead_code, unused_variables)]

use std::ffi::c_void;
use std::marker::PhantomData;
use std::sync::OnceLock;

use stwo_prover::constraint_framework::{FrameworkComponent, FrameworkEval};
use stwo_prover::core::air::accumulation::{AccumulationOps, DomainEvaluationAccumulator};
use stwo_prover::core::air::{ComponentProver, Trace};
use stwo_prover::core::backend::{Backend, BackendForChannel, Col, Column, ColumnOps};
use stwo_prover::core::channel::Blake2sChannel;
use stwo_prover::core::circle::{CirclePoint, Coset};
use stwo_prover::core::fields::m31::BaseField;
use stwo_prover::core::fields::qm31::SecureField;
use stwo_prover::core::fields::secure_column::SecureColumnByCoords;
use stwo_prover::core::fields::FieldOps;
use stwo_prover::core::fri::FriOps;
use stwo_prover::core::lookups::gkr_prover::GkrOps;
use stwo_prover::core::pcs::quotients::{ColumnSampleBatch, QuotientOps};
use stwo_prover::core::poly::circle::{CanonicCoset, CircleDomain, CircleEvaluation, CirclePoly, PolyOps, SecureEvaluation};
use stwo_prover::core::poly::line::LineEvaluation;
use stwo_prover::core::poly::twiddles::TwiddleTree;
use stwo_prover::core::poly::BitReversedOrder;
use stwo_prover::core::proof_of_work::GrindOps;
use stwo_prover::core::vcs::blake2_hash::Blake2sHash;
use stwo_prover::core::vcs::blake2_merkle::{Blake2sMerkleChannel, Blake2sMerkleHasher};
use stwo_prover::core::vcs::ops::MerkleOps;

use crate::sys;

/// `LOG_MAX_ROWS` of the reference (`mod.rs:428`).
pub const LOG_MAX_ROWS: u32 = 24;

#[derive(Copy, Clone, Debug, Default)]
pub struct HipBackend;

struct CtxHandle(*mut sys::BfhipCtx);
unsafe impl Send for CtxHandle {}

/// Exclusive use of the process-wide context for the duration of one FFI call: the C context requires SERIAL calls (one stream, one staging
/// ring), while stwo's prover may call backend operations from rayon workers (`--features parallel`). The guard is a temporary of the call
/// expression, so the lock is held exactly as long as the call (no `Sync` claim on a bare handle).
pub struct CtxGuard(std::sync::MutexGuard<'static, CtxHandle>);
impl CtxGuard {
    pub fn p(&self) -> *mut sys::BfhipCtx { (self.0).0 }
}

/// The process-wide context of GPU 0: stream, twiddle tree (`SimdBackend::precompute_twiddles(CanonicCoset::new(LOG_MAX_ROWS + 2 + blowup)…)`,
/// `mod.rs:480-484`) and scratch memory. One context per GPU; calls on it are serialised by the mutex.
fn ctx() -> CtxGuard {
    static CTX: OnceLock<std::sync::Mutex<CtxHandle>> = OnceLock::new();
    let m = CTX.get_or_init(|| {
        let mut p = std::ptr::null_mut();
        check(unsafe { sys::bfhip_ctx_create(0, LOG_MAX_ROWS + 2, &mut p) });
        std::sync::Mutex::new(CtxHandle(p))
    });
    CtxGuard(m.lock().unwrap_or_else(|e| e.into_inner()))
}

fn check(rc: i32) {
    if rc != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(sys::bfhip_last_error()) }.to_string_lossy().into_owned();
        panic!("bfhip: {msg}"); // the reference unwraps at the same places (mod.rs:511-547)
    }
}

fn qm31_words(x: SecureField) -> [u32; 4] {
    let a = x.to_m31_array();
    [a[0].0, a[1].0, a[2].0, a[3].0]
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Columns in HBM
// ---------------------------------------------------------------------------------------------------------------------------------------

/// An owned device buffer of `len` elements of `T` (`BaseField` = canonical u32, `Blake2sHash` = 32-byte record).
#[derive(Debug)]
pub struct HipColumn<T> {
    pub ptr: *mut u32,
    pub len: usize,
    _t: PhantomData<T>,
}
unsafe impl<T> Send for HipColumn<T> {}
unsafe impl<T> Sync for HipColumn<T> {}

impl<T> HipColumn<T> {
    fn words_per_elem() -> usize { std::mem::size_of::<T>() / 4 }
    fn alloc(len: usize) -> Self {
        let mut p: *mut c_void = std::ptr::null_mut();
        check(unsafe { sys::bfhip_malloc(ctx().p(), len * std::mem::size_of::<T>(), &mut p) });
        HipColumn { ptr: p as *mut u32, len, _t: PhantomData }
    }
}
impl<T> Drop for HipColumn<T> {
    fn drop(&mut self) { unsafe { sys::bfhip_free(ctx().p(), self.ptr as *mut c_void) }; }
}
impl<T> Clone for HipColumn<T> {
    fn clone(&self) -> Self {
        let c = Self::alloc(self.len);
        let host = self.download_words();
        check(unsafe { sys::bfhip_upload(ctx().p(), c.ptr as *mut c_void, host.as_ptr() as *const c_void, host.len() * 4) });
        c
    }
}
impl<T> HipColumn<T> {
    fn download_words(&self) -> Vec<u32> {
        let mut v = vec![0u32; self.len * Self::words_per_elem()];
        check(unsafe { sys::bfhip_download(ctx().p(), v.as_mut_ptr() as *mut c_void, self.ptr as *const c_void, v.len() * 4) });
        v
    }
}

impl Column<BaseField> for HipColumn<BaseField> {
    fn zeros(len: usize) -> Self {
        let c = Self::alloc(len);
        check(unsafe { sys::bfhip_memset_zero(ctx().p(), c.ptr as *mut c_void, len * 4) });
        c
    }
    unsafe fn uninitialized(len: usize) -> Self { Self::alloc(len) }
    fn to_cpu(&self) -> Vec<BaseField> { self.download_words().into_iter().map(BaseField::from_u32_unchecked).collect() }
    fn len(&self) -> usize { self.len }
    fn at(&self, index: usize) -> BaseField {
        let (idx, mut out) = (index as u64, 0u32);
        check(unsafe { sys::bfhip_gather(ctx().p(), self.ptr, &idx, 1, &mut out) });
        BaseField::from_u32_unchecked(out)
    }
    fn set(&mut self, index: usize, value: BaseField) {
        check(unsafe { sys::bfhip_upload(ctx().p(), self.ptr.add(index) as *mut c_void, &value.0 as *const u32 as *const c_void, 4) });
    }
}
impl FromIterator<BaseField> for HipColumn<BaseField> {
    fn from_iter<I: IntoIterator<Item = BaseField>>(iter: I) -> Self {
        let host: Vec<u32> = iter.into_iter().map(|x| x.0).collect();
        let c = Self::alloc(host.len());
        check(unsafe { sys::bfhip_upload(ctx().p(), c.ptr as *mut c_void, host.as_ptr() as *const c_void, host.len() * 4) });
        c
    }
}
impl Column<Blake2sHash> for HipColumn<Blake2sHash> {
    fn zeros(len: usize) -> Self {
        let c = Self::alloc(len);
        check(unsafe { sys::bfhip_memset_zero(ctx().p(), c.ptr as *mut c_void, len * 32) });
        c
    }
    unsafe fn uninitialized(len: usize) -> Self { Self::alloc(len) }
    fn to_cpu(&self) -> Vec<Blake2sHash> {
        let w = self.download_words();
        w.chunks_exact(8).map(|c| { let mut b = [0u8; 32]; for (i, x) in c.iter().enumerate() { b[4 * i..4 * i + 4].copy_from_slice(&x.to_le_bytes()); } Blake2sHash(b) }).collect()
    }
    fn len(&self) -> usize { self.len }
    fn at(&self, index: usize) -> Blake2sHash {
        let idx: Vec<u64> = (0..8).map(|k| (8 * index + k) as u64).collect();
        let mut out = [0u32; 8];
        check(unsafe { sys::bfhip_gather(ctx().p(), self.ptr, idx.as_ptr(), 8, out.as_mut_ptr()) });
        let mut b = [0u8; 32];
        for (i, x) in out.iter().enumerate() { b[4 * i..4 * i + 4].copy_from_slice(&x.to_le_bytes()); }
        Blake2sHash(b)
    }
    fn set(&mut self, index: usize, value: Blake2sHash) {
        check(unsafe { sys::bfhip_upload(ctx().p(), self.ptr.add(8 * index) as *mut c_void, value.0.as_ptr() as *const c_void, 32) });
    }
}
impl FromIterator<Blake2sHash> for HipColumn<Blake2sHash> {
    fn from_iter<I: IntoIterator<Item = Blake2sHash>>(iter: I) -> Self {
        let host: Vec<u8> = iter.into_iter().flat_map(|h| h.0).collect();
        let c = Self::alloc(host.len() / 32);
        check(unsafe { sys::bfhip_upload(ctx().p(), c.ptr as *mut c_void, host.as_ptr() as *const c_void, host.len()) });
        c
    }
}

/// `Col<HipBackend, SecureField>`: the secure column as 4 coordinate buffers (the layout of `SecureColumnByCoords`).
#[derive(Clone, Debug)]
pub struct HipSecureColumn { pub c: [HipColumn<BaseField>; 4] }
impl HipSecureColumn {
    fn ptrs(&self) -> [*const u32; 4] { [self.c[0].ptr, self.c[1].ptr, self.c[2].ptr, self.c[3].ptr] }
    fn ptrs_mut(&mut self) -> [*mut u32; 4] { [self.c[0].ptr, self.c[1].ptr, self.c[2].ptr, self.c[3].ptr] }
}
impl Column<SecureField> for HipSecureColumn {
    fn zeros(len: usize) -> Self { HipSecureColumn { c: std::array::from_fn(|_| HipColumn::<BaseField>::zeros(len)) } }
    unsafe fn uninitialized(len: usize) -> Self { HipSecureColumn { c: std::array::from_fn(|_| HipColumn::<BaseField>::uninitialized(len)) } }
    fn to_cpu(&self) -> Vec<SecureField> {
        let v: Vec<Vec<BaseField>> = self.c.iter().map(|c| c.to_cpu()).collect();
        (0..self.len()).map(|i| SecureField::from_m31_array([v[0][i], v[1][i], v[2][i], v[3][i]])).collect()
    }
    fn len(&self) -> usize { self.c[0].len }
    fn at(&self, index: usize) -> SecureField { SecureField::from_m31_array(std::array::from_fn(|k| self.c[k].at(index))) }
    fn set(&mut self, index: usize, value: SecureField) { let a = value.to_m31_array(); for k in 0..4 { self.c[k].set(index, a[k]); } }
}
impl FromIterator<SecureField> for HipSecureColumn {
    fn from_iter<I: IntoIterator<Item = SecureField>>(iter: I) -> Self {
        let v: Vec<SecureField> = iter.into_iter().collect();
        HipSecureColumn { c: std::array::from_fn(|k| v.iter().map(|x| x.to_m31_array()[k]).collect()) }
    }
}
fn coords(col: &SecureColumnByCoords<HipBackend>) -> [*const u32; 4] { std::array::from_fn(|k| col.columns[k].ptr as *const u32) }
fn coords_mut(col: &mut SecureColumnByCoords<HipBackend>) -> [*mut u32; 4] { std::array::from_fn(|k| col.columns[k].ptr) }

// ---------------------------------------------------------------------------------------------------------------------------------------
// Backend, ColumnOps, FieldOps
// ---------------------------------------------------------------------------------------------------------------------------------------
impl Backend for HipBackend {}
impl BackendForChannel<Blake2sMerkleChannel> for HipBackend {}

impl ColumnOps<BaseField> for HipBackend {
    type Column = HipColumn<BaseField>;
    /// `bfhip_bit_reverse` is out of place: permute into a fresh buffer and swap.
    fn bit_reverse_column(column: &mut Self::Column) {
        let dst = HipColumn::<BaseField>::alloc(column.len);
        check(unsafe { sys::bfhip_bit_reverse(ctx().p(), column.ptr, dst.ptr, column.len.ilog2()) });
        *column = dst;
    }
}
impl ColumnOps<SecureField> for HipBackend {
    type Column = HipSecureColumn;
    fn bit_reverse_column(column: &mut Self::Column) { for k in 0..4 { <HipBackend as ColumnOps<BaseField>>::bit_reverse_column(&mut column.c[k]); } }
}
impl ColumnOps<Blake2sHash> for HipBackend {
    type Column = HipColumn<Blake2sHash>;
    fn bit_reverse_column(_column: &mut Self::Column) { unimplemented!("hash columns are never bit-reversed on the prove path") }
}
impl FieldOps<BaseField> for HipBackend {
    fn batch_inverse(column: &Self::Column, dst: &mut Self::Column) { check(unsafe { sys::bfhip_batch_inverse_m31(ctx().p(), column.ptr, dst.ptr, column.len) }); }
}
impl FieldOps<SecureField> for HipBackend {
    /// `LogupTraceGenerator::finalize_col`'s denominators (`memory/table.rs:513`).
    fn batch_inverse(column: &HipSecureColumn, dst: &mut HipSecureColumn) {
        let (s, d) = (column.ptrs(), dst.ptrs_mut());
        check(unsafe { sys::bfhip_batch_inverse_qm31(ctx().p(), s.as_ptr(), d.as_ptr(), column.len()) });
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// PolyOps (`mod.rs:480-484` twiddles, `:497,550-562,690-702` interpolate, `:500,583,723` evaluate)
// ---------------------------------------------------------------------------------------------------------------------------------------
/// The twiddles live inside the context (layered exactly like `slow_precompute_twiddles(Coset::half_odds(R))`); the tree only carries the root.
#[derive(Clone, Debug)]
pub struct HipTwiddles { pub tw: *const u32, pub itw: *const u32, pub root_log: u32 }
unsafe impl Send for HipTwiddles {}
unsafe impl Sync for HipTwiddles {}

impl PolyOps for HipBackend {
    type Twiddles = HipTwiddles;

    fn new_canonical_ordered(coset: CanonicCoset, values: Col<Self, BaseField>) -> CircleEvaluation<Self, BaseField, BitReversedOrder> {
        // canonic-coset order -> circle-domain order -> bit-reversed: a host permutation of indices, one upload (cold path: tests only)
        let host = values.to_cpu();
        let eval = stwo_prover::core::backend::CpuBackend::new_canonical_ordered(coset, host);
        CircleEvaluation::new(coset.circle_domain(), eval.values.into_iter().collect())
    }
    fn precompute_twiddles(coset: Coset) -> TwiddleTree<Self> {
        let (mut tw, mut itw, mut root_log) = (std::ptr::null(), std::ptr::null(), 0u32);
        check(unsafe { sys::bfhip_twiddles(ctx().p(), &mut tw, &mut itw, &mut root_log) });
        assert!(coset.log_size() <= root_log, "context twiddle tree too small: create it with a larger max_log_domain");
        let t = HipTwiddles { tw, itw, root_log };
        TwiddleTree { root_coset: coset, twiddles: t.clone(), itwiddles: t }
    }
    fn interpolate(eval: CircleEvaluation<Self, BaseField, BitReversedOrder>, tw: &TwiddleTree<Self>) -> CirclePoly<Self> {
        Self::interpolate_columns([eval], tw).pop().unwrap()
    }
    /// `tree_builder.extend_evals(...)`: one batched launch per size group, in place.
    fn interpolate_columns(columns: impl IntoIterator<Item = CircleEvaluation<Self, BaseField, BitReversedOrder>>, _tw: &TwiddleTree<Self>) -> Vec<CirclePoly<Self>> {
        let cols: Vec<_> = columns.into_iter().collect();
        let mut by_log: std::collections::BTreeMap<u32, Vec<*mut u32>> = Default::default();
        for c in &cols { by_log.entry(c.domain.log_size()).or_default().push(c.values.ptr); }
        for (log, ptrs) in by_log { check(unsafe { sys::bfhip_interpolate(ctx().p(), ptrs.as_ptr(), ptrs.as_ptr(), ptrs.len() as u32, log, 0) }); }
        cols.into_iter().map(|c| CirclePoly::new(c.values)).collect()
    }
    fn eval_at_point(poly: &CirclePoly<Self>, point: CirclePoint<SecureField>) -> SecureField {
        let (x, y) = (qm31_words(point.x), qm31_words(point.y));
        let p8 = [x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]];
        let mut out = [0u32; 4];
        check(unsafe { sys::bfhip_eval_at_point(ctx().p(), poly.coeffs.ptr, poly.log_size(), 0, p8.as_ptr(), out.as_mut_ptr()) });
        SecureField::from_m31_array(out.map(BaseField::from_u32_unchecked))
    }
    fn extend(poly: &CirclePoly<Self>, log_size: u32) -> CirclePoly<Self> {
        let mut c = HipColumn::<BaseField>::zeros(1 << log_size);
        let host = poly.coeffs.download_words();
        check(unsafe { sys::bfhip_upload(ctx().p(), c.ptr as *mut c_void, host.as_ptr() as *const c_void, host.len() * 4) });
        CirclePoly::new(c)
    }
    fn evaluate(poly: &CirclePoly<Self>, domain: CircleDomain, _tw: &TwiddleTree<Self>) -> CircleEvaluation<Self, BaseField, BitReversedOrder> {
        let out = HipColumn::<BaseField>::alloc(domain.size());
        let (src, dst) = ([poly.coeffs.ptr], [out.ptr]);
        check(unsafe { sys::bfhip_evaluate(ctx().p(), src.as_ptr(), dst.as_ptr(), 1, poly.log_size(), domain.log_size(), 0) });
        CircleEvaluation::new(domain, out)
    }
    /// `tree_builder.commit(channel)` -> LDE by the blowup factor: one batched launch per size group.
    fn evaluate_polynomials(polys: &[CirclePoly<Self>], log_blowup_factor: u32, _tw: &TwiddleTree<Self>) -> Vec<CircleEvaluation<Self, BaseField, BitReversedOrder>> {
        let outs: Vec<HipColumn<BaseField>> = polys.iter().map(|p| HipColumn::alloc(1 << (p.log_size() + log_blowup_factor))).collect();
        let mut by_log: std::collections::BTreeMap<u32, (Vec<*mut u32>, Vec<*mut u32>)> = Default::default();
        for (p, o) in polys.iter().zip(&outs) { let e = by_log.entry(p.log_size()).or_default(); e.0.push(p.coeffs.ptr); e.1.push(o.ptr); }
        for (log, (src, dst)) in by_log { check(unsafe { sys::bfhip_evaluate(ctx().p(), src.as_ptr(), dst.as_ptr(), src.len() as u32, log, log + log_blowup_factor, 0) }); }
        polys.iter().zip(outs).map(|(p, o)| CircleEvaluation::new(CanonicCoset::new(p.log_size() + log_blowup_factor).circle_domain(), o)).collect()
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// MerkleOps, AccumulationOps, QuotientOps, FriOps, GrindOps
// ---------------------------------------------------------------------------------------------------------------------------------------
impl MerkleOps<Blake2sMerkleHasher> for HipBackend {
    fn commit_on_layer(log_size: u32, prev_layer: Option<&HipColumn<Blake2sHash>>, columns: &[&HipColumn<BaseField>]) -> HipColumn<Blake2sHash> {
        let out = HipColumn::<Blake2sHash>::alloc(1 << log_size);
        let ptrs: Vec<*const u32> = columns.iter().map(|c| c.ptr as *const u32).collect();
        let prev = prev_layer.map_or(std::ptr::null(), |p| p.ptr as *const c_void);
        check(unsafe { sys::bfhip_merkle_commit_layer(ctx().p(), log_size, prev, ptrs.as_ptr(), std::ptr::null(), ptrs.len() as u32, out.ptr as *mut c_void) });
        out
    }
}
impl AccumulationOps for HipBackend {
    fn accumulate(column: &mut SecureColumnByCoords<Self>, other: &SecureColumnByCoords<Self>) {
        for k in 0..4 { check(unsafe { sys::bfhip_accumulate(ctx().p(), column.columns[k].ptr, other.columns[k].ptr, other.columns[k].len) }); }
    }
    fn generate_secure_powers(felt: SecureField, n_powers: usize) -> Vec<SecureField> {
        std::iter::successors(Some(SecureField::from(BaseField::from(1))), |x| Some(*x * felt)).take(n_powers).collect()   // 103 values: host
    }
}
impl QuotientOps for HipBackend {
    /// `compute_fri_quotients` calls this once per LDE size. The C entry groups the samples by point itself (`ColumnSampleBatch::new_vec`
    /// order), so the batches are flattened back into per-column sample lists here.
    fn accumulate_quotients(domain: CircleDomain, columns: &[&CircleEvaluation<Self, BaseField, BitReversedOrder>], random_coeff: SecureField,
                            sample_batches: &[ColumnSampleBatch], _log_blowup_factor: u32) -> SecureEvaluation<Self, BitReversedOrder> {
        let mut per_col: Vec<Vec<(CirclePoint<SecureField>, SecureField)>> = vec![vec![]; columns.len()];
        for b in sample_batches { for (col, value) in &b.columns_and_values { per_col[*col].push((b.point, *value)); } }
        let (mut n_samples, mut points, mut values) = (vec![], vec![], vec![]);
        for s in &per_col {
            n_samples.push(s.len() as u32);
            for (p, v) in s { points.extend(qm31_words(p.x)); points.extend(qm31_words(p.y)); values.extend(qm31_words(*v)); }
        }
        let mut out = SecureColumnByCoords::<Self> { columns: std::array::from_fn(|_| HipColumn::alloc(domain.size())) };
        let (ptrs, o) = (columns.iter().map(|c| c.values.ptr as *const u32).collect::<Vec<_>>(), coords_mut(&mut out));
        let rc = qm31_words(random_coeff);
        check(unsafe { sys::bfhip_accumulate_quotients(ctx().p(), domain.log_size(), ptrs.as_ptr(), std::ptr::null(), ptrs.len() as u32, n_samples.as_ptr(), points.as_ptr(),
                                                       values.as_ptr(), rc.as_ptr(), o.as_ptr()) });
        SecureEvaluation::new(domain, out)
    }
}
impl FriOps for HipBackend {
    fn fold_line(eval: &LineEvaluation<Self>, alpha: SecureField, _tw: &TwiddleTree<Self>) -> LineEvaluation<Self> {
        let log = eval.len().ilog2();
        let mut out = SecureColumnByCoords::<Self> { columns: std::array::from_fn(|_| HipColumn::alloc(eval.len() / 2)) };
        let (s, d, a) = (coords(&eval.values), coords_mut(&mut out), qm31_words(alpha));
        check(unsafe { sys::bfhip_fold_line(ctx().p(), s.as_ptr(), d.as_ptr(), log, a.as_ptr()) });
        LineEvaluation::new(eval.domain().double(), out)
    }
    fn fold_circle_into_line(dst: &mut LineEvaluation<Self>, src: &SecureEvaluation<Self, BitReversedOrder>, alpha: SecureField, _tw: &TwiddleTree<Self>) {
        let (d, s, a) = (coords_mut(&mut dst.values), coords(&src.values), qm31_words(alpha));
        check(unsafe { sys::bfhip_fold_circle_into_line(ctx().p(), d.as_ptr(), s.as_ptr(), src.domain.log_size(), a.as_ptr()) });
    }
    fn decompose(_eval: &SecureEvaluation<Self, BitReversedOrder>) -> (SecureEvaluation<Self, BitReversedOrder>, SecureField) {
        unimplemented!("only reached for column sizes outside the FRI log-size range; the reference's PcsConfig::default() never does")
    }
}
impl GrindOps<Blake2sChannel> for HipBackend {
    fn grind(channel: &Blake2sChannel, pow_bits: u32) -> u64 {
        let mut nonce = 0u64;
        check(unsafe { sys::bfhip_grind(ctx().p(), channel.digest().0.as_ptr(), pow_bits, &mut nonce) });
        nonce
    }
}
impl GkrOps for HipBackend {
    // The reference proves with logUp over plain columns, not GKR (GkrOps unused); the trait bound only needs to exist.
    fn gen_eq_evals(_y: &[SecureField], _v: SecureField) -> stwo_prover::core::lookups::mle::Mle<Self, SecureField> { unimplemented!("GKR is not used by the Brainfuck AIR") }
    fn next_layer(_layer: &stwo_prover::core::lookups::gkr_prover::Layer<Self>) -> stwo_prover::core::lookups::gkr_prover::Layer<Self> { unimplemented!("GKR is not used by the Brainfuck AIR") }
    fn sum_as_poly_in_first_variable(_h: &stwo_prover::core::lookups::gkr_prover::GkrMultivariatePolyOracle<'_, Self>, _claim: SecureField)
        -> stwo_prover::core::lookups::utils::UnivariatePoly<SecureField> { unimplemented!("GKR is not used by the Brainfuck AIR") }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// ComponentProver<HipBackend> for the 13 FrameworkComponent<XEval> (`mod.rs:399-415` provers(), `components/<name>/component.rs`)
// ---------------------------------------------------------------------------------------------------------------------------------------
/// Claim-order index of a component eval (the `component` argument of the per-component C entries, `mod.rs:85-99`) and what the constraint
/// kernel needs besides the columns: the three lookup-element pairs and the component's claimed sum.
pub trait BrainfuckEval: FrameworkEval {
    const COMPONENT: i32;
    /// (z, alpha) of the Memory, Instruction and Processor relations in that order — 24 words (unused relations may be anything).
    fn lookup_words(&self) -> [u32; 24];
    fn claimed_sum(&self) -> SecureField;
}

impl<E: BrainfuckEval> ComponentProver<HipBackend> for FrameworkComponent<E> {
    /// One fused kernel per AIR: every constraint of the component on the LDE domain, each times its random-coefficient power and the inverse
    /// vanishing polynomial, accumulated into the accumulator of that size (`bfhip_eval_constraints`).
    fn evaluate_constraint_quotients_on_domain(&self, trace: &Trace<'_, HipBackend>, evaluation_accumulator: &mut DomainEvaluationAccumulator<HipBackend>) {
        let log_size = self.log_size();
        let (mut n_main, mut n_logup, mut n_cons) = (0u32, 0u32, 0u32);
        check(unsafe { sys::bfhip_component_shape(E::COMPONENT, &mut n_main, &mut n_logup, &mut n_cons) });
        // the component's slices of the committed trees (tree 0 preprocessed, 1 main, 2 interaction) at the LDE size log_size + 1
        let locs = self.trace_locations();
        let sub = trace.evals.sub_tree(locs);
        let is_first = sub[0][0].values.ptr as *const u32;
        let main: Vec<*const u32> = sub[1].iter().map(|e| e.values.ptr as *const u32).collect();
        let inter: Vec<*const u32> = sub[2].iter().map(|e| e.values.ptr as *const u32).collect();
        assert_eq!((main.len() as u32, inter.len() as u32), (n_main, 4 * n_logup));
        // accum.columns(): this component's powers are the LAST n_cons remaining ones, used reversed (constraint 0 <-> highest power)
        let [mut accum] = evaluation_accumulator.columns([(log_size + 1, n_cons as usize)]);
        let coeffs: Vec<u32> = accum.random_coeff_powers.iter().rev().flat_map(|c| qm31_words(*c)).collect();
        let (lookup, sum, acc) = (self.lookup_words(), qm31_words(self.claimed_sum()), coords_mut(accum.col));
        check(unsafe { sys::bfhip_eval_constraints(ctx().p(), E::COMPONENT, log_size, is_first, main.as_ptr(), std::ptr::null(), inter.as_ptr(), std::ptr::null(), lookup.as_ptr(),
                                                   sum.as_ptr(), coeffs.as_ptr(), acc.as_ptr()) });
    }
}

/// `interaction_trace_evaluation` of one component (`memory/table.rs:485-518` and the 12 analogues) without `LogupTraceGenerator`:
/// the main-trace columns of the component (row-granular: one value per table row) in, the 4 * n_logup interaction columns and the
/// claimed sum out.
pub fn interaction_trace_evaluation(component: i32, log_size: u32, main_rows: &[&HipColumn<BaseField>], lookup: &[u32; 24]) -> (Vec<HipColumn<BaseField>>, SecureField) {
    let (mut n_main, mut n_logup, mut n_cons) = (0u32, 0u32, 0u32);
    check(unsafe { sys::bfhip_component_shape(component, &mut n_main, &mut n_logup, &mut n_cons) });
    let rows = 1usize << (log_size - 4);
    // all but the last logUp column are 16x replicated and come back row-granular; the last one is full size
    let outs: Vec<HipColumn<BaseField>> = (0..4 * n_logup).map(|k| HipColumn::alloc(if k + 4 >= 4 * n_logup { rows * 16 } else { rows })).collect();
    let (src, dst): (Vec<*const u32>, Vec<*mut u32>) = (main_rows.iter().map(|c| c.ptr as *const u32).collect(), outs.iter().map(|c| c.ptr).collect());
    let mut sum = [0u32; 4];
    check(unsafe { sys::bfhip_logup_generate(ctx().p(), component, log_size, src.as_ptr(), lookup.as_ptr(), dst.as_ptr(), sum.as_mut_ptr()) });
    (outs, SecureField::from_m31_array(sum.map(BaseField::from_u32_unchecked)))
}
