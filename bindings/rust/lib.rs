//! Safe Rust wrapper over `libbfhip.so` — the whole-path entry points of the MI355X backend (source only: the build image has no
//! Rust toolchain). `bfhip_sys.rs` is generated from `include/bfhip.h` by `tools/gen_rust_ffi.py`.
//!
//! In `crates/brainfuck_prover` this replaces the body of `prove_brainfuck` (`brainfuck_air/mod.rs:471`):
//! ```ignore
//! let ctx = bfhip::Context::new(0, LOG_MAX_ROWS + 2)?;
//! let json = ctx.prove_brainfuck(&code, &input, LOG_MAX_ROWS)?;
//! let proof: BrainfuckProof<Blake2sMerkleHasher> = serde_json::from_slice(&json)?;   // same serde shape (mod.rs:71-99)
//! ```
#[path = "bfhip_sys.rs"]
pub mod sys;
/// stwo's backend trait surface (`Backend`, `ColumnOps`, `FieldOps`, `PolyOps`, `MerkleOps`, `QuotientOps`, `FriOps`, `AccumulationOps`, `GrindOps`,
/// `ComponentProver`) over the FFI — for `prover::prove::<HipBackend, _>` at `mod.rs:732`.
/// Never compiled (no toolchain, stwo not vendored): kept out of the default module tree behind the feature `stwo-backend`.
#[cfg(feature = "stwo-backend")]
#[path = "hip_backend.rs"]
pub mod hip_backend;

use std::ffi::{c_char, c_void, CStr, CString};

fn last_error() -> String {
    unsafe { CStr::from_ptr(sys::bfhip_last_error()).to_string_lossy().into_owned() }
}

/// One GPU + one HIP stream + the twiddle tree (`mod.rs:480-487`). Calls on a context are serial.
pub struct Context(*mut sys::BfhipCtx);

// A context may be moved to another thread; every entry point binds the calling thread to the context's GPU.
unsafe impl Send for Context {}

impl Context {
    pub fn new(device_id: i32, max_log_domain: u32) -> Result<Self, String> {
        let mut p = std::ptr::null_mut();
        if unsafe { sys::bfhip_ctx_create(device_id, max_log_domain, &mut p) } != 0 {
            return Err(last_error());
        }
        Ok(Context(p))
    }

    /// `prove_brainfuck` (`mod.rs:471-735`): compile, run, build the tables, commit, prove. Returns the serde-JSON proof bytes.
    pub fn prove_brainfuck(&self, code: &str, input: &[u8], log_max_rows: u32) -> Result<Vec<u8>, String> {
        let code = CString::new(code).map_err(|e| e.to_string())?;
        let (mut js, mut len): (*mut c_char, usize) = (std::ptr::null_mut(), 0);
        let rc = unsafe {
            sys::bfhip_prove_brainfuck(self.0, code.as_ptr(), input.as_ptr(), input.len(), log_max_rows, &mut js, &mut len,
                                       std::ptr::null_mut(), std::ptr::null_mut())
        };
        if rc != 0 {
            return Err(last_error());   // "ConstraintsNotSatisfied", "a component exceeds LOG_MAX_ROWS", HIP errors, ...
        }
        let out = unsafe { std::slice::from_raw_parts(js as *const u8, len) }.to_vec();
        unsafe { sys::bfhip_free_host(js as *mut c_void) };
        Ok(out)
    }

    /// `prove_brainfuck(&Machine)` as the reference receives it (`mod.rs:471-473`): the executed machine's register trace
    /// (`inputs.trace()`, `mod.rs:508`; 7 words per row: clk, ip, ci, ni, mp, mv, mvi) and its program words — no re-execution.
    pub fn prove_machine(&self, trace7: &[u32], program: &[u32], log_max_rows: u32) -> Result<Vec<u8>, String> {
        assert!(trace7.len() % 7 == 0);
        let mut tr: *mut sys::BfhipTrace = std::ptr::null_mut();
        let rc = unsafe {
            sys::bfhip_trace_create_from_registers(self.0, trace7.as_ptr(), trace7.len() / 7, program.as_ptr(), program.len(), &mut tr, std::ptr::null_mut(),
                                                   std::ptr::null_mut(), std::ptr::null_mut())
        };
        if rc != 0 {
            return Err(last_error());
        }
        let (mut js, mut len): (*mut c_char, usize) = (std::ptr::null_mut(), 0);
        let rc = unsafe { sys::bfhip_prove_trace(self.0, tr, log_max_rows, &mut js, &mut len, std::ptr::null_mut(), std::ptr::null_mut()) };
        unsafe { sys::bfhip_trace_destroy(self.0, tr) };
        if rc != 0 {
            return Err(last_error());
        }
        let out = unsafe { std::slice::from_raw_parts(js as *const u8, len) }.to_vec();
        unsafe { sys::bfhip_free_host(js as *mut c_void) };
        Ok(out)
    }

    /// Byte-level stwo conventions / Merkle channel of this context (`bfhip_conventions`; all zero = defaults).
    pub fn set_conventions(&self, conv: &sys::BfhipConventions) -> Result<(), String> {
        if unsafe { sys::bfhip_ctx_set_conventions(self.0, conv) } != 0 { Err(last_error()) } else { Ok(()) }
    }

    /// Keep the program-independent preprocessed tree across proofs (the reference recommits it in every call).
    pub fn reuse_preprocessed(&self, on: bool) {
        unsafe { sys::bfhip_ctx_reuse_preprocessed(self.0, on as i32) };
    }

    /// Device memory of this context in bytes: [reserved by the per-proof arena, its peak use, the twiddle trees, in use now] (`bfhip_ctx_memory`).
    pub fn memory(&self) -> Result<[u64; 4], String> {
        let mut out = [0u64; 4];
        if unsafe { sys::bfhip_ctx_memory(self.0, out.as_mut_ptr()) } != 0 { Err(last_error()) } else { Ok(out) }
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { sys::bfhip_ctx_destroy(self.0) };
    }
}

/// `verify_brainfuck` (`mod.rs:738-797`), host only. `Err` carries the rejection reason.
pub fn verify_brainfuck(proof_json: &[u8], log_max_rows: u32) -> Result<(), String> {
    let mut err = vec![0u8; 256];
    let rc = unsafe {
        sys::bfhip_verify_brainfuck(proof_json.as_ptr() as *const c_char, proof_json.len(), log_max_rows, err.as_mut_ptr() as *mut c_char, err.len())
    };
    match rc {
        0 => Ok(()),
        1 => Err(unsafe { CStr::from_ptr(err.as_ptr() as *const c_char) }.to_string_lossy().into_owned()),   // rejected: VerificationError name
        _ => Err(last_error()),                                                                                // internal error
    }
}
