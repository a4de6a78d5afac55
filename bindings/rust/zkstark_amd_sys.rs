//! Raw bindings to libzkstark_amd.so (include/zkstark_amd.h).  Not compiled in the build image.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] pub struct zk_batch { _private: [u8; 0] }
#[repr(C)] pub struct zk_ctx { _private: [u8; 0] }
#[repr(C)] pub struct zk_dom { _private: [u8; 0] }
#[repr(C)] pub struct zk_channel { _private: [u8; 0] }

pub const ZK_OK: c_int = 0;
pub const ZK_FIELD_P: u32 = 3221225473;

#[repr(C)]
pub struct zk_transcript_info {
    pub alpha_raw: [u32; 3],
    pub beta_raw: [u32; 32],
    pub free_term: u32,
    pub query_raw: u32,
    pub public_last: u32,
    pub roots: [[u8; 32]; 34],
}

#[link(name = "zkstark_amd")]
extern "C" {
    pub fn zk_last_error() -> *const c_char;
    pub fn zk_version() -> *const c_char;
    // field.rs
    pub fn zk_field_add(a: u32, b: u32) -> u32;
    pub fn zk_field_sub(a: u32, b: u32) -> u32;
    pub fn zk_field_mul(a: u32, b: u32) -> u32;
    pub fn zk_field_neg(a: u32) -> u32;
    pub fn zk_field_inv(a: u32) -> u32;
    pub fn zk_field_pow(a: u32, e: u32) -> u32;
    pub fn zk_field_from_u32(v: u32) -> u32;
    pub fn zk_field_generator() -> u32;
    pub fn zk_field_root_of_unity(log_order: u32) -> u32;
    // context + stages (prover.rs:60-225)
    pub fn zk_ctx_create(device: c_int, log_n: u32, log_blowup: u32, out: *mut *mut zk_ctx) -> c_int;
    pub fn zk_ctx_destroy(ctx: *mut zk_ctx) -> c_int;
    pub fn zk_trace_fibsq(a0: u32, a1: u32, count: usize, out: *mut u32) -> c_int;
    pub fn zk_trace_upload(ctx: *mut zk_ctx, trace: *const u32, count: usize) -> c_int;
    pub fn zk_lde(ctx: *mut zk_ctx) -> c_int;
    pub fn zk_merkle_commit(ctx: *mut zk_ctx, layer: u32, root_out: *mut u8) -> c_int;
    pub fn zk_compose(ctx: *mut zk_ctx, alpha_raw: *const u32) -> c_int;
    pub fn zk_fri_fold(ctx: *mut zk_ctx, round: u32, beta_raw: u32) -> c_int;
    pub fn zk_layer_read(ctx: *mut zk_ctx, layer: u32, offset: usize, count: usize, out: *mut u32) -> c_int;
    pub fn zk_merkle_node(ctx: *mut zk_ctx, tree: u32, index: usize, out: *mut u8) -> c_int;
    pub fn zk_merkle_path(ctx: *mut zk_ctx, tree: u32, leaf: usize, out: *mut u8, path_len: *mut usize) -> c_int;
    // generate_proof in one call (prover.rs:9)
    pub fn zk_prove(ctx: *mut zk_ctx, trace: *const u32, count: usize, proof_out: *mut u8, cap: usize,
                    proof_len: *mut usize, state_out: *mut u8) -> c_int;
    pub fn zk_prove_resident(ctx: *mut zk_ctx, proof_out: *mut u8, cap: usize, proof_len: *mut usize,
                             state_out: *mut u8) -> c_int;
    pub fn zk_last_transcript(ctx: *const zk_ctx, out: *mut zk_transcript_info) -> c_int;
    // proof.rs
    pub fn zk_verify(proof: *const u8, len: usize, log_n: u32, log_blowup: u32, public_last: u32) -> c_int;
    pub fn zk_verify_strict(proof: *const u8, len: usize, state: *const u8, log_n: u32, log_blowup: u32,
                            public_last: u32) -> c_int;
    pub fn zk_proof_size(data_len: usize) -> usize;
    pub fn zk_proof_data_len(log_n: u32, log_blowup: u32) -> usize;
    pub fn zk_compute_root_from_path(element: u32, index: usize, path: *const u8, path_len: usize, out: *mut u8) -> c_int;
    // stand-alone Merkle::new (merkle.rs:14)
    pub fn zk_merkle_build_host(device: c_int, vals: *const u32, m: usize, nodes_out: *mut u8) -> c_int;
    // device-pointer primitives
    pub fn zk_dev_merkle_build(d_vals: *const u32, log_m: u32, d_nodes: *mut u32, stream: *mut c_void) -> c_int;
    pub fn zk_prove_many(ctxs: *const *mut zk_ctx, count: usize, proofs_out: *mut u8, stride: usize, lens_out: *mut usize,
                         states_out: *mut u8) -> c_int;
    // settings of the one-call prover
    pub fn zk_ctx_set_queries(ctx: *mut zk_ctx, n_queries: u32) -> c_int;
    pub fn zk_ctx_set_hash(ctx: *mut zk_ctx, hash_kind: c_int) -> c_int;
    pub fn zk_ctx_set_host_levels(ctx: *mut zk_ctx, top_log: u32, tail_log: u32) -> c_int;
    // batched proving: 2^log_batch proofs of one size in lockstep (prover.rs:9-293 each)
    pub fn zk_batch_create(device: c_int, log_n: u32, log_blowup: u32, log_batch: u32, out: *mut *mut zk_batch) -> c_int;
    pub fn zk_batch_destroy(b: *mut zk_batch) -> c_int;
    pub fn zk_batch_size(b: *const zk_batch) -> usize;
    pub fn zk_batch_set_queries(b: *mut zk_batch, n_queries: u32) -> c_int;
    pub fn zk_batch_set_hash(b: *mut zk_batch, hash_kind: c_int) -> c_int;
    pub fn zk_batch_set_traces(b: *mut zk_batch, traces: *const u32) -> c_int;
    pub fn zk_batch_gen_fibsq(b: *mut zk_batch, a0: *const u32, a1: *const u32) -> c_int;
    pub fn zk_batch_public_last(b: *const zk_batch, out: *mut u32) -> c_int;
    pub fn zk_batch_prove(b: *mut zk_batch, proofs_out: *mut u8, stride: usize, states_out: *mut u8) -> c_int;
}
