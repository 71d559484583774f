//! Raw bindings of `include/phmm_amd.h` (libphmm_amd.so: hand-written HIP for gfx950 behind a C ABI).
//!
//! Drop this file in as `src/hmmv2/amd_sys.rs` of dbgphmm; `amd.rs` next to it is the safe layer that
//! re-implements the `impl PHMMModel` methods of `hmmv2/{freq,hint}.rs` on top of it, and `build.rs`
//! shows the two link lines.  (Written against the reference's types; this repository's image has no
//! cargo / rustc, so it is shipped as source and has not been compiled here.)
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const PHMM_OK: c_int = 0;
pub const PHMM_EINVAL: c_int = -1;
pub const PHMM_ENODEVICE: c_int = -2;
pub const PHMM_ENOMEM: c_int = -3;
pub const PHMM_ECAPACITY: c_int = -4;
pub const PHMM_EINTERNAL: c_int = -5;

pub const PHMM_READ_DEFERRED: u32 = 1;
pub const PHMM_READ_WIDE_FRONTIER: u32 = 2;
pub const PHMM_READ_FORCED_SWITCH: u32 = 4;

/// `phmm_params` = PHMMParams (hmmv2/params.rs:16-66) with every `Prob` as its f64 ln-value.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct phmm_params {
    pub p_mismatch: f64,
    pub p_match: f64,
    pub p_random: f64,
    pub p_gap_open: f64,
    pub p_gap_ext: f64,
    pub p_end: f64,
    pub p_mm: f64,
    pub p_im: f64,
    pub p_dm: f64,
    pub p_mi: f64,
    pub p_ii: f64,
    pub p_di: f64,
    pub p_md: f64,
    pub p_id: f64,
    pub p_dd: f64,
    pub n_active_nodes: i64,
    pub active_node_max_ratio: f64,
    pub n_warmup: i64,
    pub warmup_threshold: i64,
    pub n_max_gaps: i64,
}

#[repr(C)]
pub struct phmm_model {
    _p: [u8; 0],
}
#[repr(C)]
pub struct phmm_reads {
    _p: [u8; 0],
}
#[repr(C)]
pub struct phmm_mappings {
    _p: [u8; 0],
}

#[link(name = "phmm_amd")]
extern "C" {
    pub fn phmm_last_error() -> *const c_char;
    pub fn phmm_version() -> *const c_char;
    pub fn phmm_device_count() -> c_int;
    pub fn phmm_set_device(device: c_int) -> c_int;
    pub fn phmm_set_stream(hip_stream: *mut c_void) -> c_int;
    pub fn phmm_set_workspace_limit(bytes: u64) -> c_int;
    pub fn phmm_release_workspace() -> c_int;
    pub fn phmm_workspace_bytes() -> u64;

    pub fn phmm_params_new(p_mismatch: f64, p_gap_open: f64, p_gap_ext: f64, p_end: f64, n_active_nodes: i64,
                           n_warmup: i64, out: *mut phmm_params) -> c_int;
    pub fn phmm_params_uniform(p: f64, out: *mut phmm_params) -> c_int;

    pub fn phmm_model_create(n_nodes: u32, n_edges: u32, emission: *const u8, init_logp: *const f64,
                             edge_src: *const u32, edge_dst: *const u32, trans_logp: *const f64,
                             params: *const phmm_params, out: *mut *mut phmm_model) -> c_int;
    pub fn phmm_model_set_probs(m: *mut phmm_model, init_logp: *const f64, trans_logp: *const f64) -> c_int;
    pub fn phmm_model_set_params(m: *mut phmm_model, params: *const phmm_params) -> c_int;
    pub fn phmm_model_n_nodes(m: *const phmm_model) -> u32;
    pub fn phmm_model_n_edges(m: *const phmm_model) -> u32;
    pub fn phmm_model_destroy(m: *mut phmm_model);

    pub fn phmm_reads_create(bases: *const u8, offsets: *const u64, n_reads: u64, out: *mut *mut phmm_reads) -> c_int;
    pub fn phmm_reads_count(r: *const phmm_reads) -> u64;
    pub fn phmm_reads_total_bases(r: *const phmm_reads) -> u64;
    pub fn phmm_reads_last_call_info(r: *const phmm_reads, out_dense_columns: *mut u16, out_flags: *mut u32) -> c_int;
    pub fn phmm_reads_destroy(r: *mut phmm_reads);

    pub fn phmm_run_dense(m: *mut phmm_model, reads: *const phmm_reads, out_logp_forward: *mut f64,
                          out_logp_backward: *mut f64, out_node_freq: *mut f64) -> c_int;
    pub fn phmm_run_dense_edges(m: *mut phmm_model, reads: *const phmm_reads, out_logp_forward: *mut f64,
                                out_edge_freq: *mut f64, out_init_freq: *mut f64) -> c_int;
    pub fn phmm_q_score_exact(m: *const phmm_model, edge_freq: *const f64, init_freq: *const f64, out_q: *mut f64) -> c_int;
    pub fn phmm_dense_tables(m: *mut phmm_model, read: *const u8, len: u64, f_m: *mut f64, f_i: *mut f64, f_d: *mut f64,
                             f_scal: *mut f64, b_m: *mut f64, b_i: *mut f64, b_d: *mut f64, b_scal: *mut f64) -> c_int;

    pub fn phmm_mappings_create(reads: *const phmm_reads, pos_off: *const u64, nodes: *const u32, logp: *const f64,
                                out: *mut *mut phmm_mappings) -> c_int;
    pub fn phmm_mappings_total_positions(mp: *const phmm_mappings) -> u64;
    pub fn phmm_mappings_total_entries(mp: *const phmm_mappings) -> u64;
    pub fn phmm_mappings_export(mp: *const phmm_mappings, pos_off: *mut u64, nodes: *mut u32, logp: *mut f64) -> c_int;
    pub fn phmm_mappings_node_freqs(mp: *const phmm_mappings, n_nodes: u32, out_freq: *mut f64) -> c_int;
    pub fn phmm_mappings_read_logp(mp: *const phmm_mappings, out_logp: *mut f64, out_total: *mut f64) -> c_int;
    pub fn phmm_mappings_destroy(mp: *mut phmm_mappings);

    pub fn phmm_full_prob_reads(m: *mut phmm_model, reads: *const phmm_reads, mappings: *const phmm_mappings,
                                use_max_ratio: c_int, out_logp: *mut f64, out_total: *mut f64) -> c_int;
    pub fn phmm_full_prob_reads_candidates(m: *mut phmm_model, reads: *const phmm_reads, mappings: *const phmm_mappings,
                                           n_candidates: u32, init_logp: *const f64, trans_logp: *const f64,
                                           out_logp: *mut f64, out_total: *mut f64) -> c_int;
    pub fn phmm_full_prob_reads_copy_nums(m: *mut phmm_model, reads: *const phmm_reads, mappings: *const phmm_mappings,
                                          n_candidates: u32, copy_nums: *const u32, min_copy_num: u32,
                                          out_logp: *mut f64, out_total: *mut f64) -> c_int;
    pub fn phmm_full_prob_sparse_backward(m: *mut phmm_model, reads: *const phmm_reads, out_logp: *mut f64,
                                          out_total: *mut f64) -> c_int;
    pub fn phmm_run_sparse(m: *mut phmm_model, reads: *const phmm_reads, out_logp_forward: *mut f64,
                           out_logp_backward: *mut f64, out_node_freq: *mut f64) -> c_int;
    pub fn phmm_backward_sparse_tables(m: *mut phmm_model, read: *const u8, len: u64, b_m: *mut f64, b_i: *mut f64,
                                       b_d: *mut f64, b_scal: *mut f64, is_dense: *mut u8) -> c_int;
    pub fn phmm_mappings_map_nodes(model_after: *mut phmm_model, reads: *const phmm_reads, mappings: *const phmm_mappings,
                                   map_off: *const u32, map_nodes: *const u32, n_nodes_before: u32,
                                   out: *mut *mut phmm_mappings) -> c_int;
    pub fn phmm_generate_mappings(m: *mut phmm_model, reads: *const phmm_reads, mappings: *const phmm_mappings,
                                  use_max_ratio: c_int, out: *mut *mut phmm_mappings, out_node_freq: *mut f64) -> c_int;

    pub fn phmm_last_call_stats(which: c_int, out_ms: *mut f64, out_launches: *mut u64, out_cells: *mut u64) -> c_int;
    pub fn phmm_enable_timing(on: c_int) -> c_int;
}
