//! MI355X backend of the profile-HMM read-likelihood path: the `impl PHMMModel` methods of
//! `hmmv2/freq.rs` and `hmmv2/hint.rs` re-implemented over `libphmm_amd.so` (amd_sys.rs).
//!
//! Place as `src/hmmv2/amd.rs`, add `pub mod amd; pub mod amd_sys;` to `src/hmmv2.rs` behind a cargo
//! feature (`amd`), and gate the CPU bodies of the same-named methods with `#[cfg(not(feature = "amd"))]`:
//! `MultiDbg::to_likelihood` (multi_dbg/posterior.rs:247-255) and `MultiDbg::generate_mappings`
//! (multi_dbg/posterior.rs:609-630) then call into the GPU unchanged.  Error behaviour is the
//! reference's: it has no `Result` on this path, a failing call panics with the library's message.
//!
//! Handles: `AmdModel` (topology + probabilities of one PHMM), `AmdReads`, `AmdMappings` own their
//! device arrays and free them on drop.  The DP workspaces belong to the device and are shared by all
//! handles (`phmm_release_workspace`), so a mapping model and a scoring model can be alive together.
use super::amd_sys::*;
use super::common::{PHMMEdge, PHMMModel, PHMMNode};
use super::hint::{Mapping, Mappings};
use super::params::PHMMParams;
use crate::common::collection::ReadCollection;
use crate::common::Seq;
use crate::prob::Prob;
use petgraph::graph::NodeIndex;
use std::ffi::CStr;
use std::os::raw::c_int;
use std::ptr;

/// non-zero status -> panic!, preserving the reference's error behaviour
fn check(rc: c_int) {
    if rc != PHMM_OK {
        let msg = unsafe { CStr::from_ptr(phmm_last_error()) }.to_string_lossy().into_owned();
        panic!("phmm_amd error {}: {}", rc, msg);
    }
}

fn to_params_c(p: &PHMMParams) -> phmm_params {
    phmm_params {
        p_mismatch: p.p_mismatch.to_log_value(),
        p_match: p.p_match.to_log_value(),
        p_random: p.p_random.to_log_value(),
        p_gap_open: p.p_gap_open.to_log_value(),
        p_gap_ext: p.p_gap_ext.to_log_value(),
        p_end: p.p_end.to_log_value(),
        p_mm: p.p_MM.to_log_value(),
        p_im: p.p_IM.to_log_value(),
        p_dm: p.p_DM.to_log_value(),
        p_mi: p.p_MI.to_log_value(),
        p_ii: p.p_II.to_log_value(),
        p_di: p.p_DI.to_log_value(),
        p_md: p.p_MD.to_log_value(),
        p_id: p.p_ID.to_log_value(),
        p_dd: p.p_DD.to_log_value(),
        n_active_nodes: p.n_active_nodes as i64,
        active_node_max_ratio: p.active_node_max_ratio,
        n_warmup: p.n_warmup as i64,
        warmup_threshold: p.warmup_threshold as i64,
        n_max_gaps: p.n_max_gaps as i64,
    }
}

pub struct AmdModel(*mut phmm_model);
impl Drop for AmdModel {
    fn drop(&mut self) {
        unsafe { phmm_model_destroy(self.0) }
    }
}
impl AmdModel {
    /// Flatten the petgraph into the arrays the ABI takes: node id = NodeIndex, edge order = EdgeIndex
    /// order = petgraph insertion order (the library rebuilds petgraph's newest-edge-first adjacency
    /// order from it), Prob -> to_log_value().
    pub fn new<N: PHMMNode, E: PHMMEdge>(phmm: &PHMMModel<N, E>) -> AmdModel {
        let emission: Vec<u8> = phmm.nodes().map(|(_, w)| w.emission()).collect();
        let init: Vec<f64> = phmm.nodes().map(|(_, w)| w.init_prob().to_log_value()).collect();
        let (mut src, mut dst, mut tr) = (Vec::new(), Vec::new(), Vec::new());
        for (_, s, t, w) in phmm.edges() {
            src.push(s.index() as u32);
            dst.push(t.index() as u32);
            tr.push(w.trans_prob().to_log_value());
        }
        let p = to_params_c(&phmm.param);
        let mut h = ptr::null_mut();
        check(unsafe {
            phmm_model_create(emission.len() as u32, src.len() as u32, emission.as_ptr(), init.as_ptr(), src.as_ptr(),
                              dst.as_ptr(), tr.as_ptr(), &p, &mut h)
        });
        AmdModel(h)
    }
    /// next candidate on the same topology (what `dbg.clone(); set_copy_nums; to_phmm` rebuilds, posterior.rs:483-501)
    pub fn set_probs<N: PHMMNode, E: PHMMEdge>(&mut self, phmm: &PHMMModel<N, E>) {
        let init: Vec<f64> = phmm.nodes().map(|(_, w)| w.init_prob().to_log_value()).collect();
        let tr: Vec<f64> = phmm.edges().map(|(_, _, _, w)| w.trans_prob().to_log_value()).collect();
        check(unsafe { phmm_model_set_probs(self.0, init.as_ptr(), tr.as_ptr()) });
    }
}

pub struct AmdReads {
    h: *mut phmm_reads,
    offsets: Vec<u64>,
}
impl Drop for AmdReads {
    fn drop(&mut self) {
        unsafe { phmm_reads_destroy(self.h) }
    }
}
impl AmdReads {
    /// concatenated bases + offsets (ReadCollection<S>, common/collection.rs:38-83)
    pub fn new<S: Seq>(reads: &ReadCollection<S>) -> AmdReads {
        let mut bases: Vec<u8> = Vec::new();
        let mut offsets: Vec<u64> = vec![0];
        for r in reads.iter() {
            bases.extend_from_slice(r.as_ref());
            offsets.push(bases.len() as u64);
        }
        let mut h = ptr::null_mut();
        check(unsafe { phmm_reads_create(bases.as_ptr(), offsets.as_ptr(), (offsets.len() - 1) as u64, &mut h) });
        AmdReads { h, offsets }
    }
    pub fn n_reads(&self) -> usize {
        self.offsets.len() - 1
    }
}

pub struct AmdMappings(*mut phmm_mappings);
impl Drop for AmdMappings {
    fn drop(&mut self) {
        unsafe { phmm_mappings_destroy(self.0) }
    }
}
impl AmdMappings {
    /// Mappings (hint.rs:27-30, 149) -> the 3-level CSR of the ABI
    pub fn from_mappings(reads: &AmdReads, mappings: &Mappings) -> AmdMappings {
        let (mut pos_off, mut nodes, mut logp) = (vec![0u64], Vec::<u32>::new(), Vec::<f64>::new());
        for i in 0..mappings.n_reads() {
            let m: &Mapping = &mappings[i];
            for j in 0..m.len() {
                for (&v, &p) in m.nodes[j].iter().zip(m.probs[j].iter()) {  // (pub fields, hint.rs:27-30)
                    nodes.push(v.index() as u32);
                    logp.push(p.to_log_value());
                }
                pos_off.push(nodes.len() as u64);
            }
        }
        let mut h = ptr::null_mut();
        check(unsafe { phmm_mappings_create(reads.h, pos_off.as_ptr(), nodes.as_ptr(), logp.as_ptr(), &mut h) });
        AmdMappings(h)
    }
    /// the ABI's CSR -> Mappings (phmm_mappings_export), one Mapping per read
    pub fn into_mappings(self, reads: &AmdReads) -> Mappings {
        let (tp, te) = unsafe { (phmm_mappings_total_positions(self.0), phmm_mappings_total_entries(self.0)) };
        let mut pos_off = vec![0u64; tp as usize + 1];
        let mut nodes = vec![0u32; te as usize];
        let mut logp = vec![0f64; te as usize];
        check(unsafe { phmm_mappings_export(self.0, pos_off.as_mut_ptr(), nodes.as_mut_ptr(), logp.as_mut_ptr()) });
        let mut out = Vec::with_capacity(reads.n_reads());
        for r in 0..reads.n_reads() {
            let vs: Vec<Vec<(NodeIndex, Prob)>> = (reads.offsets[r]..reads.offsets[r + 1])
                .map(|g| {
                    (pos_off[g as usize]..pos_off[g as usize + 1])
                        .map(|a| (NodeIndex::new(nodes[a as usize] as usize), Prob::from_log_prob(logp[a as usize])))
                        .collect()
                })
                .collect();
            out.push(Mapping::from_nodes_and_probs(&vs));
        }
        Mappings::new(out)
    }
}

impl<N: PHMMNode, E: PHMMEdge> PHMMModel<N, E> {
    /// drop-in for `to_full_prob_reads` (freq.rs:175-192): forward_with_mapping_score_only per read when
    /// mappings are given, else forward_sparse_score_only(use_max_ratio); the product over reads.
    pub fn to_full_prob_reads_amd<S: Seq>(&self, reads: &ReadCollection<S>, mappings: Option<&Mappings>,
                                          use_max_ratio: bool) -> Prob {
        let m = AmdModel::new(self);
        let r = AmdReads::new(reads);
        let mp = mappings.map(|mp| AmdMappings::from_mappings(&r, mp));
        let mut total = 0f64;
        check(unsafe {
            phmm_full_prob_reads(m.0, r.h, mp.as_ref().map_or(ptr::null(), |x| x.0 as *const _), use_max_ratio as c_int,
                                 ptr::null_mut(), &mut total)
        });
        Prob::from_log_prob(total)
    }

    /// drop-in for `generate_mappings` (hint.rs:193-220): run_with_mapping when `mappings` is given (hint.rs:206-208),
    /// else run_sparse_adaptive(use_max_ratio); then to_mapping_by_score_ratio / to_mapping.
    pub fn generate_mappings_amd<S: Seq>(&self, reads: &ReadCollection<S>, mappings: Option<&Mappings>,
                                         use_max_ratio: bool) -> Mappings {
        let m = AmdModel::new(self);
        let r = AmdReads::new(reads);
        let mp = mappings.map(|mp| AmdMappings::from_mappings(&r, mp));
        let mut out = ptr::null_mut();
        check(unsafe {
            phmm_generate_mappings(m.0, r.h, mp.as_ref().map_or(ptr::null(), |x| x.0 as *const _), use_max_ratio as c_int,
                                   &mut out, ptr::null_mut())
        });
        AmdMappings(out).into_mappings(&r)
    }

    /// `generate_mappings` on a read handle the caller keeps (`AmdReads::new(reads)` once per read set).  `infer` passes the
    /// SAME reads at every k (multi_dbg/posterior.rs:698-826): the handle remembers how many dense warm-up columns
    /// each read needed in its last adaptive call, the next call groups reads by that (results do not depend on the
    /// grouping), and only the very first call of a read set runs without hints (bench.py: cold_hint_ms vs ms_per_step).
    pub fn generate_mappings_amd_on(&self, reads: &AmdReads, mappings: Option<&Mappings>, use_max_ratio: bool) -> Mappings {
        let m = AmdModel::new(self);
        let mp = mappings.map(|mp| AmdMappings::from_mappings(reads, mp));
        let mut out = ptr::null_mut();
        check(unsafe {
            phmm_generate_mappings(m.0, reads.h, mp.as_ref().map_or(ptr::null(), |x| x.0 as *const _), use_max_ratio as c_int,
                                   &mut out, ptr::null_mut())
        });
        AmdMappings(out).into_mappings(reads)
    }

    /// drop-in for `to_full_prob_sparse_backward` (freq.rs:153-163; backward_sparse per read, backward.rs:146-185)
    pub fn to_full_prob_sparse_backward_amd<S: Seq>(&self, reads: &ReadCollection<S>) -> Prob {
        let (m, r) = (AmdModel::new(self), AmdReads::new(reads));
        let mut total = 0f64;
        check(unsafe { phmm_full_prob_sparse_backward(m.0, r.h, ptr::null_mut(), &mut total) });
        Prob::from_log_prob(total)
    }

    /// `run` over a read set + to_full_prob_forward + to_node_freqs summed (freq.rs:89-119, 245-255) -> (ln P per read, node usage)
    pub fn run_dense_amd<S: Seq>(&self, reads: &ReadCollection<S>) -> (Vec<Prob>, Vec<f64>) {
        let (m, r) = (AmdModel::new(self), AmdReads::new(reads));
        let mut lf = vec![0f64; r.n_reads()];
        let mut nf = vec![0f64; self.n_nodes()];
        check(unsafe { phmm_run_dense(m.0, r.h, lf.as_mut_ptr(), ptr::null_mut(), nf.as_mut_ptr()) });
        (lf.into_iter().map(Prob::from_log_prob).collect(), nf)
    }

    /// run_sparse (freq.rs:51-55) over a read set + to_node_freqs (freq.rs:245-255), summed over the reads (dense Vec:
    /// the reference's NodeFreqs is a SparseVec alias, freq.rs:200)
    pub fn to_node_freqs_sparse_amd<S: Seq>(&self, reads: &ReadCollection<S>) -> Vec<f64> {
        let (m, r) = (AmdModel::new(self), AmdReads::new(reads));
        let mut nf = vec![0f64; self.n_nodes()];
        check(unsafe { phmm_run_sparse(m.0, r.h, ptr::null_mut(), ptr::null_mut(), nf.as_mut_ptr()) });
        nf
    }

    /// to_edge_and_init_freqs summed over the reads (freq.rs:276-298) and q_score_exact (q.rs:66-96) -> (init, trans, prior)
    pub fn q_score_amd<S: Seq>(&self, reads: &ReadCollection<S>) -> (f64, f64, f64) {
        let (m, r) = (AmdModel::new(self), AmdReads::new(reads));
        let (mut ef, mut nf, mut q) = (vec![0f64; self.n_edges().max(1)], vec![0f64; self.n_nodes()], [0f64; 3]);
        check(unsafe { phmm_run_dense_edges(m.0, r.h, ptr::null_mut(), ef.as_mut_ptr(), nf.as_mut_ptr()) });
        check(unsafe { phmm_q_score_exact(m.0, ef.as_ptr(), nf.as_ptr(), q.as_mut_ptr()) });
        (q[0], q[1], q[2])
    }
}

/// The cut that pays (multi_dbg/posterior.rs:483-515): ONE call per sampler iteration for all candidate
/// copy-number vectors instead of `dbg.clone(); set_copy_nums; to_phmm; to_full_prob_reads` per candidate inside a
/// rayon par_iter.  Topology, reads and mappings stay on the device for the whole k; a candidate is a row of
/// copy numbers (PHMM node id = full-edge id, multi_dbg.rs:1569-1576) and init / trans are derived on the device as
/// SeqGraph::to_phmm does (seq_graph.rs:110-135, 160-209).
pub struct AmdLikelihood {
    model: AmdModel,
    reads: AmdReads,
    mappings: AmdMappings,
    n_nodes: usize,
}
impl AmdLikelihood {
    pub fn new<N: PHMMNode, E: PHMMEdge, S: Seq>(phmm: &PHMMModel<N, E>, reads: &ReadCollection<S>,
                                                 mappings: &Mappings) -> AmdLikelihood {
        let model = AmdModel::new(phmm);
        let r = AmdReads::new(reads);
        let mp = AmdMappings::from_mappings(&r, mappings);
        AmdLikelihood { model, reads: r, mappings: mp, n_nodes: phmm.n_nodes() }
    }
    /// copy_nums: candidates x n_nodes, row-major -> ln P(R | X_c) per candidate (Score.likelihood, posterior.rs:259-277)
    pub fn likelihoods(&mut self, copy_nums: &[u32], min_copy_num: u32) -> Vec<Prob> {
        assert!(copy_nums.len() % self.n_nodes == 0);
        let c = copy_nums.len() / self.n_nodes;
        let mut totals = vec![0f64; c];
        check(unsafe {
            phmm_full_prob_reads_copy_nums(self.model.0, self.reads.h, self.mappings.0, c as u32, copy_nums.as_ptr(),
                                           min_copy_num, ptr::null_mut(), totals.as_mut_ptr())
        });
        totals.into_iter().map(Prob::from_log_prob).collect()
    }
}
