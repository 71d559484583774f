"""Shared synthetic-input builders for the tests (seeded, small)."""
import numpy as np

from dbgphmm_amd import PHMMParams, dbg_from_haplotypes, diverge, random_genome, sample_reads, vectorised_to_phmm


def small_dbg_model(genome_len=300, k=12, p=0.01, seed=7, diploid=True, min_copy_num=0):
    hap_a = random_genome(genome_len, seed)
    haps = [hap_a, diverge(hap_a, 0.02, seed + 1)] if diploid else [hap_a]
    sg = dbg_from_haplotypes(haps, k)
    param = PHMMParams.uniform(p).with_(n_warmup=k)
    return vectorised_to_phmm(sg, param, min_copy_num), sg


def finite_close(a, b, atol, floor=None):
    """compare log arrays: both -inf, or |a-b| <= atol; entries where the reference value is
    below `floor` may be -inf on the GPU (underflow of the scaled linear domain)."""
    a, b = np.asarray(a, float), np.asarray(b, float)
    with np.errstate(invalid="ignore"):
        return _finite_close(a, b, atol, floor)


def _finite_close(a, b, atol, floor):
    both_inf = np.isneginf(a) & np.isneginf(b)
    ok = both_inf | (np.abs(a - b) <= atol)
    if floor is not None:
        ok |= (b < floor) & (np.isneginf(a) | (np.abs(a - b) <= 1e-3))
    return ok
