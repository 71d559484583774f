// placeholder, replaced below in this round
#include "phmm_internal.h"
extern "C" {
int phmm_mappings_node_freqs(const phmm_mappings *, uint32_t, double *) { return phmm::fail(PHMM_EINTERNAL, "not built yet"); }
int phmm_full_prob_reads(phmm_model *, const phmm_reads *, const phmm_mappings *, int, double *, double *) { return phmm::fail(PHMM_EINTERNAL, "not built yet"); }
int phmm_full_prob_reads_candidates(phmm_model *, const phmm_reads *, const phmm_mappings *, uint32_t, const double *, const double *, double *, double *) { return phmm::fail(PHMM_EINTERNAL, "not built yet"); }
int phmm_generate_mappings(phmm_model *, const phmm_reads *, const phmm_mappings *, int, phmm_mappings **, double *) { return phmm::fail(PHMM_EINTERNAL, "not built yet"); }
}
