"""dbgphmm_amd: MI355X-native profile-HMM read-likelihood path of dbgphmm (src/hmmv2).

Compute lives in libphmm_amd.so (hand-written HIP for gfx950 behind the C ABI of
include/phmm_amd.h).  This package is the thin host mirror of the reference's
`PHMMParams` / `PHMMModel` surface plus the graph->PHMM parameter builders.
"""
from .params import PHMMParams, MAX_ACTIVE_NODES  # noqa: F401
from .graph import (PHMMArrays, SeqGraph, mock_linear, mock_crossing, toy_repeat,  # noqa: F401
                    dbg_from_haplotypes, random_genome, diverge, sample_reads, vectorised_to_phmm,
                    mutate_exact, tandem_repeat_polyploid_with_unique_homo_ends, genome_phmm,
                    sample_genome_reads, kp1_node_map)
from .model import PHMMModel, PHMMOutput, ReadCollection, Mappings, DenseTables  # noqa: F401
from ._ffi import PhmmError, build  # noqa: F401

from . import formats  # noqa: F401,E402  (DBG / MAP / FASTA files either side of the path)
