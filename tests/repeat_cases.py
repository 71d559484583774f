"""The reference's own datasets for this path: tandem-repeat genomes (hmmv2/tests/dbg.rs:64-79 over
genome.rs:294-340) and the sim.sh-shaped case (scripts/sim.sh:184-214 -> bin/draft.rs:94-110), restated with our own
PRNG (the property, not the data: the reference's rand / xoshiro streams cannot be regenerated here).

Datasets as in hmmv2/tests/dbg.rs:51-63: 20x, 1000-state fragments drawn from the genome's own PHMM,
PHMMParams::uniform(0.001); DBG = the genome's k-mers with their true copy numbers, forward strand.
"""
import numpy as np

import dbgphmm_amd as D

# name -> (arguments of tandem_repeat_polyploid_with_unique_homo_ends, in the reference's order)
GENOMES = {
    "u1k": (1000, 1, 0, 0.0, 0, 50, 2, 0.01, 0),        # dbg.rs:65-67 unit 1 kb x 1: unique sequence
    "u20": (20, 50, 0, 0.0, 0, 50, 2, 0.01, 0),         # dbg.rs:69-71 unit 20 bp x 50
    "u20n200": (20, 200, 0, 0.02, 0, 300, 2, 0.02, 0),  # dbg.rs:73-75 unit 20 bp x 200, 2 % inside the repeat
    "u100": (100, 10, 0, 0.0, 0, 50, 2, 0.01, 0),       # dbg.rs:77-79 unit 100 bp x 10
    # scripts/sim.sh:187 (run_n4): -U 10000 -N 4 -E 2000 -H 0.01 --H0 0.0002 -P 2 (genome seed 0 -> div_init_seed 1)
    "sim_n4": (10000, 4, 0, 0.0002, 1, 2000, 2, 0.01, 0),
    # a longer u100: 100 bp unit x 100 (10 kb of repeat) -- long reads whose frontier leaves the 64-lane class again and
    # again (slices and bursts of both frontier passes)
    "u100n100": (100, 100, 0, 0.0, 0, 50, 2, 0.01, 0),
    # scripts/sim.sh:218 (run_n10): -U 2000 -N 10 -E 2000
    "sim_n10": (2000, 10, 0, 0.0002, 1, 2000, 2, 0.01, 0),
}


def dataset(name, k, coverage=20, read_len=1000, p=0.001, read_seed=0, min_copy_num=1, max_reads=None):
    """-> (PHMMArrays of the k-mer graph, reads, SeqGraph, haplotypes)"""
    haps = D.tandem_repeat_polyploid_with_unique_homo_ends(*GENOMES[name])
    sg = D.dbg_from_haplotypes(haps, k)
    param = D.PHMMParams.uniform(p).with_(n_warmup=k)
    arrays = D.vectorised_to_phmm(sg, param, min_copy_num)
    reads = D.sample_genome_reads(haps, param, coverage, read_len, read_seed, max_reads)
    return arrays, reads, sg, haps
