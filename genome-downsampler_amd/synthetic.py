"""Synthetic inputs for the BASELINE.json configs that libs/reads-gen does not produce directly
(SURVEY.md section 8d): the amplicon panel of configs[2] and the GRCh38-shaped contig set of
configs[4].  Deterministic (numpy PCG64 with fixed seeds).  Used by the tests (tests/workloads.py
re-exports this module), by bench.py's other_configs legs and by the lab scripts."""
import importlib

import numpy as np

# GRCh38 chr1-22, X, Y lengths in Mb (rounded) -- only the proportions matter
GRCH38_MB = [248, 242, 198, 190, 182, 171, 159, 145, 138, 134, 135, 133, 114, 107, 102, 90, 83,
             80, 59, 64, 47, 51, 156, 57]


def amplicon_panel(genome_length=29_903, size=400, step=300):
    """tiled amplicons [start, end] (inclusive), primers are the 25 bp at each end"""
    starts = np.arange(0, genome_length - size, step, dtype=np.int64)
    starts = starts[starts + size - 1 < genome_length][:98]
    return starts.astype(np.uint32), (starts + size - 1).astype(np.uint32)


def amplicon_reads(n_pairs, seed=12345, read_length=150, straddle_fraction=0.10):
    """configs[2]: pairs drawn from 98 tiled amplicons with weights x - x^2 + 0.1; mate 1 starts
    within 25 bp of the amplicon start, mate 2 ends within 25 bp of the amplicon end; a tenth of
    the pairs take mate 2 from the NEXT amplicon (they must be dropped by the FILTER)."""
    rng = np.random.default_rng(seed)
    a0, a1 = amplicon_panel()
    k = a0.size
    x = np.arange(k) / (k - 1)
    w = x - x * x + 0.1
    w /= w.sum()
    amp = rng.choice(k, size=n_pairs, p=w)
    straddle = (rng.random(n_pairs) < straddle_fraction) & (amp < k - 1)
    amp2 = np.where(straddle, amp + 1, amp)
    s1 = a0[amp].astype(np.int64) + rng.integers(0, 26, size=n_pairs)
    e2 = a1[amp2].astype(np.int64) - rng.integers(0, 26, size=n_pairs)
    starts = np.empty(2 * n_pairs, np.uint32)
    ends = np.empty(2 * n_pairs, np.uint32)
    starts[0::2] = s1
    ends[0::2] = s1 + read_length - 1
    starts[1::2] = e2 - read_length + 1
    ends[1::2] = e2
    return starts, ends, a0, a1, straddle


def wgs_shape(total_length, total_pairs, read_length=150):
    """contig lengths and pair counts of the configs[4] shape (24 contigs ~ GRCh38 proportions)"""
    frac = np.array(GRCH38_MB, dtype=np.float64) / sum(GRCH38_MB)
    lengths = np.maximum((frac * total_length).astype(np.int64), 2 * read_length + 1)
    pairs = np.maximum((frac * total_pairs).astype(np.int64), 1)
    return lengths, pairs


def wgs_contigs(total_length, total_pairs, read_length=150, seed=12345, only=None):
    """configs[4] shape: 24 contigs with lengths proportional to GRCh38, reads proportional to
    length, rand_reads_uniform(seed + c) per contig.  only: the contig ids to generate (a rank's share of the
    full-size configuration), in that order -- the seeds stay those of the whole genome."""
    pkg = importlib.import_module("genome-downsampler_amd")
    lengths, pairs = wgs_shape(total_length, total_pairs, read_length)
    ids = list(range(lengths.size)) if only is None else list(only)
    ss, ee = [], []
    for c in ids:
        s, e = pkg.reads_gen(pkg.KIND_UNIFORM, int(pairs[c]), int(lengths[c]), read_length, seed=seed + c)
        ss.append(s)
        ee.append(e)
    offs = np.concatenate([[0], np.cumsum(2 * pairs[ids])]).astype(np.uint64)
    return np.concatenate(ss), np.concatenate(ee), offs, lengths[ids].astype(np.uint32)


def cfg5_heaviest_share(world=8, read_length=150, max_coverage=50):
    """configs[4] at FULL size (24 contigs, 1.5e9 positions, 1e9 reads) dealt to `world` ranks by
    sharding.assign_contigs: the contig ids of the rank with the most reads (its contigs at full length are one
    GPU's real share -- chr1-sized contigs of > 10^8 positions, unlike the 1/8-scale genome of wgs_contigs(...))"""
    sharding = importlib.import_module("genome-downsampler_amd.sharding")
    lengths, pairs = wgs_shape(int(1.5e9), int(0.5e9), read_length)
    owned = sharding.assign_contigs((2 * pairs).tolist(), world, contig_lengths=lengths.tolist(),
                                    read_length=read_length, max_coverage=max_coverage)
    heaviest = max(owned, key=lambda o: sum(int(pairs[c]) for c in o))
    return heaviest, owned


def clipped_mix(starts, ends, fraction, max_clip=50, seed=7):
    """a fraction of the reads soft-clipped by 1 ... max_clip bases at either end (BamApi derives a read's span from
    its CIGAR, libs/bam-api/src/read.cpp:11-13, so real BAMs are never of one length): returns new (starts, ends);
    reads stay inside their contig because they only shrink"""
    rng = np.random.default_rng(seed)
    s = np.array(starts, dtype=np.uint32, copy=True)
    e = np.array(ends, dtype=np.uint32, copy=True)
    pick = np.flatnonzero(rng.random(s.size) < fraction)
    clip = rng.integers(1, max_clip + 1, size=pick.size).astype(np.uint32)
    front = rng.random(pick.size) < 0.5
    s[pick[front]] += clip[front]
    e[pick[~front]] -= clip[~front]
    return s, e


def lengthened_mix(starts, ends, contig_read_offsets, contig_lengths, fraction, max_extra=20, seed=9):
    """a fraction of the reads LENGTHENED by 1 ... max_extra bases (a deletion against the reference lengthens a read's
    span: BamApi takes it from the CIGAR's reference length, libs/bam-api/src/read.cpp:11-13), never past the read's
    contig: returns the new ends"""
    rng = np.random.default_rng(seed)
    e = np.array(ends, dtype=np.uint32, copy=True)
    offs = np.asarray(contig_read_offsets, dtype=np.int64)
    last = np.repeat(np.asarray(contig_lengths, dtype=np.int64) - 1, np.diff(offs))
    j = rng.choice(e.size, size=int(e.size * fraction), replace=False)
    e[j] = np.minimum(e[j].astype(np.int64) + rng.integers(1, max_extra + 1, size=j.size), last[j]).astype(np.uint32)
    return e
