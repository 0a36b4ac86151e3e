"""Multi-GPU composition of the solver path: contigs are independent solves (the reference is
single-contig, libs/bam-api/src/bam_api.cpp:422), so they shard across ranks with no data-path
collective; the only exchange is the gather of the keep bitmasks (RCCL over xGMI on the GPU
box, gloo in the CPU tests).  Nothing here exists in the reference (it has no multi-GPU path).

This module is host-side plumbing only: which contigs a rank owns, how the reads of those
contigs are sliced, and how per-rank masks merge back into one global ReadIndex bitmask."""
import numpy as np


# Cost model of one rank's share (measured on MI355X): the bandwidth-bound stages
# cost time per READ and add up over a rank's contigs; the selection sweep is one serial chain per
# contig, all of a rank's chains side by side, so it costs the LONGEST contig's length -- unless the share's
# sweep is cut into stretches (cut points, speculative boundaries): then it is throughput too, per position.
# Whether it is cut is decided exactly as the solver decides it (csrc/qmcp_api.hip: launch_uniform_sweep,
# spec_wanted, spec_burn_blocks), from the AGGREGATE depth of everything the rank owns -- not contig by
# contig: a rank that mixes one deep contig with shallow ones sweeps whole contigs, and is priced so.
NS_PER_READ = 0.008        # prepare + partition + offsets + ranking: ~0.8 ms per 1e8 reads
NS_PER_POSITION = 1.5      # block-scan sweep on shallow data; deep data (event sweep) is ~0.5
NS_PER_POSITION_STRETCHES = 0.012   # the same sweep cut into stretches (2.0 ms per 187.5 M positions)
STRETCH_DEPTH = 11.0       # kSpecDepth (= kGenDepth): aggregate coverage in units of M below which boundaries are speculated on
CUT_DEPTH = 1.3            # kSpecMinDepth: below it nearly every window holds a real cut point
MAX_SPLIT_CONTIGS = 256    # the stretch tables take calls of fewer contigs than this


def spec_burn_blocks(depth):
    """run-in of a speculative boundary, in blocks (csrc/qmcp_api.hip: spec_burn_blocks)"""
    return 320 if depth < 2.1 else 640 if depth < 2.6 else 1152 if depth < 3.1 else 2304 if depth < 4.1 else 1536


def spec_depth_in_sigma(depth, max_coverage):
    """the depth at which M = 50 -- where the run-in table was measured -- sits as many standard deviations above M as
    this depth does at max_coverage; never below the depth itself (csrc/api/uniform_sweep.inc.hip: spec_depth_in_sigma)"""
    if not depth > 1.0:
        return depth
    y = (max_coverage / 50.0) ** 0.5 * (depth - 1.0) / depth ** 0.5
    x = 0.5 * (y + (y * y + 4.0) ** 0.5)
    return max(x * x, depth)


def share_sweeps_as_stretches(reads, positions, n_contigs, read_length, max_coverage):
    """the solver's own predicate on a whole share (a list of contigs solved in one call)"""
    if not read_length or not max_coverage or positions <= 0 or n_contigs >= MAX_SPLIT_CONTIGS:
        return False
    depth = float(reads) * float(read_length) / (float(positions) * float(max_coverage))
    if depth <= CUT_DEPTH:
        return positions >= 128 * read_length
    return depth < STRETCH_DEPTH and positions >= 8 * spec_burn_blocks(spec_depth_in_sigma(depth, max_coverage)) * read_length


def share_cost(reads, positions, longest, n_contigs, read_length=None, max_coverage=None):
    if share_sweeps_as_stretches(reads, positions, n_contigs, read_length, max_coverage):
        return NS_PER_READ * reads + NS_PER_POSITION_STRETCHES * positions
    return NS_PER_READ * reads + NS_PER_POSITION * longest


def rank_cost(read_counts, contig_lengths, contigs, read_length=None, max_coverage=None):
    reads = sum(int(read_counts[c]) for c in contigs)
    positions = sum(int(contig_lengths[c]) for c in contigs)
    longest = max([int(contig_lengths[c]) for c in contigs], default=0)
    return share_cost(reads, positions, longest, len(contigs), read_length, max_coverage)


def assign_contigs(read_counts, world_size, contig_lengths=None, read_length=None, max_coverage=None):
    """deterministic longest-processing-time assignment of contigs to ranks.  With contig_lengths the
    cost of a rank is reads * NS_PER_READ + (its longest contig) * NS_PER_POSITION -- the sweep's
    chains run side by side, so a rank pays for its longest one --, without it the read count alone;
    with read_length and max_coverage as well, a share whose sweep the solver cuts into stretches (see
    above) pays per position instead.  Returns a list (per rank) of ascending contig ids."""
    n = len(read_counts)
    if contig_lengths is None:
        contig_lengths = [0] * n
    def alone(c):
        return rank_cost(read_counts, contig_lengths, [c], read_length, max_coverage)
    order = sorted(range(n), key=lambda c: (-alone(c), c))
    owned = [[] for _ in range(world_size)]
    for c in order:
        # the rank whose cost AFTER taking c is smallest
        r = min(range(world_size), key=lambda k: (rank_cost(read_counts, contig_lengths, owned[k] + [c],
                                                            read_length, max_coverage), k))
        owned[r].append(c)
    return [sorted(o) for o in owned]


def local_problem(starts, ends, contig_read_offsets, contig_lengths, contigs):
    """slice the reads of `contigs` (ascending) into one contiguous local problem"""
    offs = np.asarray(contig_read_offsets, dtype=np.uint64)
    parts_s, parts_e, counts = [], [], []
    for c in contigs:
        lo, hi = int(offs[c]), int(offs[c + 1])
        parts_s.append(np.asarray(starts[lo:hi], dtype=np.uint32))
        parts_e.append(np.asarray(ends[lo:hi], dtype=np.uint32))
        counts.append(hi - lo)
    s = np.concatenate(parts_s) if parts_s else np.zeros(0, np.uint32)
    e = np.concatenate(parts_e) if parts_e else np.zeros(0, np.uint32)
    local_offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    lengths = np.asarray(contig_lengths, dtype=np.uint32)[list(contigs)] if len(contigs) else \
        np.zeros(0, np.uint32)
    return s, e, local_offs, lengths


def merge_masks(gathered_masks, owned, contig_read_offsets, n_reads):
    """place every rank's local keep bits at their global ReadIndex positions"""
    offs = np.asarray(contig_read_offsets, dtype=np.uint64)
    bits = np.zeros(((int(n_reads) + 63) // 64) * 64, dtype=np.uint8)
    for rank, contigs in enumerate(owned):
        local = np.unpackbits(np.ascontiguousarray(gathered_masks[rank]).view(np.uint8),
                              bitorder="little")
        pos = 0
        for c in contigs:
            lo, hi = int(offs[c]), int(offs[c + 1])
            bits[lo:hi] = local[pos:pos + (hi - lo)]
            pos += hi - lo
    return np.packbits(bits, bitorder="little").view(np.uint64).copy()


def gather_masks(local_mask, max_words, dist, device=None):
    """all-gather of the per-rank keep bitmasks, padded to `max_words` 64-bit words.
    `dist` is torch.distributed (nccl == RCCL on the GPU box, gloo on CPU)."""
    import torch
    world = dist.get_world_size()
    buf = torch.zeros(max_words, dtype=torch.int64, device=device)
    src = torch.from_numpy(np.ascontiguousarray(local_mask).view(np.int64))
    buf[:src.numel()] = src.to(buf.device)
    out = torch.zeros(max_words * world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, buf)
    host = out.cpu().numpy().view(np.uint64)
    return [host[r * max_words:(r + 1) * max_words] for r in range(world)]
