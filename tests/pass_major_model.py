"""Host-side model of the PASS-MAJOR layout of the range-ranked route (csrc/kernels/pass_major.inc.hip) -- every
index its producer writes and its two consumers read, with bounds asserted -- so that the layout's arithmetic is
checked on the CPU before (and beside) any GPU run.  Written after round 2's abandoned 16-bit-index experiment
ended in a GPU memory-access fault whose code was not kept: this round the indexing was modelled first.

Layout.  The reads of a call are cut into PASSES of 8 192 consecutive reads.  k_pm_prepare_sort sorts every pass
by position range (digit = clamped global start >> shift, <= 256 ranges), stably, IN PLACE: slot P * 8192 + j of
the two 16-bit streams holds, for the pass's j-th record in (range, read index) order,
    keys16 = global start & (2^shift - 1)        idx16 = read index - P * 8192        (both < 2^15 resp. 2^13)
and two tables laid out [range][pass] (row pitch a multiple of 4) hold, per (range d, pass P),
    cnt[d][P] = records of range d in pass P       lst[d][P] = where they begin inside the pass.
An exclusive scan over the flattened cnt table gives T[d][P] = number of records of lower ranges + of range d in
lower passes: the position the slice WOULD have in a range-major array, which is never materialised.  Range d's
records in read-index order are the concatenation of its slices in pass order; flat position x in [T[d][0],
T[d+1][0]) lies in the last pass P with T[d][P] <= x, at slot P * 8192 + lst[d][P] + (x - T[d][P])."""
import numpy as np

PASS = 8192


def pitch_for(n):
    n_pass = (n + PASS - 1) // PASS
    return (n_pass + 3) & ~3


def producer(gstart, shift):
    """-> keys16, idx16 (length n), cnt, lst ([256][pitch] uint32)"""
    n = gstart.size
    pitch = pitch_for(n)
    keys16 = np.zeros(n, np.uint16)
    idx16 = np.zeros(n, np.uint16)
    cnt = np.zeros((256, pitch), np.uint32)
    lst = np.zeros((256, pitch), np.uint32)
    for P in range((n + PASS - 1) // PASS):
        lo, hi = P * PASS, min(n, (P + 1) * PASS)
        d = (gstart[lo:hi] >> shift).astype(np.int64)
        assert d.max() < 256
        order = np.argsort(d, kind="stable")
        keys16[lo:hi] = (gstart[lo:hi][order] & ((1 << shift) - 1)).astype(np.uint16)
        idx16[lo:hi] = order.astype(np.uint16)
        c = np.bincount(d, minlength=256).astype(np.uint32)
        cnt[:, P] = c
        lst[:, P] = np.concatenate([[0], np.cumsum(c)[:-1]]).astype(np.uint32)
    return keys16, idx16, cnt, lst


def scan_table(cnt):
    """exclusive scan over the flattened [range][pass] table, total appended (launch_exclusive_scan, write_total)"""
    flat = cnt.reshape(-1).astype(np.uint64)
    return np.concatenate([[0], np.cumsum(flat)]).astype(np.uint32)


def relevant_passes(d, shift, ltot, roff, poff):
    """passes that can hold records of range d: those of the contigs whose positions overlap the range"""
    pos0 = d << shift
    if pos0 > ltot:
        return 0, 0
    pos1 = min(pos0 + (1 << shift), ltot + 1) - 1       # last position of the range (ltot itself included: the
    c_first = int(np.searchsorted(poff, pos0, side="right") - 1)   # clamp rule puts a zero-length contig's reads there)
    c_last = int(np.searchsorted(poff, pos1, side="right") - 1)
    c_first = min(max(c_first, 0), len(roff) - 2)
    c_last = min(max(c_last, 0), len(roff) - 2)
    return int(roff[c_first]) // PASS, (int(roff[c_last + 1]) + PASS - 1) // PASS


def flat_to_slot(T, lst, pitch, d, x, p_lo, p_hi):
    """the consumer's mapping, as a binary search over the range's row (the device walks a cursor instead)"""
    row = T[d * pitch + p_lo:d * pitch + p_hi + 1]
    k = int(np.searchsorted(row, x, side="right") - 1)       # last pass with T <= x
    assert 0 <= k < p_hi - p_lo, (d, x, k)
    P = p_lo + k
    return P, P * PASS + int(lst[d, P]) + (x - int(row[k]))


def wave_cursor_walk(T, lst, pitch, d, lo, hi, p_lo, p_hi, chunk=1024, depth=1):
    """The device's walk, lane for lane: the workgroup's 16 waves take 64 consecutive flat positions each per chunk;
    a wave keeps a cursor k0 (last pass whose slice begins at or before its first position), advances it with one
    row read of 64 candidates per step, counts the slice borders inside its 64 positions, and every lane derives its
    pass and slot.  Returns the slots in flat order."""
    row = T[d * pitch + p_lo:d * pitch + p_hi + 1].astype(np.int64)
    n_rel = p_hi - p_lo

    def cand_of(k0):
        i = k0 + 1 + np.arange(64)
        return np.where(i <= n_rel, row[np.minimum(i, n_rel)], np.int64(1) << 40)   # beyond the row: never <= x

    slots = np.zeros(hi - lo, np.int64)
    passes = np.zeros(hi - lo, np.int64)
    n_chunks = (hi - lo + chunk - 1) // chunk
    for w in range(chunk // 64):
        k0 = 0
        for c in range(n_chunks):
            j0 = lo + c * chunk + 64 * w
            if j0 >= hi:
                break
            # advance: passes whose slices begin at or before j0
            while True:
                cand = cand_of(k0)
                n_before = int((cand <= j0).sum())
                k0 += n_before
                if n_before < 64:
                    break
            cand = cand_of(k0)
            x = j0 + np.arange(64)
            n_in = int((cand <= j0 + 63).sum())
            s = np.full(64, k0, np.int64)
            for t in range(n_in):
                s += (x >= cand[t]).astype(np.int64)
            if n_in == 64:      # more than 64 borders inside 64 positions (empty slices): the slow, exact way
                s = np.array([np.searchsorted(row[:n_rel + 1], xi, side="right") - 1 for xi in x], np.int64)
            live = x < hi
            s = np.minimum(s, n_rel - 1)
            P = p_lo + s
            slot = P * PASS + lst[d, P].astype(np.int64) + (x - row[s])
            slots[(x - lo)[live]] = slot[live]
            passes[(x - lo)[live]] = P[live]
    return slots, passes
