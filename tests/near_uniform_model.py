"""Host-side model of the NEAR-UNIFORM route (csrc/kernels/near_uniform.inc.hip): calls whose reads have one
dominant span ell and a few shorter ones (soft clips, insertions -- BamApi takes a read's span from its CIGAR,
libs/bam-api/src/read.cpp:11-13).  Written and checked against the oracle before the kernels, as the pass-major
layout was.

The canonical greedy (oracle/qmcp_oracle.c: at a position with a deficit take the unselected covering reads with the
largest end, then the largest start, then the smallest index) files every read under its END: bucket v = end - ell + 1.
A read of the dominant span starts at its bucket (v == start); a shorter one is RELEASED late, at start > v, and from
then on goes before the bucket's regular members (same end, larger start).  A read selected at time p covers [p, end]
whatever its start, so the sweep over counts is the uniform one as long as no exception is ever wanted.

  model      the uniform sweep over the REGULAR reads only, with need'(p) = min(cov_all(p), M) - (selected exceptions
             covering p from the time they were selected);
  verify     an unselected exception x = (v, s, e) is taken by the true greedy at the first time t in [s, e] at which
             the demand exceeds what the regular members of the buckets (v, t] still offer.  That needs every bucket
             in (v, t] exhausted in the model's FINAL counts, so nearly every exception is cleared by one look at
             bucket s; the rest replay the model's time-resolved picks over the run of exhausted buckets around s
             from final counts alone (below the run nothing is picked late);
  iterate    per contig the earliest such event (highest priority first) is exact -- before it the greedy and the model
             agree -- so that exception is marked selected at that time, need' drops by one on [t, e], and the model is
             swept again; the loop ends when no exception is wanted.  The kept set is then the first S(p) regular
             reads of every start p in index order plus the selected exceptions."""
import numpy as np


def regular_sweep(c, need, ell):
    """the uniform greedy over counts: S[p] kept among the c[p] regular reads starting at p; need may be any int array.
    Also returns the time-resolved picks as a list of (t, bucket, k) for the brute-force checks."""
    L = c.size
    cur = np.zeros(L, np.int64)
    picks = []
    for t in range(L):
        lo = max(0, t - ell + 1)
        d = int(need[t]) - int(cur[lo:t].sum())
        u = t
        while d > 0 and u >= lo:
            k = min(d, int(c[u] - cur[u]))
            if k > 0:
                cur[u] += k
                d -= k
                picks.append((t, u, k))
            u -= 1
    return cur, picks


def split_reads(starts, ends, ell):
    span = ends.astype(np.int64) - starts.astype(np.int64) + 1
    reg = np.flatnonzero(span == ell)
    exc = np.flatnonzero(span != ell)
    assert (span[exc] < ell).all(), "only shorter exceptions are modelled"
    return reg, exc


def coverage(starts, ends, L):
    d = np.zeros(L + 1, np.int64)
    np.add.at(d, starts.astype(np.int64), 1)
    np.add.at(d, ends.astype(np.int64) + 1, -1)
    return np.cumsum(d)[:L]


def verify(c, S, need, ell, exc_s, exc_e, exc_idx, unpicked):
    """earliest (time, priority) at which an unselected exception would be taken, from FINAL counts only.
    Returns None, or (t, x) with x an index into the exception arrays; or "unresolved"."""
    L = c.size
    best = None
    for x in np.flatnonzero(unpicked):
        s, e = int(exc_s[x]), int(exc_e[x])
        v = e - ell + 1                        # the exception's bucket (may lie before the contig)

        def exhausted(u):
            return u < 0 or S[u] == c[u]
        # every bucket in (v, s] must be exhausted for x ever to be reached
        if not all(exhausted(u) for u in range(s, v, -1)):
            continue
        # the run of exhausted buckets downwards from v
        u1 = v + 1
        while u1 - 1 >= 0 and exhausted(u1 - 1) and s - (u1 - 1) < ell:
            u1 -= 1
        if u1 - 1 >= 0 and exhausted(u1 - 1):
            # ell used-up buckets in a row below the read: not modelled.  It only matters if nothing earlier is wanted
            # (behind a wanted exception the sweep ran on a need it could not meet): the read enters the contest with
            # its release time and the lowest priority, and the model gives up only if that wins.
            key = (s, 1 << 40, 0, 0)
            if best is None or key < best[0]:
                best = (key, "unresolved")
            continue
        u1 = max(u1, 0)
        # replay from the anchor u1 - 1 (the bucket below the run: never exhausted, so nothing below it is picked
        # at or after its own time, and the final counts below it are what every later window sees).  The replayed
        # buckets [base, t) keep their own time-resolved counts: they may leave the window while x is alive.
        anchor = u1 - 1
        base = anchor if anchor >= 0 else 0
        cur = np.zeros(e - base + 2, np.int64)
        avail = 0            # what the regular members of buckets (v, t] still offer
        if anchor >= 0:
            lo = max(0, anchor - ell + 1)
            d = int(need[anchor]) - int(S[lo:anchor].sum())
            cur[0] = min(max(d, 0), int(c[anchor]))
        t = u1
        while t <= e and t < L:
            lo = max(0, t - ell + 1)
            fixed = int(S[lo:base].sum()) if base > lo else 0
            repl = int(cur[max(lo, base) - base:t - base].sum())
            d = max(int(need[t]) - fixed - repl, 0)
            if t > v:
                avail += int(c[t])
            if t >= s and d > avail:
                key = (t, -e, -s, int(exc_idx[x]))
                if best is None or key < best[0]:
                    best = (key, x)
                break
            avail -= min(d, avail)
            u = t
            while d > 0 and u >= base:     # top-down through what the replayed buckets still offer
                k = min(d, int(c[u]) - int(cur[u - base]))
                cur[u - base] += k
                d -= k
                u -= 1
            if t >= s and not exhausted(t):
                break      # bucket t keeps members for good: x is never reached later
            t += 1
    if best is not None and best[1] == "unresolved":
        return "unresolved"
    return None if best is None else (best[0][0], best[1])


def solve_near_uniform(starts, ends, L, M, ell, max_iter=64):
    """-> keep (bool per read), iterations, or None when the model gives up (the product then takes the general route)"""
    starts = starts.astype(np.int64)
    ends = ends.astype(np.int64)
    reg, exc = split_reads(starts, ends, ell)
    c = np.bincount(starts[reg], minlength=L).astype(np.int64)
    cov_all = coverage(starts, ends, L)
    need = np.minimum(cov_all, M).astype(np.int64)
    xs, xe = starts[exc], ends[exc]
    picked_at = np.full(exc.size, -1, np.int64)
    # an exception that starts where everything is kept is selected the moment it is released
    for x in range(exc.size):
        if cov_all[xs[x]] <= M:
            picked_at[x] = xs[x]
            need[xs[x]:xe[x] + 1] -= 1
    it = 0
    while True:
        S, _ = regular_sweep(c, need, ell)
        r = verify(c, S, need, ell, xs, xe, exc, picked_at < 0)
        if r is None:
            break
        if r == "unresolved" or it >= max_iter:
            return None
        t, x = r
        picked_at[x] = t
        need[t:xe[x] + 1] -= 1
        it += 1
    keep = np.zeros(starts.size, bool)
    keep[exc[picked_at >= 0]] = True
    order = np.argsort(starts[reg], kind="stable")
    rs = reg[order]
    first = np.concatenate([[0], np.cumsum(c)])
    for p in np.flatnonzero(S):
        keep[rs[first[p]:first[p] + S[p]]] = True
    return keep, it
