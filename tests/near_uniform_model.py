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


# How far below a read (in spans) the replay looks for a state to start from -- an anchor (a bucket that keeps members
# for good) or a cut point.  Round 3 looked one span down; shallow data (1.5 x M: nearly everything is kept over long
# stretches) has runs of used-up buckets longer than that, and the replay works from any distance: nothing below an
# anchor is picked at or after the anchor's time, however far the anchor is (k_nu_replay: kNuReach).
REACH = 6


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
        while u1 - 1 >= 0 and exhausted(u1 - 1) and s - (u1 - 1) < REACH * ell:
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


# ------------------------------------------------------------------------------------------------------------------
# Batched rounds (round 3, second form): every exception the sweep is seen to want is selected in the SAME round --
# tentatively -- and every selection is then certified against the next sweep: y is wanted at time t iff
#       d'(t) + k(t)  >  A_y(t) + r_y(t)
# with d' the sweep's own demand at t (its need already lowered by every selection), k(t) the exceptions selected at
# exactly t, A_y what the regular members of the buckets above y's still offer and r_y the exceptions above y that are
# candidates at t (released, alive, not selected before t).  A selected y must be wanted at its time and not before,
# an unselected one never.  Per contig the earliest open question (then the highest priority) is settled exactly --
# everything before it is certified -- and never changes again; the others are settled tentatively.
def _priority(e, s, idx):
    return (-e, -s, idx)


def _walk(c, S, need, ell, s, e, others, sel_time, L, cov_all=None, M=None):
    """the replay of one exception (s, e, its selection time or -1) against the sweep's final counts: returns the
    first (time, kind) at which the sweep and the selection disagree, kind in {"select", "unselect"}; None; or
    "unresolved".  others = [(s, e, idx, sel_time, above?)] exceptions of the same contig whose lives overlap."""
    v = e - ell + 1

    def exhausted(u):
        return u < 0 or S[u] == c[u]
    u1 = v + 1
    while u1 - 1 >= 0 and exhausted(u1 - 1) and s - (u1 - 1) < REACH * ell:
        u1 -= 1
    cut = None
    if u1 - 1 >= 0 and exhausted(u1 - 1):
        # no anchor within ell buckets.  A CUT POINT does as well: at a position p* with cov_all(p*) <= M every read
        # covering it is kept, so at time p* every bucket in (p* - ell, p*] is used up -- a known state to start from
        if cov_all is not None:
            for p in range(s - 1, max(s - REACH * ell, -1), -1):
                if cov_all[p] <= M:
                    cut = p
                    break
        if cut is None:
            return "unresolved"
    if cut is not None:
        base = max(cut - ell + 1, 0)
        cur = np.zeros(e - base + 2, np.int64)
        cur[:cut - base + 1] = c[base:cut + 1]
        u1 = cut + 1
        anchor = -1
    else:
        u1 = max(u1, 0)
        anchor = u1 - 1
        base = anchor if anchor >= 0 else 0
        cur = np.zeros(e - base + 2, np.int64)
        if anchor >= 0:
            lo = max(0, anchor - ell + 1)
            d = int(need[anchor]) - int(S[lo:anchor].sum())
            cur[0] = min(max(d, 0), int(c[anchor]))
    avail = 0
    t = u1
    last = e if sel_time < 0 else sel_time
    while t <= last and t < L:
        lo = max(0, t - ell + 1)
        fixed = int(S[lo:base].sum()) if (base > lo and cut is None) else 0   # (behind a cut nothing older matters)
        repl = int(cur[max(lo, base) - base:t - base].sum())
        raw = int(need[t]) - fixed - repl      # (may be negative: a selection that was not needed leaves a surplus)
        d = max(raw, 0)
        if t > v:
            avail += int(c[t])
        if t >= s:
            k = sum(1 for (zs, ze, zi, zt, above) in others if zt == t) + (1 if sel_time == t else 0)
            r = sum(1 for (zs, ze, zi, zt, above) in others if above and zs <= t <= ze and (zt < 0 or zt >= t))
            wanted = raw + k > avail + r
            if sel_time < 0 or t < sel_time:
                if wanted:
                    return (t, "select")
            elif not wanted:
                return (t, "unselect")
        avail -= min(d, avail)
        u = t
        while d > 0 and u >= base:
            kk = min(d, int(c[u]) - int(cur[u - base]))
            cur[u - base] += kk
            d -= kk
            u -= 1
        if sel_time < 0 and t >= s and not exhausted(t):
            break
        t += 1
    return None


def solve_near_uniform_batched(starts, ends, L, M, ell, max_rounds=200):
    starts = starts.astype(np.int64)
    ends = ends.astype(np.int64)
    reg, exc = split_reads(starts, ends, ell)
    c = np.bincount(starts[reg], minlength=L).astype(np.int64)
    cov_all = coverage(starts, ends, L)
    base_need = np.minimum(cov_all, M).astype(np.int64)
    xs, xe = starts[exc], ends[exc]
    sel = np.full(exc.size, -1, np.int64)
    for x in range(exc.size):
        if cov_all[xs[x]] <= M:
            sel[x] = xs[x]
    rounds = 0
    while True:
        rounds += 1
        need = base_need.copy()
        for x in np.flatnonzero(sel >= 0):
            need[sel[x]:xe[x] + 1] -= 1
        need_capped = need   # (the host model's sweep drops what it cannot meet: same as the device's cap)
        S, _ = regular_sweep(c, need_capped, ell)
        # the list: unselected exceptions the quick look does not clear, and every selected one
        listed = []
        for x in range(exc.size):
            if sel[x] >= 0:
                listed.append(x)
                continue
            s, e = int(xs[x]), int(xe[x])
            v = e - ell + 1
            if all((u < 0 or S[u] == c[u]) for u in range(s, v, -1)):
                listed.append(x)
        open_q = []
        for x in listed:
            s, e = int(xs[x]), int(xe[x])
            px = _priority(e, s, int(exc[x]))
            others = []
            for z in listed:
                if z == x or xe[z] < s or xs[z] > e:
                    continue
                others.append((int(xs[z]), int(xe[z]), int(exc[z]), int(sel[z]), _priority(int(xe[z]), int(xs[z]), int(exc[z])) < px))
            r = _walk(c, S, need, ell, s, e, others, int(sel[x]), L, cov_all, M)
            if r == "unresolved":
                open_q.append(((s, (1 << 40, 0, 0)), x, "unresolved"))
            elif r is not None:
                open_q.append(((r[0], px), x, r[1]))
        if not open_q:
            break
        if rounds >= max_rounds:
            return None
        open_q.sort(key=lambda q: q[0])
        if open_q[0][2] == "unresolved":
            return None
        # the earliest is exact; the rest are settled the same way, tentatively (one per exception)
        for (key, x, kind) in open_q:
            if kind == "select":
                sel[x] = key[0]
            elif kind == "unselect":
                sel[x] = -1
    keep = np.zeros(starts.size, bool)
    keep[exc[sel >= 0]] = True
    order = np.argsort(starts[reg], kind="stable")
    rs = reg[order]
    first = np.concatenate([[0], np.cumsum(c)])
    for p in np.flatnonzero(S):
        keep[rs[first[p]:first[p] + S[p]]] = True
    return keep, rounds
