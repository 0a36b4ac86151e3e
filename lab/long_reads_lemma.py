"""lab (host only, no GPU): reads LONGER than the dominant span ell (deletions) -- what a near-uniform route for them
can rest on.  Checked here against the oracle on random instances:

 (1) LEMMA.  Let P be the long reads the canonical greedy keeps.  The greedy's selection of the REGULAR reads is the
     one-span greedy over the regular reads alone with need'(p) = need(p) - #{x in P covering p}, every x counted from
     its START s, whatever the time t in [s, v) (v = end - ell + 1) at which the greedy took it from the pool of early
     reads: there is no deficit in [s, t), so the unit x supplies there is never drawn on, and at t the others supply one
     less either way.  (Unlike a shorter exception, whose selection TIME matters, a long read needs only a yes / no.)
 (2) CERTIFICATE.  With P applied, x in P is needed iff at some t in [s, v) the sweep's raw demand (before its clamp at
     zero) is >= 0, i.e. the coverage without x would fall short of need(t); a long read that is not needed anywhere in
     [s, v) is dropped from P (of several such reads that overlap, the one of LOWEST priority first: the greedy takes
     the larger end at a deficit), its unit goes back into need', and the sweep runs again: rounds until P is stable.
     What is left of the dropped reads joins its bucket v as that bucket's LAST member.
   python lab/long_reads_lemma.py [cases = 200]"""
import sys, os
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "oracle")); sys.path.insert(0, R)
import oracle_py


def sweep(c, need, ell):
    """one-span greedy over counts with any integer need: kept counts S and the raw demand d(t) seen at every time"""
    L = c.size
    cur = np.zeros(L, np.int64); raw = np.zeros(L, np.int64)
    win = 0
    for t in range(L):
        lo = max(0, t - ell + 1)
        if t - ell >= 0: win -= cur[t - ell]          # (final: a bucket ell back gives nothing any more)
        d = int(need[t]) - int(cur[lo:t].sum())
        raw[t] = d
        u = t
        while d > 0 and u >= lo:
            k = min(d, int(c[u] - cur[u]))
            cur[u] += k; d -= k; u -= 1
    return cur, raw


cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(7)
lemma_ok = cert_ok = 0; rounds_hist = {}; n_long = n_kept = n_late = 0
for case in range(cases):
    L = int(rng.integers(600, 2500)); ell = int(rng.choice([20, 50, 80])); M = int(rng.choice([4, 8, 15]))
    depth = float(rng.choice([1.5, 3, 6, 12])); n = int(depth * M * L / ell)
    s = rng.integers(0, L - ell - 12, size=n).astype(np.int64); e = s + ell - 1
    lg = rng.random(n) < float(rng.choice([0.005, 0.02, 0.1]))
    e = np.where(lg, e + rng.integers(1, 11, size=n), e)
    mask = oracle_py.solve(s.astype(np.uint32), e.astype(np.uint32), np.array([L], np.uint32), M)
    kept = np.unpackbits(mask.view(np.uint8), bitorder="little")[:n].astype(bool)
    cov = np.zeros(L + 1, np.int64); np.add.at(cov, s, 1); np.add.at(cov, e + 1, -1); cov = np.cumsum(cov)[:L]
    need = np.minimum(cov, M)
    c = np.bincount(s[~lg], minlength=L).astype(np.int64)
    want = np.bincount(s[~lg & kept], minlength=L)

    def cov_of(sel):
        a = np.zeros(L + 1, np.int64); np.add.at(a, s[sel], 1); np.add.at(a, e[sel] + 1, -1)
        return np.cumsum(a)[:L]

    S, _ = sweep(c, need - cov_of(lg & kept), ell)
    lemma_ok += int(np.array_equal(S, want))
    # (2) from "every long read is in P", by rounds
    P = lg.copy(); idx = np.flatnonzero(lg); rounds = 0
    while True:
        rounds += 1
        # (a dropped long read is an ordinary member of its bucket v from time v on -- its LAST member: same end as the
        #  regular ones, smaller start)
        c2 = c + np.bincount(e[lg & ~P] - ell + 1, minlength=L)
        S, raw = sweep(c2, need - cov_of(P), ell)
        # every long read is certified against this sweep, both ways: x in P must be needed somewhere in [s, v) (raw
        # demand >= 0 with its unit applied); x not in P must never be wanted there (raw demand <= the long reads outside P
        # that outrank it and are in the pool at that time).  All violations are settled at once, tentatively -- the
        # earliest one of a round is exact, as in the short exceptions' scheme -- and the loop ends when none is left.
        out = [i for i in idx if not P[i]]
        flips = []
        for i in idx:
            w0, w1 = int(s[i]), int(e[i] - ell + 1)
            if P[i]:
                if (raw[w0:w1] <= -1).all(): flips.append(i)
            else:
                for t in range(w0, w1):
                    above = sum(1 for j in out if j != i and s[j] <= t < e[j] - ell + 1 and (e[j], s[j], -j) > (e[i], s[i], -i))
                    if raw[t] >= 1 + above:
                        flips.append(i); break
        if not flips: break
        # of the members of P that are nowhere needed, only those go that no other such member of LOWER priority overlaps
        # (coverage against window): a surplus of one over two overlapping members is the lower one's -- the greedy takes
        # the larger end at a deficit -- and the other is looked at again in the next round
        failing = [i for i in flips if P[i]]
        for i in flips:
            if P[i]:
                lower = [j for j in failing if j != i and s[j] <= e[i] - ell and e[j] >= s[i] and (e[j], s[j], -j) < (e[i], s[i], -i)]
                if lower: continue
            P[i] = not P[i]
        if rounds > 60: break
    rounds_hist[rounds] = rounds_hist.get(rounds, 0) + 1
    # the dropped reads a bucket keeps are its last S(v) - c(v) ones: larger start first, then smaller index
    late = np.zeros(n, bool)
    dropped = np.flatnonzero(lg & ~P)
    for v in np.unique(e[dropped] - ell + 1):
        take = int(S[v]) - int(c[v])
        if take > 0:
            mem = [i for i in dropped if e[i] - ell + 1 == v]
            mem.sort(key=lambda i: (-s[i], i))
            late[mem[:take]] = True
    n_long += int(lg.sum()); n_kept += int((lg & kept).sum()); n_late += int(late.sum())
    Sreg = np.minimum(S, c)
    cert_ok += int(np.array_equal(P | late, lg & kept) and np.array_equal(Sreg, want))
print(f"{cases} cases, {n_long} long reads, {n_kept} kept by the oracle ({n_late} of them as their bucket's last member)")
print(f"(1) regular selection reproduced with the kept long reads applied from their start: {lemma_ok} of {cases}")
print(f"(2) rounds from 'all long reads' end on the oracle's pool-selected set: {cert_ok} of {cases}; rounds {dict(sorted(rounds_hist.items()))}")
