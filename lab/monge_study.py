"""Lab study: is there intra-contig parallelism for the uniform-span sweep on data that is
deeper than M everywhere (no cut points), e.g. cfg5's Poisson coverage 2 x M?

In kept-count form the sweep is a longest-path problem: K(p) = kept reads with start <= p is the
pointwise-minimal solution of
    K(p) >= K(p-1),   K(p) >= K(p+1) - c(p+1),   K(p) >= K(p-ell) + need(p)
so the state entering a block is the vector x = K over the ell positions before it, and a block (or any run
of blocks) acts on it as a MAX-PLUS LINEAR map  y(i) = max_j ( x(j) + W[j][i] ).  A scan over such maps
would give intra-contig parallelism -- at ell^3 per product in general, ell^2 if the W are (inverse)
Monge  (W[j][i] + W[j+1][i+1] >= W[j][i+1] + W[j+1][i]),  because then (max,+) products can use SMAWK /
divide and conquer.  This script builds W for runs of blocks of synthetic data by longest paths and counts
Monge violations, and counts the rows of W that differ by more than a constant (1: the run forgets its
entry state, as behind a cut point; ell: every entry position matters).

usage: python lab/monge_study.py [ell=30] [mean reads per position=0.67] [M=10] [blocks=1,2,4,8]"""
import sys
import numpy as np

ell = int(sys.argv[1]) if len(sys.argv) > 1 else 30
lam = float(sys.argv[2]) if len(sys.argv) > 2 else 0.67
M = int(sys.argv[3]) if len(sys.argv) > 3 else 10
runs = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 2, 4, 8]
NEG = -10**9


def transfer(c, need, n_blocks, first):
    """W[j][i]: longest path from source position first - ell + j to target position first + (n_blocks-1)*ell + i,
    over the constraint graph restricted to positions >= first - ell (the state window and what follows)"""
    lo = first - ell
    hi = first + n_blocks * ell          # exclusive
    n = hi - lo
    W = np.full((ell, ell), NEG, dtype=np.int64)
    for j in range(ell):
        d = np.full(n, NEG, dtype=np.int64)
        d[j] = 0
        # edges: p-1 -> p (0), p+1 -> p (-c(p+1)), p-ell -> p (need(p)); Bellman-Ford to a fixpoint
        for _ in range(4 * n):
            old = d.copy()
            d[1:] = np.maximum(d[1:], d[:-1])                                  # K(p) >= K(p-1)
            d[:-1] = np.maximum(d[:-1], d[1:] - c[lo + 1:hi])                  # K(p) >= K(p+1) - c(p+1)
            d[ell:] = np.maximum(d[ell:], d[:-ell] + need[lo + ell:hi])        # K(p) >= K(p-ell) + need(p)
            if np.array_equal(old, d):
                break
        W[j] = d[n - ell:]
    return W


def monge_violations(W):
    a = W[:-1, :-1] + W[1:, 1:]
    b = W[:-1, 1:] + W[1:, :-1]
    ok = (W[:-1, :-1] > NEG // 2) & (W[1:, 1:] > NEG // 2) & (W[:-1, 1:] > NEG // 2) & (W[1:, :-1] > NEG // 2)
    return int(((a < b) & ok).sum()), int(((a > b) & ok).sum()), int(ok.sum())


rng = np.random.default_rng(5)
L = 40 * ell
c = rng.poisson(lam, size=L + ell).astype(np.int64)
c[L - ell + 1:] = 0
CP = np.concatenate([[0], np.cumsum(c)])
p = np.arange(L + ell)
cov = CP[p + 1] - CP[np.maximum(0, p + 1 - ell)]
need = np.minimum(cov, M)
print(f"ell {ell}, {lam} reads per position, M {M}: coverage {cov[ell:L - ell].mean():.1f} "
      f"(min {cov[ell:L - ell].min()}), positions with coverage <= M: {(cov[ell:L - ell] <= M).sum()}")
for nb in runs:
    tot_inv = tot_mon = tot = 0
    ranks = []
    for first in range(4 * ell, 12 * ell, ell):
        W = transfer(c, need, nb, first)
        inv_viol, mon_viol, cells = monge_violations(W)
        tot_inv += inv_viol; tot_mon += mon_viol; tot += cells
        # rows that differ by more than an additive constant: 1 means the map forgets the state it is
        # given (W[j][i] = a(j) + b(i): a cut point inside the run), ell means every entry position matters
        ranks.append(len({tuple((row - row.max()).tolist()) for row in W}))
    print(f"  runs of {nb} block(s): 2x2 minors violating inverse Monge {tot_inv} / Monge {tot_mon} of {tot}; "
          f"rows distinct up to a constant: mean {np.mean(ranks):.1f} of {ell}")
