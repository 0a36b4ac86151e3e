"""CPU model of the pass-major layout (tests/pass_major_model.py, round 4 form: slices padded to whole groups of 64
slots, one descriptor word per wave-slot): the producer's slots and tables, the scanned table,
the descriptors, the three consumers -- every index in bounds, every range's records recovered in
read-index order, and the keep mask of walk + settle equal to "the S(p) lowest read indices of every start
position p" for random quotas -- on ragged inputs (last pass partial, ranges straddling contig borders, empty ranges
and empty slices, one range holding everything, slices of a few records)."""
import numpy as np
import pytest

import pass_major_model as pm


def _case(rng, lengths, counts, shift, skew=None):
    lengths = np.asarray(lengths, np.int64)
    counts = np.asarray(counts, np.int64)
    poff = np.concatenate([[0], np.cumsum(lengths)])
    roff = np.concatenate([[0], np.cumsum(counts)])
    gs = []
    for c, (L, k) in enumerate(zip(lengths, counts)):
        if skew is not None:
            s = skew(rng, int(k), int(L))
        else:
            s = rng.integers(0, max(int(L), 1), size=int(k))
        gs.append(poff[c] + np.minimum(s, max(int(L) - 1, 0)))
    return np.concatenate(gs).astype(np.uint32), roff, poff, int(poff[-1])


CASES = [
    # lengths, counts, shift
    ([100_000], [3 * 8192 + 77], 9),                                   # one contig, last pass partial
    ([70_001, 33_333, 250_000, 1_000], [9_000, 20_011, 30_000, 5], 11),  # ranges straddle contig borders
    ([5_000, 5_000, 5_000], [0, 25_000, 0], 6),                         # contigs without reads
    ([300], [20_000], 3),                                               # tiny genome, 38 ranges
    ([40_000], [17_000], 15),                                           # ONE range holds everything
    ([(1 << 21) - 5], [40_000], 13),                                    # 256 ranges, sparse: slices of a few records
]


def _expected_mask(gs, quota_of):
    """the quota_of(p) lowest read indices of every global start p"""
    n = gs.size
    order = np.argsort(gs, kind="stable")
    mask = np.zeros(n, bool)
    sorted_g = gs[order]
    starts = np.concatenate([[0], np.nonzero(np.diff(sorted_g))[0] + 1, [n]])
    for a, b in zip(starts[:-1], starts[1:]):
        k = min(int(quota_of(int(sorted_g[a]))), b - a)
        mask[order[a:a + k]] = True
    return mask


@pytest.mark.parametrize("lengths,counts,shift", CASES)
def test_layout_round_trip_bounds_and_mask(lengths, counts, shift):
    rng = np.random.default_rng(len(lengths) * 1000 + shift)
    gs, roff, poff, ltot = _case(rng, lengths, counts, shift)
    n = gs.size
    n_ranges = (ltot >> shift) + 1
    assert n_ranges <= 256
    keys16, idx16, cntp, lstw = pm.producer(gs, shift, n_ranges)
    pitch = pm.pitch_for(n)
    stride = pm.stride_for(n_ranges)
    Tp = pm.scan_table(cntp)
    assert Tp.size == 256 * pitch + 1 and int(Tp[-1]) % 64 == 0 and int(Tp[-1]) >= n
    desc, range_start = pm.descriptors(Tp, lstw, n, n_ranges)
    assert int(range_start[-1]) == n
    # random quotas: 0, 1, a few, everything
    width = 1 << shift
    quota = rng.choice([0, 0, 1, 1, 2, 5, 1 << 20], size=ltot + width)
    seen = np.zeros(n, bool)
    mask = np.zeros(n, bool)
    kept_total = 0
    amb_all = {}
    for d in range(n_ranges):
        g0, n_ws = pm.range_wave_slots(Tp, pitch, d)
        true_count = int(range_start[d + 1]) - int(range_start[d])
        if n_ws == 0:
            assert true_count == 0
            continue
        # every record of the range, in read-index order, through the descriptors
        reads = []
        for ws in range(n_ws):
            slot0, nv, P = pm.unpack(desc[g0 + ws], stride)
            assert slot0 + nv <= pitch * stride and slot0 // stride == P
            r = P * pm.PASS + idx16[slot0:slot0 + nv].astype(np.int64)
            assert idx16[slot0:slot0 + nv].max() < pm.PASS
            reads.append(r)
        reads = np.concatenate(reads)
        assert reads.size == true_count and reads.max() < n and np.all(np.diff(reads) > 0)
        assert np.all((gs[reads] >> shift) == d)
        assert not seen[reads].any()
        seen[reads] = True
        # bucket counts
        hist = pm.offsets(keys16, desc, Tp, pitch, d, shift, stride)
        assert np.array_equal(hist, np.bincount((gs[reads] & (width - 1)).astype(np.int64), minlength=width))
        # walk (settled below)
        amb, kept = pm.walk(keys16, idx16, desc, Tp, pitch, d, quota[d * width:(d + 1) * width], stride, n, mask, rng)
        kept_total += kept
        amb_all[d] = amb
    assert seen.all()
    for d, amb in amb_all.items():
        kept_total += pm.settle(amb, keys16, idx16, desc, Tp, pitch, d, n, stride, mask)
    want = _expected_mask(gs, lambda p: quota[p])
    assert np.array_equal(mask, want) and kept_total == int(want.sum())


def test_skewed_passes_with_long_and_empty_slices():
    """reads sorted by position inside the contig: a pass's records fall into one or two ranges (slices of
    thousands of records, most slices empty); quotas that run out inside chunks"""
    rng = np.random.default_rng(2)
    gs, roff, poff, ltot = _case(rng, [200_000], [5 * 8192 + 1], 10,
                                 skew=lambda r, k, L: np.sort(r.integers(0, L // 50, size=k) * 50))
    n = gs.size
    n_ranges = (ltot >> 10) + 1
    keys16, idx16, cntp, lstw = pm.producer(gs, 10, n_ranges)
    pitch = pm.pitch_for(n)
    Tp = pm.scan_table(cntp)
    desc, range_start = pm.descriptors(Tp, lstw, n, n_ranges)
    stride = pm.stride_for(n_ranges)
    quota = rng.integers(0, 12, size=ltot + 1024)
    mask = np.zeros(n, bool)
    ambs = {}
    for d in range(n_ranges):
        g0, n_ws = pm.range_wave_slots(Tp, pitch, d)
        if n_ws == 0:
            continue
        ambs[d], _ = pm.walk(keys16, idx16, desc, Tp, pitch, d, quota[d * 1024:(d + 1) * 1024], stride, n, mask, rng)
    assert sum(len(a) for a in ambs.values()) > 0        # the case is meant to produce quota-crossing groups
    for d, amb in ambs.items():
        pm.settle(amb, keys16, idx16, desc, Tp, pitch, d, n, stride, mask)
    assert np.array_equal(mask, _expected_mask(gs, lambda p: quota[p]))
