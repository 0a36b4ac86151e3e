"""CPU model of the pass-major layout (tests/pass_major_model.py): the producer's slots and tables, the scanned
table, the consumers' flat-position -> slot mapping (by binary search and by the device's per-wave cursor walk):
every index in bounds, every range's records recovered in read-index order -- on ragged inputs (last pass partial,
ranges straddling contig borders, empty ranges and empty slices, one range holding everything)."""
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
    ([(1 << 21) - 5], [40_000], 13),                                        # 256 ranges, sparse: many empty slices
]


@pytest.mark.parametrize("lengths,counts,shift", CASES)
def test_layout_round_trip_and_bounds(lengths, counts, shift):
    rng = np.random.default_rng(len(lengths) * 1000 + shift)
    gs, roff, poff, ltot = _case(rng, lengths, counts, shift)
    n = gs.size
    assert (ltot >> shift) < 256
    keys16, idx16, cnt, lst = pm.producer(gs, shift)
    pitch = pm.pitch_for(n)
    T = pm.scan_table(cnt)
    assert T.size == 256 * pitch + 1 and int(T[-1]) == n
    assert int(idx16.max()) < pm.PASS and int(keys16.max()) < (1 << shift)
    seen = np.zeros(n, bool)
    for d in range((ltot >> shift) + 1):
        lo, hi = int(T[d * pitch]), int(T[(d + 1) * pitch])
        p_lo, p_hi = pm.relevant_passes(d, shift, ltot, roff, poff)
        # no record of the range lies in a pass outside [p_lo, p_hi)
        assert int(cnt[d, :p_lo].sum()) == 0 and int(cnt[d, p_hi:].sum()) == 0
        if lo == hi:
            continue
        assert p_hi > p_lo
        prev = -1
        for x in range(lo, hi, max(1, (hi - lo) // 257)):     # sampled by binary search ...
            P, slot = pm.flat_to_slot(T, lst, pitch, d, x, p_lo, p_hi)
            assert 0 <= slot < n and slot // pm.PASS == P
            read = P * pm.PASS + int(idx16[slot])
            assert read < n and (int(gs[read]) >> shift) == d and int(keys16[slot]) == int(gs[read]) & ((1 << shift) - 1)
        slots, passes = pm.wave_cursor_walk(T, lst, pitch, d, lo, hi, p_lo, p_hi)   # ... and all of it by the cursor walk
        assert slots.min() >= 0 and slots.max() < n and np.array_equal(slots // pm.PASS, passes)
        reads = passes * pm.PASS + idx16[slots].astype(np.int64)
        assert reads.max() < n and np.all(np.diff(reads) > 0)            # read-index order, no repeats
        assert np.all((gs[reads] >> shift) == d)
        assert np.array_equal(keys16[slots], (gs[reads] & ((1 << shift) - 1)).astype(np.uint16))
        assert not seen[reads].any()
        seen[reads] = True
    assert seen.all()        # every read is some range's record exactly once


def test_skewed_passes_with_long_and_empty_slices():
    """reads sorted by position inside the contig: a pass's records fall into one or two ranges (slices of
    thousands of records, most slices empty)"""
    rng = np.random.default_rng(2)
    gs, roff, poff, ltot = _case(rng, [200_000], [5 * 8192 + 1], 10,
                                 skew=lambda r, k, L: np.sort(r.integers(0, L, size=k)))
    keys16, idx16, cnt, lst = pm.producer(gs, 10)
    pitch, n = pm.pitch_for(gs.size), gs.size
    T = pm.scan_table(cnt)
    for d in range((ltot >> 10) + 1):
        lo, hi = int(T[d * pitch]), int(T[(d + 1) * pitch])
        if lo == hi:
            continue
        p_lo, p_hi = pm.relevant_passes(d, 10, ltot, roff, poff)
        slots, passes = pm.wave_cursor_walk(T, lst, pitch, d, lo, hi, p_lo, p_hi)
        reads = passes * pm.PASS + idx16[slots].astype(np.int64)
        assert slots.max() < n and np.all(np.diff(reads) > 0) and np.all((gs[reads] >> 10) == d)
