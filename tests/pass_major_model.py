"""Host-side model of the PASS-MAJOR layout of the range-ranked route (csrc/kernels/pass_major.inc.hip), round 4 form
-- every index its producer writes and its consumers read, with bounds asserted -- so that the layout's arithmetic is
checked on the CPU before (and beside) any GPU run.  (Round 2's abandoned 16-bit-index experiment ended in a GPU
memory-access fault whose code was not kept; since then a layout is modelled first.)

Layout.  The reads of a call are cut into PASSES of 8 192 consecutive reads.  k_pm_prepare_sort sorts every pass by
position range (digit = clamped global start >> shift, <= 256 ranges), stably, and writes the pass's records of range
d -- its SLICE (d, P) -- at a slot that is a multiple of 64: pass P owns the slots [P * stride, (P + 1) * stride),
stride = 8192 + 64 * n_ranges, and the slices follow each other padded to whole groups of 64 slots.  Two 16-bit
streams: keys16[slot] = global start & (2^shift - 1), idx16[slot] = read index - P * 8192.  Two tables laid out
[range][pass] (row pitch a multiple of 4):
    cntp[d][P] = records of the slice rounded up to a multiple of 64      lstw[d][P] = (first slot - P * stride) / 64
                                                                                       | true record count << 16
An exclusive scan over the flattened cntp table gives Tp[d][P]: the PADDED FLAT position of the slice -- range d's
records in read-index order are its slices in pass order, each padded to whole groups.  A group of 64 padded flat
positions (a WAVE-SLOT, g = position / 64) therefore lies in exactly one slice, and a one-word descriptor per
wave-slot says where: desc[g] = pass << 15 | slot group inside the pass << 6 | (records in the group - 1).  The
consumers never search:
    k_pm_offsets  walks a range's wave-slots in any order (LDS histogram of the positions);
    k_pm_walk     walks them in order, sixteen (one chunk) per step, against the per-position quotas, and marks the kept
                  records' read indices (pass * 8192 + idx16);
    k_pm_settle   decides the (chunk, position) groups whose quota ran out inside a chunk.
"""
import numpy as np

PASS = 8192
CHUNK_WS = 16          # wave-slots per chunk (one per wave of the walking workgroup)


def pitch_for(n):
    n_pass = (n + PASS - 1) // PASS
    return (n_pass + 3) & ~3


def stride_for(n_ranges):
    return PASS + 64 * n_ranges


def producer(gstart, shift, n_ranges):
    """-> keys16, idx16 (pitch * stride slots, 0xFFFF where nothing was written), cntp, lstw ([256][pitch] uint32)"""
    n = gstart.size
    pitch = pitch_for(n)
    stride = stride_for(n_ranges)
    keys16 = np.full(pitch * stride, 0xFFFF, np.uint16)
    idx16 = np.full(pitch * stride, 0xFFFF, np.uint16)
    cntp = np.zeros((256, pitch), np.uint32)
    lstw = np.zeros((256, pitch), np.uint32)
    for P in range((n + PASS - 1) // PASS):
        lo, hi = P * PASS, min(n, (P + 1) * PASS)
        d = (gstart[lo:hi] >> shift).astype(np.int64)
        assert d.max() < n_ranges <= 256
        order = np.argsort(d, kind="stable")
        c = np.bincount(d, minlength=256).astype(np.int64)
        pad = (c + 63) // 64 * 64
        first = np.concatenate([[0], np.cumsum(pad)[:-1]])
        assert first[-1] + pad[-1] <= stride
        compact = np.concatenate([[0], np.cumsum(c)[:-1]])
        for dd in np.nonzero(c)[0]:
            src = order[compact[dd]:compact[dd] + c[dd]]
            at = P * stride + first[dd]
            keys16[at:at + c[dd]] = (gstart[lo:hi][src] & ((1 << shift) - 1)).astype(np.uint16)
            idx16[at:at + c[dd]] = src.astype(np.uint16)
        cntp[:, P] = pad.astype(np.uint32)
        assert (first // 64).max() < (1 << 16) and c.max() < (1 << 16)
        lstw[:, P] = ((first // 64) | (c << 16)).astype(np.uint32)
    return keys16, idx16, cntp, lstw


def scan_table(cntp):
    """exclusive scan over the flattened [range][pass] table, total appended (launch_exclusive_scan, write_total)"""
    flat = cntp.reshape(-1).astype(np.uint64)
    return np.concatenate([[0], np.cumsum(flat)]).astype(np.uint32)


def unpack(dsc, stride):
    """-> first slot, records, pass of a wave-slot's descriptor"""
    dsc = int(dsc)
    P = dsc >> 15
    return P * stride + ((dsc >> 6) & 511) * 64, (dsc & 63) + 1, P


def descriptors(Tp, lstw, n, n_ranges):
    """k_pm_descr + k_pm_range_table: one thread per (range, pass) table entry -> the slice's wave-slot descriptors, and
    (row sums of the true counts, scanned) the ranges' TRUE flat starts.  -> desc [G], range_start [257] (true)"""
    pitch = pitch_for(n)
    G = int(Tp[-1]) // 64
    desc = np.full(G, 0xFFFFFFFF, np.uint32)
    true_rows = np.zeros(256, np.int64)
    for d in range(256):
        for P in range(pitch):
            w = int(lstw[d, P])
            cnt, lst64 = w >> 16, w & 0xFFFF
            if cnt == 0:
                continue
            true_rows[d] += cnt
            t = int(Tp[d * pitch + P])
            assert t % 64 == 0
            g = t // 64
            n_ws = (cnt + 63) // 64
            for j in range(n_ws):
                assert P < (1 << 17) and lst64 + j < 512 and g + j < G and desc[g + j] == 0xFFFFFFFF
                desc[g + j] = (P << 15) | ((lst64 + j) << 6) | (min(64, cnt - 64 * j) - 1)
    assert not (desc == 0xFFFFFFFF).any()      # every wave-slot of the padded flat space belongs to a slice
    range_start = np.concatenate([[0], np.cumsum(true_rows)]).astype(np.uint32)
    assert int(range_start[-1]) <= n
    return desc, range_start


def range_wave_slots(Tp, pitch, d):
    lo, hi = int(Tp[d * pitch]), int(Tp[(d + 1) * pitch])
    assert lo % 64 == 0 and hi % 64 == 0
    return lo // 64, (hi - lo) // 64


def offsets(keys16, desc, Tp, pitch, d, shift, stride):
    """k_pm_offsets: the range's positions, histogrammed (order-free)"""
    g0, n_ws = range_wave_slots(Tp, pitch, d)
    hist = np.zeros(1 << shift, np.int64)
    for ws in range(n_ws):
        slot0, nv, _ = unpack(desc[g0 + ws], stride)
        k = keys16[slot0:slot0 + nv]
        assert k.max() < (1 << shift)
        np.add.at(hist, k.astype(np.int64), 1)
    return hist


def walk(keys16, idx16, desc, Tp, pitch, d, quota, stride, n, mask, rng):
    """k_pm_walk for one range.  quota: int array [1 << shift] = S(p).  Chunks of 16 wave-slots in order; inside a chunk
    the draws happen in an arbitrary order (rng permutation: the device's LDS arbitration).  Sets mask[read index] of the
    kept records; -> amb [(chunk, position key, skip)], kept"""
    g0, n_ws = range_wave_slots(Tp, pitch, d)
    q = quota.astype(np.int64).copy()
    amb = []
    kept = 0
    n_chunks = (n_ws + CHUNK_WS - 1) // CHUNK_WS
    for c in range(n_chunks):
        recs = []          # (key, read index)
        for w in range(CHUNK_WS):
            ws = c * CHUNK_WS + w
            if ws >= n_ws:
                continue
            slot0, nv, P = unpack(desc[g0 + ws], stride)
            for lane in range(nv):
                i = int(idx16[slot0 + lane])
                assert i < PASS
                recs.append((int(keys16[slot0 + lane]), P * PASS + i))
        order = rng.permutation(len(recs))
        old = [0] * len(recs)
        for i in order:
            old[i] = q[recs[i][0]]
            q[recs[i][0]] -= 1
        for i, (key, read) in enumerate(recs):
            aft = q[key]
            if old[i] > 0 and aft >= 0:
                assert read < n and not mask[read]
                mask[read] = True
                kept += 1
            if old[i] == 1 and aft < 0:
                amb.append((c, key, int(-aft)))
    return amb, kept


def settle(amb, keys16, idx16, desc, Tp, pitch, d, n, stride, mask):
    """k_pm_settle: a listed (chunk, position) group keeps all of the chunk's records at that position but the last
    `skip` in read-index order"""
    g0, n_ws = range_wave_slots(Tp, pitch, d)
    kept = 0
    for (c, key, skip) in amb:
        members = []
        for w in range(CHUNK_WS):
            ws = c * CHUNK_WS + w
            if ws >= n_ws:
                continue
            slot0, nv, P = unpack(desc[g0 + ws], stride)
            for lane in range(nv):
                if int(keys16[slot0 + lane]) == key:
                    members.append(P * PASS + int(idx16[slot0 + lane]))
        assert len(members) > skip
        for i in members[:len(members) - skip]:
            assert i < n and not mask[i]
            mask[i] = True
            kept += 1
    return kept
