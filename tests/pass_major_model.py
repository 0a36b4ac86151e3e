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
wave-slot says where: desc[g] = (first slot / 64) << 6 | (records in the group - 1).  The consumers never search:
    k_pm_offsets  walks a range's wave-slots in any order (LDS histogram of the positions);
    k_pm_walk     walks them in order, sixteen (one chunk) per step, against the per-position quotas; every wave owns
                  a stretch of the kept list (at lo_p + 1024 range + wave * 64 chunks) and appends its kept records'
                  slots there; per chunk it notes {where, how many} at kpw[g0 + 16 range + wave * chunks + chunk];
    k_pm_tiles    builds every pass's 128 mask words from the lists (inv[] maps a pass's slot groups to their
                  wave-slots' notes), every word written once;
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


def descriptors(Tp, lstw, n, n_ranges):
    """k_pm_descr: one thread per (range, pass) table entry -> the slice's wave-slot descriptors, the inverse map (slot
    group -> index of the wave-slot's note: g0 + 16 range + wave * chunks + chunk), and (row sums of the true counts,
    scanned) the ranges' TRUE flat starts.  -> desc [G], inv [pitch * stride / 64], range_start [257] (true), used64
    [pitch] (slot groups every pass uses)"""
    pitch = pitch_for(n)
    stride = stride_for(n_ranges)
    s64 = stride // 64
    G = int(Tp[-1]) // 64
    desc = np.full(G, 0xFFFFFFFF, np.uint32)
    inv = np.full(pitch * s64, 0xFFFFFFFF, np.uint32)
    true_rows = np.zeros(256, np.int64)
    used64 = np.zeros(pitch, np.uint32)
    for d in range(256):
        lo_p, hi_p = int(Tp[d * pitch]), int(Tp[(d + 1) * pitch])
        g0, n_chunks = lo_p // 64, ((hi_p - lo_p) // 64 + CHUNK_WS - 1) // CHUNK_WS
        kb = g0 + 16 * d
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
            used64[P] = max(used64[P], lst64 + n_ws)
            for j in range(n_ws):
                group = P * s64 + lst64 + j
                assert group < (1 << 26) and g + j < G and desc[g + j] == 0xFFFFFFFF and inv[group] == 0xFFFFFFFF
                desc[g + j] = (group << 6) | (min(64, cnt - 64 * j) - 1)
                ws = g + j - g0
                inv[group] = kb + (ws % CHUNK_WS) * n_chunks + ws // CHUNK_WS
    assert not (desc == 0xFFFFFFFF).any()      # every wave-slot of the padded flat space belongs to a slice
    range_start = np.concatenate([[0], np.cumsum(true_rows)]).astype(np.uint32)
    assert int(range_start[-1]) <= n
    return desc, inv, range_start, used64


def range_wave_slots(Tp, pitch, d):
    lo, hi = int(Tp[d * pitch]), int(Tp[(d + 1) * pitch])
    assert lo % 64 == 0 and hi % 64 == 0
    return lo // 64, (hi - lo) // 64


def offsets(keys16, desc, Tp, pitch, d, shift):
    """k_pm_offsets: the range's positions, histogrammed (order-free)"""
    g0, n_ws = range_wave_slots(Tp, pitch, d)
    hist = np.zeros(1 << shift, np.int64)
    for ws in range(n_ws):
        dsc = int(desc[g0 + ws])
        slot0, nv = (dsc >> 6) * 64, (dsc & 63) + 1
        k = keys16[slot0:slot0 + nv]
        assert k.max() < (1 << shift)
        np.add.at(hist, k.astype(np.int64), 1)
    return hist


def walk(keys16, desc, Tp, pitch, d, quota, rng):
    """k_pm_walk for one range.  quota: int array [1 << shift] = S(p).  Chunks of 16 wave-slots in order; inside a chunk
    the draws happen in an arbitrary order (rng permutation: the device's LDS arbitration).  Wave w appends its kept
    records' slots at list_base(w) = lo_p + 1024 d + w * 64 * chunks and notes (where, how many) per chunk.
    -> L (dict list position -> slot), kpw (dict note index -> (position, count)), amb [(chunk, position key, skip)],
    kept"""
    g0, n_ws = range_wave_slots(Tp, pitch, d)
    lo_p = g0 * 64
    q = quota.astype(np.int64).copy()
    L, kpw, amb = {}, {}, []
    kept = 0
    n_chunks = (n_ws + CHUNK_WS - 1) // CHUNK_WS
    cur = [0] * CHUNK_WS
    for c in range(n_chunks):
        recs = []          # (wave, lane, key, slot)
        for w in range(CHUNK_WS):
            ws = c * CHUNK_WS + w
            if ws >= n_ws:
                continue
            dsc = int(desc[g0 + ws])
            slot0, nv = (dsc >> 6) * 64, (dsc & 63) + 1
            for lane in range(nv):
                recs.append((w, lane, int(keys16[slot0 + lane]), slot0 + lane))
        order = rng.permutation(len(recs))
        old = [0] * len(recs)
        for i in order:
            old[i] = q[recs[i][2]]
            q[recs[i][2]] -= 1
        per_wave = {}
        for i, (w, lane, key, slot) in enumerate(recs):
            aft = q[key]
            if old[i] > 0 and aft >= 0:
                per_wave.setdefault(w, []).append(slot)
            if old[i] == 1 and aft < 0:
                amb.append((c, key, int(-aft)))
        for w in range(CHUNK_WS):
            ws = c * CHUNK_WS + w
            if ws >= n_ws:
                continue
            ent = per_wave.get(w, [])
            list_base = lo_p + 1024 * d + w * 64 * n_chunks
            pos = list_base + cur[w]
            assert cur[w] + len(ent) <= 64 * n_chunks       # a wave keeps at most what it walks
            note = g0 + 16 * d + w * n_chunks + c
            assert note not in kpw
            kpw[note] = (pos, len(ent))
            for k, slot in enumerate(ent):
                assert pos + k not in L
                L[pos + k] = slot
            cur[w] += len(ent)
            kept += len(ent)
    return L, kpw, amb, kept


def tiles(L, kpw, inv, idx16, used64, n, n_ranges):
    """k_pm_tiles: every pass's mask words from the kept lists"""
    s64 = stride_for(n_ranges) // 64
    mask = np.zeros(n, bool)
    for P in range((n + PASS - 1) // PASS):
        for t in range(int(used64[P])):
            note = int(inv[P * s64 + t])
            assert note != 0xFFFFFFFF
            pos, cnt = kpw[note]
            for e in range(cnt):
                slot = L[pos + e]
                assert slot // (s64 * 64) == P
                i = int(idx16[slot])
                assert i < PASS and P * PASS + i < n and not mask[P * PASS + i]
                mask[P * PASS + i] = True
    return mask


def settle(amb, keys16, idx16, desc, Tp, pitch, d, n, n_ranges, mask):
    """k_pm_settle: a listed (chunk, position) group keeps all of the chunk's records at that position but the last
    `skip` in read-index order"""
    stride = stride_for(n_ranges)
    g0, n_ws = range_wave_slots(Tp, pitch, d)
    kept = 0
    for (c, key, skip) in amb:
        members = []
        for w in range(CHUNK_WS):
            ws = c * CHUNK_WS + w
            if ws >= n_ws:
                continue
            dsc = int(desc[g0 + ws])
            slot0, nv = (dsc >> 6) * 64, (dsc & 63) + 1
            for lane in range(nv):
                if int(keys16[slot0 + lane]) == key:
                    members.append(slot0 + lane)
        assert len(members) > skip
        for slot in members[:len(members) - skip]:
            P = slot // stride
            i = P * PASS + int(idx16[slot])
            assert i < n and not mask[i]
            mask[i] = True
            kept += 1
    return kept
