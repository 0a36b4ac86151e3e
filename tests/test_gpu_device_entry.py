"""Device-resident entry point (qmcp_hip_solve_device) driven the way bench.py drives it: torch
owns the buffers, the solver is ordered after the caller's stream."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_solve_device_after_producer_stream(pkg, oracle):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("gpu test without a GPU")
    dev = torch.device("cuda", 0)
    s, e = pkg.reads_gen(pkg.KIND_LOW_BOTH_SIDES, 300_000, 50_000)
    want = oracle.solve(s, e, 50_000, 40)
    words = pkg.mask_words(s.size)
    with pkg.Solver(0) as sv:
        for use_side_stream in (False, True):
            stream = torch.cuda.Stream(dev) if use_side_stream else torch.cuda.current_stream(dev)
            with torch.cuda.stream(stream):
                # producer work on the caller's stream: the reads arrive through a chain of device
                # ops (copy, arithmetic round trip) the solve must wait for
                d_s = torch.from_numpy(s.view(np.int32)).to(dev, non_blocking=True)
                d_e = torch.from_numpy(e.view(np.int32)).to(dev, non_blocking=True)
                d_s = (d_s + 7) - 7
                d_e = (d_e * 3) // 3
                d_mask = torch.full((words,), -1, dtype=torch.int64, device=dev)
                st = sv.solve_device(d_s.data_ptr(), d_e.data_ptr(), s.size, 50_000, 40,
                                     d_mask.data_ptr(), stream=stream.cuda_stream)
            got = d_mask.cpu().numpy().view(np.uint64)
            assert np.array_equal(got, want)
            assert st.n_kept == int(np.unpackbits(got.view(np.uint8)).sum())
            # find_pairs on the device mask, in place
            sv.complete_pairs_device(d_mask.data_ptr(), s.size, stream=stream.cuda_stream)
            assert np.array_equal(d_mask.cpu().numpy().view(np.uint64), oracle.find_pairs(want, s.size))


def test_error_paths_leave_context_usable(pkg, solver):
    z = np.zeros(4, np.uint32)
    with pytest.raises(pkg.QmcpError) as ei:  # offsets do not end at n_reads
        solver.solve(z, z, np.array([10, 10], np.uint32), 1, contig_read_offsets=np.array([0, 2, 3], np.uint64))
    assert ei.value.code == -1
    with pytest.raises(pkg.QmcpError) as ei:  # mixed spans beyond what the event sweep's rings are sized for
        solver.solve([0, 5], [1 << 24, 6], (1 << 24) + 10, 1)
    assert ei.value.code == -3
    # (spans beyond the LDS rings are fine: rings in global memory)
    assert pkg.mask_to_indices(solver.solve([0, 5], [20000, 6], 30000, 1), 2).tolist() == [0]
    assert solver.solve([0, 5], [9, 6], 10, 1).size == 1


def test_two_phase_entry_and_two_contexts_in_flight(pkg, oracle):
    """qmcp_hip_solve_device_begin / _end: one pending solve per context, every other entry point of the
    context refused meanwhile; two contexts keep two solves in flight and agree with the blocking call"""
    import torch
    s, e = pkg.reads_gen(pkg.KIND_UNIFORM, 400_000, 60_000, seed=77)
    d_s = torch.from_numpy(s.view(np.int32)).cuda()
    d_e = torch.from_numpy(e.view(np.int32)).cuda()
    words = pkg.mask_words(s.size)
    masks = [torch.zeros(words, dtype=torch.int64, device="cuda") for _ in range(2)]
    want = oracle.solve(s, e, 60_000, 50)
    with pkg.Solver(0) as a, pkg.Solver(0) as b:
        a.solve_device_begin(d_s.data_ptr(), d_e.data_ptr(), s.size, 60_000, 50, masks[0].data_ptr())
        b.solve_device_begin(d_s.data_ptr(), d_e.data_ptr(), s.size, 60_000, 50, masks[1].data_ptr())
        with pytest.raises(pkg.QmcpError):       # a second begin on a busy context
            a.solve_device_begin(d_s.data_ptr(), d_e.data_ptr(), s.size, 60_000, 50, masks[0].data_ptr())
        with pytest.raises(pkg.QmcpError):       # ... and any other entry point of it
            a.coverage(s, e, 60_000)
        st_a, st_b = a.solve_end(), b.solve_end()
        with pytest.raises(pkg.QmcpError):       # nothing pending any more
            a.solve_end()
        assert st_a.n_kept == st_b.n_kept == int(np.unpackbits(want.view(np.uint8)).sum())
        for m in masks:
            assert np.array_equal(m.cpu().numpy().view(np.uint64), want)
        # the context is usable again
        assert np.array_equal(a.solve(s, e, 60_000, 50), want)


def test_two_phase_entry_over_batches_of_one_shape_and_different_content(pkg, oracle):
    """same-shaped batches back to back through _begin / _end on one context: another read length, mixed
    lengths, an invalid read -- nothing of an earlier batch may leak into a later one (assuming the previous
    batch's span statistics instead of waiting for them was tried: it un-staggers two pipelined solves and
    costs 12 %)"""
    import torch
    n, L = 400_000, 50_000
    rng = np.random.default_rng(31)

    def batch(span_lo, span_hi):
        span = rng.integers(span_lo, span_hi + 1, size=n)
        s = (rng.random(n) * (L - span + 1)).astype(np.int64)
        return s.astype(np.uint32), (s + span - 1).astype(np.uint32)

    batches = [batch(150, 150), batch(150, 150), batch(100, 100), batch(100, 100), batch(90, 140), batch(150, 150)]
    words = pkg.mask_words(n)
    d_mask = torch.zeros(words, dtype=torch.int64, device="cuda")
    with pkg.Solver(0) as sv:
        for k, (s, e) in enumerate(batches):
            d_s = torch.from_numpy(s.view(np.int32)).cuda()
            d_e = torch.from_numpy(e.view(np.int32)).cuda()
            sv.solve_device_begin(d_s.data_ptr(), d_e.data_ptr(), n, L, 20, d_mask.data_ptr())
            st = sv.solve_end()
            want = oracle.solve(s, e, L, 20)
            assert np.array_equal(d_mask.cpu().numpy().view(np.uint64), want), k
            assert st.min_span == int((e - s + 1).min()) and st.max_span == int((e - s + 1).max())
        # same shape again, but one read now ends past the contig
        s, e = batches[-1]
        e = e.copy()
        e[12345] = L + 7
        d_s = torch.from_numpy(s.view(np.int32)).cuda()
        d_e = torch.from_numpy(e.view(np.int32)).cuda()
        with pytest.raises(pkg.QmcpError):
            sv.solve_device_begin(d_s.data_ptr(), d_e.data_ptr(), n, L, 20, d_mask.data_ptr())
            sv.solve_end()
        # and the context still works
        s, e = batches[0]
        assert np.array_equal(sv.solve(s, e, L, 20), oracle.solve(s, e, L, 20))
