"""ctypes binding of the MI355X quasi-MCP solver (C ABI: include/qmcp_hip.h).

The directory name has a hyphen, so import it with
``importlib.import_module("genome-downsampler_amd")`` (the repo root on sys.path) or through
``__graft_entry__.load_package()``.

There is no CPU fallback anywhere in this package: if the HIP library is missing it raises at
import, and without a GPU every solver call raises :class:`QmcpError`.
"""
import contextlib
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
# (QMCP_HIP_LIB: another build of the same library -- lab/variants/<name>/libqmcp_hip.so -- instead of the product's;
#  lab use only: the host mirror library still links the product's)
HIP_LIB_PATH = os.environ.get("QMCP_HIP_LIB") or os.path.join(LIB_DIR, "libqmcp_hip.so")
HOST_LIB_PATH = os.environ.get("QMCP_HOST_LIB") or os.path.join(LIB_DIR, "libqmcp_host.so")  # (e.g. a sanitizer build of the host mirror)

# every symbol include/qmcp_hip.h declares
ABI_SYMBOLS = (
    "qmcp_hip_abi_version", "qmcp_hip_last_error", "qmcp_hip_device_count", "qmcp_hip_create",
    "qmcp_hip_destroy", "qmcp_hip_solve_host", "qmcp_hip_solve_device", "qmcp_hip_coverage_host",
    "qmcp_hip_filtered_coverage_host", "qmcp_hip_complete_pairs_device",
    "qmcp_hip_complete_pairs_host", "qmcp_hip_amplicon_filter_host", "qmcp_hip_set_profiling",
    "qmcp_hip_kernel_times", "qmcp_hip_filter_solve_host", "qmcp_hip_solve_device_begin",
    "qmcp_hip_solve_end", "qmcp_hip_demand_host", "qmcp_hip_solve_host64", "qmcp_hip_multi_create",
    "qmcp_hip_multi_destroy", "qmcp_hip_multi_solve_host", "qmcp_hip_kept_indices_host",
    "qmcp_hip_default_options", "qmcp_hip_set_options", "qmcp_hip_get_options",
)

QMCP_OK = 0
PATH_UNIFORM, PATH_GENERAL, PATH_NEAR_UNIFORM = 1, 2, 3
KIND_UNIFORM, KIND_LOW_BOTH_SIDES, KIND_HOLE, KIND_ZERO_BOTH_SIDES = 0, 1, 2, 3


# status codes of include/qmcp_hip.h
QMCP_OK, QMCP_EINVAL, QMCP_EREAD, QMCP_ERANGE, QMCP_ENODEVICE, QMCP_EHIP, QMCP_ENOMEM = 0, -1, -2, -3, -4, -5, -6


class QmcpError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"qmcp_hip error {code}: {message}")
        self.code = code


class HostBreakdown(C.Structure):
    """qmcp_hip_host_breakdown (include/qmcp_hip.h)"""
    _fields_ = [("ms_total", C.c_float), ("ms_narrow_h2d", C.c_float), ("ms_solve", C.c_float),
                ("ms_d2h", C.c_float), ("host_threads", C.c_uint32), ("chunks", C.c_uint32),
                ("columns_sent", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [
        ("n_reads", C.c_uint64), ("n_kept", C.c_uint64), ("total_length", C.c_uint64),
        ("n_contigs", C.c_uint32), ("path", C.c_uint32), ("min_span", C.c_uint32),
        ("max_span", C.c_uint32), ("sort_passes", C.c_uint32), ("sweep_stretches", C.c_uint32),
        ("ms_total", C.c_float), ("ms_prepare", C.c_float), ("ms_scan", C.c_float),
        ("ms_sort", C.c_float), ("ms_sweep", C.c_float), ("ms_mark", C.c_float),
        ("ms_h2d", C.c_float), ("ms_d2h", C.c_float), ("columns_sent", C.c_uint32),
        ("spec_boundaries", C.c_uint32), ("spec_mismatches", C.c_uint32),
        ("spec_retry_mismatches", C.c_uint32), ("sweep_blocks_changed", C.c_uint32), ("sweep_blocks", C.c_uint32),
        ("arena_grown_mid_solve", C.c_uint32), ("near_uniform_exceptions", C.c_uint32),
        ("near_uniform_selected", C.c_uint32), ("near_uniform_rounds", C.c_uint32),
        ("near_uniform_giveup", C.c_uint32),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class Options(C.Structure):
    """qmcp_hip_options (include/qmcp_hip.h): 0 = the library chooses"""
    _fields_ = [
        ("struct_size", C.c_uint32), ("pass_major", C.c_int32), ("sweep", C.c_int32), ("cut_points", C.c_int32),
        ("speculation", C.c_int32), ("speculation_run_in", C.c_uint32), ("near_uniform", C.c_int32),
        ("near_uniform_rounds", C.c_uint32), ("near_uniform_min_depth", C.c_float), ("near_uniform_debug", C.c_int32),
        ("force_sort_route", C.c_int32), ("keep_expand", C.c_int32), ("mixed_sweep_in_lds", C.c_int32),
        ("rank_min_reads", C.c_uint32), ("host_threads", C.c_uint32), ("copy_streams", C.c_uint32),
        ("host_both_columns", C.c_int32),
    ]


SWEEP_AUTO, SWEEP_FAST, SWEEP_GENERAL, SWEEP_EVENTS = 0, 1, 2, 3
_SWEEP_NAMES = {None: 0, "auto": 0, "fast": 1, "gen": 2, "general": 2, "ev": 3, "events": 3}


if not os.path.exists(HIP_LIB_PATH):
    raise ImportError(
        f"{HIP_LIB_PATH} is missing: build it with `make lib` (or __graft_entry__.build()); "
        "this package has no CPU fallback")

_hip = C.CDLL(HIP_LIB_PATH, mode=C.RTLD_GLOBAL)
_host = C.CDLL(HOST_LIB_PATH) if os.path.exists(HOST_LIB_PATH) else None

_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_hip.qmcp_hip_last_error.restype = C.c_char_p
_hip.qmcp_hip_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
_hip.qmcp_hip_destroy.argtypes = [C.c_void_p]
_hip.qmcp_hip_destroy.restype = None
_hip.qmcp_hip_default_options.argtypes = [C.POINTER(Options)]
_hip.qmcp_hip_default_options.restype = None
_hip.qmcp_hip_set_options.argtypes = [C.c_void_p, C.POINTER(Options)]
_hip.qmcp_hip_get_options.argtypes = [C.c_void_p, C.POINTER(Options)]
_hip.qmcp_hip_solve_host.argtypes = [C.c_void_p, _u32p, _u32p, C.c_uint64, _u64p, _u32p,
                                     C.c_uint32, C.c_uint32, _u64p, C.POINTER(Stats)]
_hip.qmcp_hip_solve_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, _u64p, _u32p,
                                       C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                       C.POINTER(Stats)]
_hip.qmcp_hip_solve_device_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, _u64p, _u32p,
                                             C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
_hip.qmcp_hip_solve_end.argtypes = [C.c_void_p, C.POINTER(Stats)]
_hip.qmcp_hip_solve_host64.argtypes = [C.c_void_p, _u64p, _u64p, C.c_uint64, _u64p, _u32p, C.c_uint32,
                                       C.c_uint32, _u64p, C.POINTER(Stats), C.POINTER(HostBreakdown)]
_hip.qmcp_hip_multi_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
_hip.qmcp_hip_multi_destroy.argtypes = [C.c_void_p]
_hip.qmcp_hip_multi_destroy.restype = None
_hip.qmcp_hip_multi_solve_host.argtypes = [C.c_void_p, _u32p, _u32p, C.c_uint64, _u64p, _u32p, C.c_uint32,
                                           C.c_uint32, _u64p, C.POINTER(Stats), C.POINTER(C.c_int)]
_hip.qmcp_hip_demand_host.argtypes = [C.c_void_p, _u32p, _u32p, C.c_uint64, C.c_uint32, C.c_uint32,
                                      C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
_hip.qmcp_hip_coverage_host.argtypes = [C.c_void_p, _u32p, _u32p, C.c_uint64, _u64p, _u32p,
                                        C.c_uint32, _u32p]
_hip.qmcp_hip_filtered_coverage_host.argtypes = [C.c_void_p, _u32p, _u32p, C.c_uint64, _u64p, _u32p,
                                                 C.c_uint32, _u64p, _u32p]
_hip.qmcp_hip_complete_pairs_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
_hip.qmcp_hip_complete_pairs_host.argtypes = [C.c_void_p, _u64p, C.c_uint64]
_hip.qmcp_hip_amplicon_filter_host.argtypes = [C.c_void_p, _u32p, _u32p, _u32p, _u32p, C.c_uint64,
                                               _u32p, _u32p, C.c_uint32, C.c_uint32, C.c_uint32,
                                               _u64p]
_hip.qmcp_hip_filter_solve_host.argtypes = [C.c_void_p, _u32p, _u32p, _u32p, _u32p, C.c_uint64, _u32p,
                                            _u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                            C.c_uint32, C.c_int, _u64p, _u64p, C.POINTER(Stats)]
_hip.qmcp_hip_set_profiling.argtypes = [C.c_void_p, C.c_int]
_hip.qmcp_hip_kernel_times.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
if _host is not None:
    _host.qmcp_host_reads_gen.argtypes = [C.c_uint32, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32,
                                          _u32p, _u32p, _u32p]
    _host.qmcp_host_reads_gen_aos.argtypes = _host.qmcp_host_reads_gen.argtypes
    _host.qmcp_host_solver_names.argtypes = [C.c_char_p, C.c_size_t]
    _host.qmcp_host_solve.argtypes = [C.c_char_p, _u32p, _u32p, C.c_uint64, C.c_uint32, C.c_uint32,
                                      C.c_int, _u64p]
    _host.qmcp_host_solve.restype = C.c_int64
    _host.qmcp_host_plugin_solve_timed.argtypes = [C.c_char_p, _u32p, _u32p, C.c_uint64, C.c_uint32,
                                                   C.c_uint32, _u64p, C.POINTER(C.c_float)]
    _host.qmcp_host_plugin_solve_timed.restype = C.c_int64
    _u16p, _u8p = C.POINTER(C.c_uint16), C.POINTER(C.c_uint8)
    _host.qmcp_host_write_synthetic_bam.argtypes = [C.c_char_p, C.c_uint32, C.c_uint64, _u32p, _u16p, _u32p,
                                                    _u8p, _u32p, _u32p, _u32p, _u32p]
    _host.qmcp_host_read_bam.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_uint32, C.c_uint32,
                                         C.c_uint64, _u64p, _u32p, _u32p, _u32p, _u32p, _u8p, C.c_uint64, _u64p,
                                         _u64p, _u32p]
    _host.qmcp_host_read_bam.restype = C.c_int64
    _host.qmcp_host_downsample_bam.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32,
                                               C.c_uint32, C.c_uint32]
    _host.qmcp_host_downsample_bam.restype = C.c_int64
    _host.qmcp_host_check_bam.argtypes = [C.c_char_p, _u64p, C.c_char_p, C.c_size_t]
    _host.qmcp_host_bamapi_probe.argtypes = [_u32p, _u32p, C.c_uint64, C.c_uint32, C.c_int, _u64p,
                                             C.c_uint64, _u32p, _u32p, _u64p]
    _host.qmcp_host_bamapi_probe.restype = C.c_int64
    _host.qmcp_host_amplicons_from_files.argtypes = [C.c_char_p, C.c_char_p, _u32p, _u32p, C.c_size_t]


def _p32(a):
    return a.ctypes.data_as(_u32p) if a is not None else None


def _p64(a):
    return a.ctypes.data_as(_u64p) if a is not None else None


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _check(rc):
    if rc != QMCP_OK:
        raise QmcpError(rc, _hip.qmcp_hip_last_error().decode())


def abi_version():
    return _hip.qmcp_hip_abi_version()


def device_count():
    return _hip.qmcp_hip_device_count()


def exported_symbols():
    return [s for s in ABI_SYMBOLS if hasattr(_hip, s)]


def mask_words(n_reads):
    return (int(n_reads) + 63) // 64


def mask_to_indices(mask, n_reads):
    """ascending kept ReadIndex list == the reference's Solution vector"""
    bits = np.unpackbits(np.ascontiguousarray(mask).view(np.uint8), bitorder="little")[:n_reads]
    return np.flatnonzero(bits).astype(np.uint64)


def indices_to_mask(indices, n_reads):
    bits = np.zeros(mask_words(n_reads) * 64, dtype=np.uint8)
    bits[np.asarray(indices, dtype=np.int64)] = 1
    return np.packbits(bits, bitorder="little").view(np.uint64).copy()


def _contig_tables(n_reads, contig_read_offsets, contig_lengths):
    if contig_read_offsets is None:
        lengths = np.atleast_1d(np.asarray(contig_lengths, dtype=np.uint32))
        assert lengths.size == 1, "contig_read_offsets required for several contigs"
        offs = np.array([0, n_reads], dtype=np.uint64)
    else:
        offs = np.ascontiguousarray(contig_read_offsets, dtype=np.uint64)
        lengths = np.ascontiguousarray(contig_lengths, dtype=np.uint32)
    return offs, lengths


class Solver:
    """One solver context == one reference solver instance (created once, solved many times)."""

    def __init__(self, device=0):
        self._ctx = C.c_void_p()
        _check(_hip.qmcp_hip_create(int(device), C.byref(self._ctx)))
        self.device = device
        self.last_stats = None

    def close(self):
        if self._ctx:
            _hip.qmcp_hip_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_profiling(self, enabled):
        """bracket kernel launches with HIP events on the solver stream (resets the totals):
        True / 1 every kernel, 2 only the selection sweep, False / 0 off"""
        _check(_hip.qmcp_hip_set_profiling(self._ctx, 2 if enabled == 2 else int(bool(enabled))))

    def kernel_times(self):
        """{kernel: (launches, total_ms)} accumulated since set_profiling(True)"""
        buf = C.create_string_buffer(8192)
        n = _hip.qmcp_hip_kernel_times(self._ctx, buf, len(buf))
        if n < 0:
            _check(n)
        out = {}
        for line in buf.value.decode().splitlines():
            name, launches, ms = line.split("\t")
            out[name] = (int(launches), float(ms))
        return out

    def solve(self, starts, ends, contig_lengths, max_coverage, contig_read_offsets=None):
        """host arrays in, host keep bitmask (np.uint64 words) out"""
        starts, ends = _u32(starts), _u32(ends)
        n = starts.size
        offs, lengths = _contig_tables(n, contig_read_offsets, contig_lengths)
        mask = np.zeros(max(mask_words(n), 1), dtype=np.uint64)
        st = Stats()
        _check(_hip.qmcp_hip_solve_host(self._ctx, _p32(starts), _p32(ends), n, _p64(offs),
                                        _p32(lengths), lengths.size, int(max_coverage), _p64(mask),
                                        C.byref(st)))
        self.last_stats = st
        return mask[:mask_words(n)]

    def solve64(self, start_inds, end_inds, contig_lengths, max_coverage, contig_read_offsets=None):
        """the reference's own size_t columns in (qmcp_hip_solve_host64: narrowed inside the library),
        host keep bitmask out; self.last_breakdown tells how (threads, chunks, columns sent)"""
        s64 = np.ascontiguousarray(start_inds, dtype=np.uint64)
        e64 = np.ascontiguousarray(end_inds, dtype=np.uint64)
        n = s64.size
        offs, lengths = _contig_tables(n, contig_read_offsets, contig_lengths)
        mask = np.zeros(max(mask_words(n), 1), dtype=np.uint64)
        st, bd = Stats(), HostBreakdown()
        _check(_hip.qmcp_hip_solve_host64(self._ctx, _p64(s64), _p64(e64), n, _p64(offs), _p32(lengths),
                                          lengths.size, int(max_coverage), _p64(mask), C.byref(st), C.byref(bd)))
        self.last_stats, self.last_breakdown = st, bd
        return mask[:mask_words(n)]

    def solve_device(self, d_starts, d_ends, n_reads, contig_lengths, max_coverage, d_mask,
                     contig_read_offsets=None, stream=0):
        """device pointers (ints) in; the mask is written to d_mask (device pointer)"""
        offs, lengths = _contig_tables(n_reads, contig_read_offsets, contig_lengths)
        st = Stats()
        _check(_hip.qmcp_hip_solve_device(self._ctx, C.c_void_p(d_starts), C.c_void_p(d_ends),
                                          int(n_reads), _p64(offs), _p32(lengths), lengths.size,
                                          int(max_coverage), C.c_void_p(d_mask),
                                          C.c_void_p(stream), C.byref(st)))
        self.last_stats = st
        return st

    def get_options(self):
        o = Options()
        _check(_hip.qmcp_hip_get_options(self._ctx, C.byref(o)))
        return o

    def set_options(self, options=None, **fields):
        """qmcp_hip_set_options: `options` (an Options; default: the library's defaults) with `fields` set on top --
        e.g. set_options(sweep="ev", cut_points=1); sweep takes "fast" | "gen" | "ev" or a SWEEP_* number"""
        o = Options()
        if options is not None:
            C.memmove(C.byref(o), C.byref(options), C.sizeof(Options))
        else:
            _hip.qmcp_hip_default_options(C.byref(o))
        for k, v in fields.items():
            if k == "sweep" and not isinstance(v, int):
                v = _SWEEP_NAMES[v]
            if not hasattr(o, k) or k == "struct_size":
                raise AttributeError(f"qmcp_hip_options has no field {k!r}")
            setattr(o, k, v)
        o.struct_size = C.sizeof(Options)
        _check(_hip.qmcp_hip_set_options(self._ctx, C.byref(o)))

    @contextlib.contextmanager
    def options(self, **fields):
        """with solver.options(sweep="ev"): ...   -- the fields on top of the current options, restored afterwards"""
        old = self.get_options()
        self.set_options(old, **fields)
        try:
            yield self
        finally:
            self.set_options(old)

    def solve_device_begin(self, d_starts, d_ends, n_reads, contig_lengths, max_coverage, d_mask,
                           contig_read_offsets=None, stream=0):
        """enqueue a device-resident solve and return without waiting for it (one pending solve per
        Solver; pipeline with a second Solver on the same device); solve_end() collects it"""
        offs, lengths = _contig_tables(n_reads, contig_read_offsets, contig_lengths)
        _check(_hip.qmcp_hip_solve_device_begin(self._ctx, C.c_void_p(d_starts), C.c_void_p(d_ends),
                                                int(n_reads), _p64(offs), _p32(lengths), lengths.size,
                                                int(max_coverage), C.c_void_p(d_mask), C.c_void_p(stream)))

    def solve_end(self):
        st = Stats()
        _check(_hip.qmcp_hip_solve_end(self._ctx, C.byref(st)))
        self.last_stats = st
        return st

    def coverage(self, starts, ends, contig_lengths, contig_read_offsets=None, keep_mask=None):
        starts, ends = _u32(starts), _u32(ends)
        n = starts.size
        offs, lengths = _contig_tables(n, contig_read_offsets, contig_lengths)
        cov = np.zeros(max(int(lengths.sum()), 1), dtype=np.uint32)
        if keep_mask is None:
            _check(_hip.qmcp_hip_coverage_host(self._ctx, _p32(starts), _p32(ends), n, _p64(offs),
                                               _p32(lengths), lengths.size, _p32(cov)))
        else:
            km = np.ascontiguousarray(keep_mask, dtype=np.uint64)
            _check(_hip.qmcp_hip_filtered_coverage_host(self._ctx, _p32(starts), _p32(ends), n,
                                                        _p64(offs), _p32(lengths), lengths.size,
                                                        _p64(km), _p32(cov)))
        return cov[:int(lengths.sum())]

    def demand(self, starts, ends, ref_genome_length, max_coverage):
        """(b, d) of the reference's flow network for one contig, computed on the device:
        create_b_function / create_demand_function, quasi_mcp_cpu_max_flow_solver.cpp:58-87"""
        s, e = _u32(starts), _u32(ends)
        b = np.zeros(int(ref_genome_length) + 1, np.int32)
        d = np.zeros(int(ref_genome_length) + 1, np.int32)
        _check(_hip.qmcp_hip_demand_host(self._ctx, _p32(s), _p32(e), s.size, int(ref_genome_length),
                                         int(max_coverage), b.ctypes.data_as(C.POINTER(C.c_int32)),
                                         d.ctypes.data_as(C.POINTER(C.c_int32))))
        return b, d

    def complete_pairs(self, mask, n_reads):
        out = np.ascontiguousarray(mask, dtype=np.uint64).copy()
        _check(_hip.qmcp_hip_complete_pairs_host(self._ctx, _p64(out), int(n_reads)))
        return out

    def complete_pairs_device(self, d_mask, n_reads, stream=0):
        _check(_hip.qmcp_hip_complete_pairs_device(self._ctx, C.c_void_p(d_mask), int(n_reads),
                                                   C.c_void_p(stream)))

    def filter_solve(self, starts, ends, ref_genome_length, max_coverage, amp_starts=None,
                     amp_ends=None, seq_lengths=None, qualities=None, min_length=0, min_mapq=0,
                     complete_pairs=False):
        """FILTER -> compaction -> solve -> (find_pairs) in one device-resident call; returns
        (keep mask over the ORIGINAL read indices, number of pairs the pre-pass dropped)"""
        starts, ends = _u32(starts), _u32(ends)
        n = starts.size
        a0 = _u32(amp_starts) if amp_starts is not None else None
        a1 = _u32(amp_ends) if amp_ends is not None else None
        sl = _u32(seq_lengths) if seq_lengths is not None else None
        q = _u32(qualities) if qualities is not None else None
        mask = np.zeros(max(mask_words(n), 1), dtype=np.uint64)
        dropped = C.c_uint64(0)
        st = Stats()
        _check(_hip.qmcp_hip_filter_solve_host(self._ctx, _p32(starts), _p32(ends), _p32(sl), _p32(q), n,
                                               _p32(a0), _p32(a1), 0 if a0 is None else a0.size,
                                               int(min_length), int(min_mapq), int(ref_genome_length),
                                               int(max_coverage), int(bool(complete_pairs)),
                                               _p64(mask), C.byref(dropped), C.byref(st)))
        self.last_stats = st
        return mask[:mask_words(n)], int(dropped.value)

    def amplicon_filter(self, starts, ends, amp_starts, amp_ends, seq_lengths=None, qualities=None,
                        min_length=0, min_mapq=0):
        starts, ends = _u32(starts), _u32(ends)
        a0, a1 = _u32(amp_starts), _u32(amp_ends)
        sl = _u32(seq_lengths) if seq_lengths is not None else None
        q = _u32(qualities) if qualities is not None else None
        n = starts.size
        out = np.zeros(max(mask_words(n // 2), 1), dtype=np.uint64)
        _check(_hip.qmcp_hip_amplicon_filter_host(self._ctx, _p32(starts), _p32(ends), _p32(sl),
                                                  _p32(q), n, _p32(a0), _p32(a1), a0.size,
                                                  int(min_length), int(min_mapq), _p64(out)))
        return out[:mask_words(n // 2)]


# ---------------------------------------------------------------- host mirror (libqmcp_host.so)
class MultiSolver:
    """several devices behind one call (qmcp_hip_multi_*): contigs dealt to the devices by cost, one
    context and one host thread per entry of `devices` (a device may be named more than once)"""

    def __init__(self, devices):
        self.devices = [int(d) for d in devices]
        arr = (C.c_int * len(self.devices))(*self.devices)
        self._m = C.c_void_p()
        _check(_hip.qmcp_hip_multi_create(arr, len(self.devices), C.byref(self._m)))
        self.last_stats = []
        self.last_assignment = None

    def close(self):
        if getattr(self, "_m", None):
            _hip.qmcp_hip_multi_destroy(self._m)
            self._m = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def solve(self, starts, ends, contig_lengths, max_coverage, contig_read_offsets=None):
        starts, ends = _u32(starts), _u32(ends)
        offs, lengths = _contig_tables(starts.size, contig_read_offsets, contig_lengths)
        mask = np.zeros(mask_words(starts.size), dtype=np.uint64)
        stats = (Stats * len(self.devices))()
        where = (C.c_int * lengths.size)()
        _check(_hip.qmcp_hip_multi_solve_host(self._m, _p32(starts), _p32(ends), starts.size, _p64(offs),
                                              _p32(lengths), lengths.size, int(max_coverage), _p64(mask),
                                              stats, where))
        self.last_stats = list(stats)
        self.last_assignment = list(where)
        return mask


def _need_host():
    if _host is None:
        raise ImportError(f"{HOST_LIB_PATH} is missing: build it with `make lib`")


def reads_gen(kind, pairs, genome_length, read_length=150, seed=12345, with_qualities=False,
              aos=False):
    """reads-gen restatement (libs/reads-gen): returns (starts, ends[, qualities]) as uint32"""
    _need_host()
    n = 2 * int(pairs)
    s = np.empty(n, dtype=np.uint32)
    e = np.empty(n, dtype=np.uint32)
    q = np.empty(n, dtype=np.uint32) if with_qualities else None
    fn = _host.qmcp_host_reads_gen_aos if aos else _host.qmcp_host_reads_gen
    rc = fn(int(seed), int(kind), int(pairs), int(genome_length), int(read_length), _p32(s), _p32(e),
            _p32(q))
    if rc != 0:
        raise ValueError(f"reads_gen failed ({rc})")
    return (s, e, q) if with_qualities else (s, e)


def amplicons_from_files(bed_path, tsv_path=None):
    """BED (+ optional TSV) -> (amp_starts, amp_ends), as BamApi::set_amplicon_filter builds them"""
    _need_host()
    cap = 1 << 16
    a0 = np.empty(cap, dtype=np.uint32)
    a1 = np.empty(cap, dtype=np.uint32)
    n = _host.qmcp_host_amplicons_from_files(str(bed_path).encode(),
                                             str(tsv_path).encode() if tsv_path else None,
                                             _p32(a0), _p32(a1), cap)
    if n < 0:
        raise OSError(f"cannot build amplicon set from {bed_path} / {tsv_path} ({n})")
    return a0[:n].copy(), a1[:n].copy()


def bamapi_probe(starts, ends, ref_genome_length, ids, layout=0):
    """in-memory BamApi of the host mirror: (find_input_cover, find_filtered_cover(ids), find_pairs(ids))"""
    _need_host()
    starts, ends = _u32(starts), _u32(ends)
    ids = np.ascontiguousarray(ids, dtype=np.uint64)
    cin = np.zeros(max(ref_genome_length, 1), dtype=np.uint32)
    cout = np.zeros(max(ref_genome_length, 1), dtype=np.uint32)
    paired = np.zeros(max(starts.size, 1), dtype=np.uint64)
    n = _host.qmcp_host_bamapi_probe(_p32(starts), _p32(ends), starts.size, int(ref_genome_length),
                                     int(layout), _p64(ids), ids.size, _p32(cin), _p32(cout),
                                     _p64(paired))
    if n < 0:
        raise RuntimeError("bamapi probe failed")
    return cin[:ref_genome_length], cout[:ref_genome_length], paired[:n].copy()


def solver_names():
    _need_host()
    buf = C.create_string_buffer(4096)
    n = _host.qmcp_host_solver_names(buf, len(buf))
    if n < 0:
        raise RuntimeError("solver name buffer too small")
    return [x for x in buf.value.decode().split("\n") if x]


def host_solve(solver_name, starts, ends, ref_genome_length, max_coverage, with_pairs=False):
    """SolverManager::get(name).solve(M, BamApi) through the C++ adapter; ascending ReadIndex"""
    _need_host()
    starts, ends = _u32(starts), _u32(ends)
    kept = np.empty(max(starts.size, 1), dtype=np.uint64)
    n = _host.qmcp_host_solve(solver_name.encode(), _p32(starts), _p32(ends), starts.size,
                              int(ref_genome_length), int(max_coverage), int(bool(with_pairs)),
                              _p64(kept))
    if n < 0:
        raise KeyError(solver_name)
    return kept[:n].copy()


def plugin_solve_timed(solver_name, starts, ends, ref_genome_length, max_coverage):
    """the reference's "solve took" span (src/app.cpp:132-139) at the plugin boundary: a BamApi holding
    SOAPairedReads (size_t columns) -> Solver::solve -> Solution.  Returns (kept ReadIndex array, dict of
    host wall-clock milliseconds)"""
    _need_host()
    starts, ends = _u32(starts), _u32(ends)
    kept = np.empty(max(starts.size, 1), dtype=np.uint64)
    t = (C.c_float * 9)()
    n = _host.qmcp_host_plugin_solve_timed(solver_name.encode(), _p32(starts), _p32(ends), starts.size,
                                           int(ref_genome_length), int(max_coverage), _p64(kept), t)
    if n < 0:
        raise KeyError(solver_name)
    names = ("solve_call_ms", "library_ms", "narrow_h2d_ms", "device_solve_ms", "d2h_ms", "expand_ms")
    out = {k: round(float(t[i]), 3) for i, k in enumerate(names)}
    out["host_threads"], out["chunks"], out["columns_sent"] = int(t[6]), int(t[7]), int(t[8])
    return kept[:n].copy(), out


def write_synthetic_bam(path, ref_length, names, flags, pos, mapq, clip_front, match, deletion, match2):
    """single-reference BAM whose record i has qname "q<names[i]>" and CIGAR <clip>S<match>M<del>D<match2>M
    (zero-length parts left out); written by the in-repo BGZF writer (tests only)"""
    _need_host()
    n = len(names)
    a = lambda x, t: np.ascontiguousarray(x, dtype=t)
    names, pos, clip_front, match, deletion, match2 = (a(x, np.uint32) for x in (names, pos, clip_front, match, deletion, match2))
    flags, mapq = a(flags, np.uint16), a(mapq, np.uint8)
    rc = _host.qmcp_host_write_synthetic_bam(str(path).encode(), int(ref_length), n, _p32(names),
                                             flags.ctypes.data_as(C.POINTER(C.c_uint16)), _p32(pos),
                                             mapq.ctypes.data_as(C.POINTER(C.c_uint8)), _p32(clip_front),
                                             _p32(match), _p32(deletion), _p32(match2))
    if rc != 0:
        raise OSError(f"cannot write {path}")


def read_bam(path, bed=None, tsv=None, amplicon_mode=0, min_length=0, min_mapq=0, capacity=1 << 24):
    """BamApi(path, config).get_paired_reads_soa() of the host mirror: dict of columns + filtered-out ids"""
    _need_host()
    ids = np.empty(capacity, np.uint64)
    cols = {k: np.empty(capacity, np.uint32) for k in ("starts", "ends", "qualities", "seq_lengths")}
    first = np.empty(capacity, np.uint8)
    filt = np.empty(capacity, np.uint64)
    nf, ref = C.c_uint64(0), C.c_uint32(0)
    n = _host.qmcp_host_read_bam(str(path).encode(), str(bed).encode() if bed else None,
                                 str(tsv).encode() if tsv else None, int(amplicon_mode), int(min_length),
                                 int(min_mapq), capacity, _p64(ids), _p32(cols["starts"]), _p32(cols["ends"]),
                                 _p32(cols["qualities"]), _p32(cols["seq_lengths"]),
                                 first.ctypes.data_as(C.POINTER(C.c_uint8)), capacity, _p64(filt), C.byref(nf),
                                 C.byref(ref))
    if n < 0:
        raise OSError(f"read_bam({path}) failed ({n})")
    out = {k: v[:n].copy() for k, v in cols.items()}
    out.update(bam_ids=ids[:n].copy(), is_first=first[:n].astype(bool), filtered_out=filt[:nf.value].copy(),
               ref_genome_length=int(ref.value))
    return out


def check_bam(path):
    """read_bam's own verdict on a file (no exit-on-error as in BamApi): (True, reads imported, "") or
    (False, 0, the reader's message)"""
    _need_host()
    n = C.c_uint64(0)
    buf = C.create_string_buffer(512)
    rc = _host.qmcp_host_check_bam(str(path).encode(), C.byref(n), buf, 512)
    return rc == 0, int(n.value) if rc == 0 else 0, buf.value.decode(errors="replace")


def copy_records(in_path, out_path, ids):
    """BamApi::write_bam: the header and the records with the given running ids, to a BAM (".bam") or to SAM text"""
    _need_host()
    ids = np.ascontiguousarray(ids, dtype=np.uint64)
    err = C.create_string_buffer(512)
    _host.qmcp_host_copy_records.restype = C.c_int64
    _host.qmcp_host_copy_records.argtypes = [C.c_char_p, C.c_char_p, _u64p, C.c_uint64, C.c_char_p, C.c_size_t]
    n = _host.qmcp_host_copy_records(str(in_path).encode(), str(out_path).encode(), _p64(ids), ids.size, err, 512)
    if n < 0:
        raise OSError(err.value.decode())
    return int(n)


def downsample_bam(solver_name, in_path, out_path, max_coverage, filtered_path=None, min_length=0, min_mapq=0):
    """BamApi(in) -> solve -> find_pairs -> write_paired_reads(out): App::execute's file-to-file flow"""
    _need_host()
    n = _host.qmcp_host_downsample_bam(solver_name.encode(), str(in_path).encode(), str(out_path).encode(),
                                       str(filtered_path).encode() if filtered_path else None,
                                       int(max_coverage), int(min_length), int(min_mapq))
    if n < 0:
        raise KeyError(solver_name)
    return int(n)
