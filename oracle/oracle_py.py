"""ctypes binding of the CPU oracle (oracle/qmcp_oracle.c).  TEST INFRASTRUCTURE ONLY:
importable from tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py --
never from the product package."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QMCP_ORACLE_LIB") or os.path.join(_HERE, "libqmcp_oracle.so")  # (e.g. a sanitizer build: tools/sanitize_cpu.sh)
if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} missing: run `make oracle`")
_lib = C.CDLL(LIB_PATH)

_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)


class GraphInfo(C.Structure):
    _fields_ = [("n_arcs", C.c_uint64), ("n_terminal_arcs", C.c_uint64),
                ("total_supply", C.c_int64), ("total_demand", C.c_int64), ("arc_fnv", C.c_uint64)]


_lib.qmcp_oracle_reads_fnv.restype = C.c_uint64
_lib.qmcp_oracle_reads_fnv.argtypes = [_u32p, _u32p, _u32p, C.c_uint64]
_lib.qmcp_oracle_mask_fnv.restype = C.c_uint64
_lib.qmcp_oracle_mask_fnv.argtypes = [_u64p, C.c_uint64]
_lib.qmcp_oracle_b_function.argtypes = [_u32p, _u32p, C.c_uint64, C.c_uint32, C.c_uint32, _i32p]
_lib.qmcp_oracle_demand_function.argtypes = [_i32p, C.c_uint32]
_lib.qmcp_oracle_demand_function.restype = None
_lib.qmcp_oracle_graph.argtypes = [_u32p, _u32p, C.c_uint64, C.c_uint32, C.c_uint32,
                                   C.POINTER(GraphInfo), _i64p]
_lib.qmcp_oracle_select.argtypes = [_u32p, _u32p, C.c_uint64, C.c_uint32, C.c_uint32, _u64p,
                                    C.c_uint64]
_lib.qmcp_oracle_solve.argtypes = [_u32p, _u32p, C.c_uint64, _u64p, _u32p, C.c_uint32, C.c_uint32,
                                   _u64p]
_lib.qmcp_oracle_cover.argtypes = [_u32p, _u32p, C.c_uint64, C.c_uint32, _u64p, C.c_uint64, _u32p]
_lib.qmcp_oracle_is_out_cover_valid.argtypes = [_u32p, _u32p, C.c_uint32, C.c_uint32]
_lib.qmcp_oracle_check_flow.argtypes = [_u32p, _u32p, C.c_uint64, C.c_uint32, C.c_uint32, _u64p,
                                        C.c_uint64, _i64p]
_lib.qmcp_oracle_maxflow_value.restype = C.c_int64
_lib.qmcp_oracle_maxflow_value.argtypes = [_u32p, _u32p, C.c_uint64, C.c_uint32, C.c_uint32]
_lib.qmcp_oracle_find_pairs.argtypes = [_u64p, C.c_uint64]
_lib.qmcp_oracle_find_pairs.restype = None
_lib.qmcp_oracle_amplicon_filter.argtypes = [_u32p, _u32p, _u32p, _u32p, C.c_uint64, _u32p, _u32p,
                                             C.c_uint32, C.c_uint32, C.c_uint32, _u64p]
_lib.qmcp_oracle_amplicon_filter.restype = None


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _p32(a):
    return a.ctypes.data_as(_u32p) if a is not None else None


def _p64(a):
    return a.ctypes.data_as(_u64p) if a is not None else None


def mask_words(n):
    return (int(n) + 63) // 64


def reads_fnv(starts, ends, qualities=None):
    s, e = _u32(starts), _u32(ends)
    q = _u32(qualities) if qualities is not None else None
    return int(_lib.qmcp_oracle_reads_fnv(_p32(s), _p32(e), _p32(q), s.size))


def mask_fnv(mask, n_reads):
    m = np.ascontiguousarray(mask, dtype=np.uint64)
    return int(_lib.qmcp_oracle_mask_fnv(_p64(m), int(n_reads)))


def b_function(starts, ends, ref_len, M):
    s, e = _u32(starts), _u32(ends)
    b = np.zeros(ref_len + 1, dtype=np.int32)
    rc = _lib.qmcp_oracle_b_function(_p32(s), _p32(e), s.size, ref_len, M, b.ctypes.data_as(_i32p))
    if rc:
        raise ValueError(f"oracle b_function rc={rc}")
    return b


def demand_function(starts, ends, ref_len, M):
    d = b_function(starts, ends, ref_len, M)
    _lib.qmcp_oracle_demand_function(d.ctypes.data_as(_i32p), ref_len)
    return d


def graph(starts, ends, ref_len, M, want_arcs=False):
    s, e = _u32(starts), _u32(ends)
    info = GraphInfo()
    arcs = None
    if want_arcs:
        arcs = np.zeros(3 * (s.size + 2 * ref_len + 1), dtype=np.int64)
    rc = _lib.qmcp_oracle_graph(_p32(s), _p32(e), s.size, ref_len, M, C.byref(info),
                                arcs.ctypes.data_as(_i64p) if arcs is not None else None)
    if rc:
        raise ValueError(f"oracle graph rc={rc}")
    if want_arcs:
        return info, arcs[:3 * info.n_arcs].reshape(-1, 3)
    return info


def solve(starts, ends, contig_lengths, M, contig_read_offsets=None):
    s, e = _u32(starts), _u32(ends)
    n = s.size
    lengths = np.atleast_1d(np.asarray(contig_lengths, dtype=np.uint32))
    if contig_read_offsets is None:
        offs = np.array([0, n], dtype=np.uint64)
    else:
        offs = np.ascontiguousarray(contig_read_offsets, dtype=np.uint64)
    mask = np.zeros(max(mask_words(n), 1), dtype=np.uint64)
    rc = _lib.qmcp_oracle_solve(_p32(s), _p32(e), n, _p64(offs), _p32(lengths), lengths.size, int(M),
                                _p64(mask))
    if rc:
        raise ValueError(f"oracle solve rc={rc}")
    return mask[:mask_words(n)]


def cover(starts, ends, ref_len, keep_mask=None):
    s, e = _u32(starts), _u32(ends)
    cov = np.zeros(max(ref_len, 1), dtype=np.uint32)
    km = np.ascontiguousarray(keep_mask, dtype=np.uint64) if keep_mask is not None else None
    rc = _lib.qmcp_oracle_cover(_p32(s), _p32(e), s.size, ref_len, _p64(km), 0, _p32(cov))
    if rc:
        raise ValueError(f"oracle cover rc={rc}")
    return cov[:ref_len]


def is_out_cover_valid(in_cover, out_cover, M):
    a, b = _u32(in_cover), _u32(out_cover)
    return bool(_lib.qmcp_oracle_is_out_cover_valid(_p32(a), _p32(b), a.size, int(M)))


def check_flow(starts, ends, ref_len, M, keep_mask):
    s, e = _u32(starts), _u32(ends)
    km = np.ascontiguousarray(keep_mask, dtype=np.uint64)
    val = C.c_int64(0)
    ok = _lib.qmcp_oracle_check_flow(_p32(s), _p32(e), s.size, ref_len, int(M), _p64(km), 0,
                                     C.byref(val))
    return bool(ok), int(val.value)


def maxflow_value(starts, ends, ref_len, M):
    s, e = _u32(starts), _u32(ends)
    return int(_lib.qmcp_oracle_maxflow_value(_p32(s), _p32(e), s.size, ref_len, int(M)))


def find_pairs(mask, n_reads):
    m = np.ascontiguousarray(mask, dtype=np.uint64).copy()
    _lib.qmcp_oracle_find_pairs(_p64(m), int(n_reads))
    return m


def amplicon_filter(starts, ends, amp_starts, amp_ends, seq_lengths=None, qualities=None,
                    min_length=0, min_mapq=0):
    s, e = _u32(starts), _u32(ends)
    a0, a1 = _u32(amp_starts), _u32(amp_ends)
    sl = _u32(seq_lengths) if seq_lengths is not None else None
    q = _u32(qualities) if qualities is not None else None
    out = np.zeros(max(mask_words(s.size // 2), 1), dtype=np.uint64)
    _lib.qmcp_oracle_amplicon_filter(_p32(s), _p32(e), _p32(sl), _p32(q), s.size, _p32(a0),
                                     _p32(a1), a0.size, int(min_length), int(min_mapq), _p64(out))
    return out[:mask_words(s.size // 2)]
