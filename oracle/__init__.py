"""ctypes binding of the CPU oracle (oracle/stark101_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under zkstark_amd/ imports this package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

P = 3221225473
MODE_NTT, MODE_NAIVE = 0, 1


def _source_hash():
    import hashlib
    h = hashlib.sha256()
    for name in ("stark101_oracle.c", "stark101_oracle.h", "Makefile"):
        with open(os.path.join(_HERE, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:24]


def _stale():
    try:
        with open(_LIB_PATH + ".hash") as f:
            return f.read().strip() != _source_hash()
    except OSError:
        return True


def build(force=False):
    """Compile liboracle.so with gcc (seconds).  An existing library is used when it was built from the sources
    in the tree (a hash beside it says so; several test workers may import at once, hence the lock)."""
    if not force and os.path.exists(_LIB_PATH) and not _stale():
        return _LIB_PATH
    import fcntl
    with open(_LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or not os.path.exists(_LIB_PATH) or _stale():
            subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
            with open(_LIB_PATH + ".hash.tmp", "w") as f:
                f.write(_source_hash() + "\n")
            os.replace(_LIB_PATH + ".hash.tmp", _LIB_PATH + ".hash")
    return _LIB_PATH


class _Debug(C.Structure):
    _fields_ = [
        ("trace", C.c_void_p), ("f_eval", C.c_void_p), ("cp_layers", C.c_void_p), ("roots", C.c_void_p),
        ("alpha_raw", C.c_uint32 * 3), ("beta_raw", C.c_uint32 * 32), ("free_term", C.c_uint32),
        ("query_raw", C.c_uint32), ("public_last", C.c_uint32), ("cp_degree", C.c_uint32),
    ]


_lib = None


def usable_cores():
    """Cores this process may really use: the affinity mask capped by a cgroup CPU quota.  (A GPU box shows 256 logical
    CPUs and grants 16: OpenMP's default team of 256 threads then runs several times slower than 16 threads.)"""
    import math
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()
            if q != "max":
                n = min(n, max(1, math.ceil(int(q) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        u32, sz, vp = C.c_uint32, C.c_size_t, C.c_void_p
        for name in ("add", "sub", "mul"):
            getattr(L, "orc_" + name).restype = u32
            getattr(L, "orc_" + name).argtypes = [u32, u32, u32]
        L.orc_neg.restype = u32; L.orc_neg.argtypes = [u32, u32]
        L.orc_pow.restype = u32; L.orc_pow.argtypes = [u32, u32, u32]
        L.orc_inv.restype = u32; L.orc_inv.argtypes = [u32, u32]
        L.orc_from_u32.restype = u32; L.orc_from_u32.argtypes = [u32, u32]
        L.orc_from_i32.restype = u32; L.orc_from_i32.argtypes = [C.c_int32, u32]
        L.orc_div.restype = u32; L.orc_div.argtypes = [u32, u32, u32]
        L.orc_rem.restype = u32; L.orc_rem.argtypes = [u32, u32, u32]
        L.orc_order.restype = u32; L.orc_order.argtypes = [u32, u32]
        L.orc_generator.restype = u32; L.orc_generator.argtypes = [u32]
        L.orc_lagrange_naive.restype = None; L.orc_lagrange_naive.argtypes = [vp, vp, sz, vp, u32]
        L.orc_poly_solve_naive.restype = u32; L.orc_poly_solve_naive.argtypes = [vp, sz, u32, u32]
        L.orc_poly_div.restype = sz; L.orc_poly_div.argtypes = [vp, sz, vp, sz, vp, vp, C.POINTER(sz), u32]
        L.orc_poly_div_i32.restype = sz; L.orc_poly_div_i32.argtypes = [vp, sz, vp, sz, vp, vp, C.POINTER(sz)]
        L.orc_fri_coef_fold.restype = None; L.orc_fri_coef_fold.argtypes = [vp, sz, u32, vp, u32]
        L.orc_set_threads.restype = None; L.orc_set_threads.argtypes = [C.c_int]
        L.orc_ntt.restype = None; L.orc_ntt.argtypes = [vp, u32, u32]
        L.orc_intt.restype = None; L.orc_intt.argtypes = [vp, u32, u32]
        L.orc_virtual_point.restype = u32; L.orc_virtual_point.argtypes = [vp, u32]
        L.orc_trace_fibsq.restype = None; L.orc_trace_fibsq.argtypes = [u32, u32, sz, vp]
        L.orc_lde.restype = None; L.orc_lde.argtypes = [vp, u32, u32, vp]
        L.orc_compose.restype = None; L.orc_compose.argtypes = [vp, u32, u32, vp, u32, vp]
        L.orc_fri_fold_eval.restype = None; L.orc_fri_fold_eval.argtypes = [vp, u32, u32, u32, u32, vp]
        L.orc_sha256.restype = None; L.orc_sha256.argtypes = [vp, sz, vp]
        L.orc_merkle_build.restype = C.c_int; L.orc_merkle_build.argtypes = [vp, sz, vp]
        L.orc_node_hash.restype = None; L.orc_node_hash.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
        L.orc_merkle_trace.restype = sz; L.orc_merkle_trace.argtypes = [vp, sz, sz, vp]
        L.orc_compute_root_from_path.restype = None
        L.orc_compute_root_from_path.argtypes = [u32, sz, vp, sz, vp]
        L.orc_set_hash.restype = None; L.orc_set_hash.argtypes = [C.c_int]
        L.orc_set_fieldhash_batch.restype = None; L.orc_set_fieldhash_batch.argtypes = [C.c_int]
        L.orc_fieldhash_permute.restype = None; L.orc_fieldhash_permute.argtypes = [vp]
        L.orc_prove.restype = C.c_int
        L.orc_prove.argtypes = [u32, u32, u32, u32, C.c_int, vp, sz, C.POINTER(sz), vp, C.POINTER(_Debug)]
        L.orc_prove_prefixed.restype = C.c_int
        L.orc_prove_prefixed.argtypes = [vp, sz, u32, u32, u32, u32, vp, sz, C.POINTER(sz), vp]
        L.orc_verify.restype = C.c_int; L.orc_verify.argtypes = [vp, sz, u32, u32, u32]
        L.orc_set_queries.restype = None; L.orc_set_queries.argtypes = [u32]
        L.orc_proof_size.restype = sz; L.orc_proof_size.argtypes = [sz]
        L.orc_proof_data_len.restype = sz; L.orc_proof_data_len.argtypes = [u32, u32]
        _lib = L
        L.orc_set_threads(usable_cores())      # not OpenMP's default (every logical CPU, whatever the quota)
    return _lib


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- field ---------------------------------------------------------------
def add(a, b, p=P): return lib().orc_add(a, b, p)
def sub(a, b, p=P): return lib().orc_sub(a, b, p)
def mul(a, b, p=P): return lib().orc_mul(a, b, p)
def neg(a, p=P): return lib().orc_neg(a, p)
def pow_(a, e, p=P): return lib().orc_pow(a, e, p)
def inv(a, p=P): return lib().orc_inv(a, p)
def from_u32(v, p=P): return lib().orc_from_u32(v, p)
def from_i32(v, p=P): return lib().orc_from_i32(v, p)
def div(a, b, p=P): return lib().orc_div(a, b, p)
def rem(a, rhs, p=P): return lib().orc_rem(a, rhs, p)
def order(a, p=P): return lib().orc_order(a, p)
def generator(p=P): return lib().orc_generator(p)


def gen_of_order_log(log_order):
    return pow_(5, (P - 1) >> log_order)


# ---- polynomial (low degree first) ----------------------------------------
def lagrange_naive(xs, ys, p=P):
    xs, ys = _u32(xs), _u32(ys)
    out = np.zeros(len(xs), dtype=np.uint32)
    lib().orc_lagrange_naive(_ptr(xs), _ptr(ys), len(xs), _ptr(out), p)
    return out


def poly_solve_naive(coef, x, p=P):
    coef = _u32(coef)
    return lib().orc_poly_solve_naive(_ptr(coef), len(coef), x, p)


def poly_div(num, den, p=P):
    num, den = _u32(num), _u32(den)
    q = np.zeros(max(len(num), 1), dtype=np.uint32)
    r = np.zeros(max(len(num), 1), dtype=np.uint32)
    rl = C.c_size_t(0)
    ql = lib().orc_poly_div(_ptr(num), len(num), _ptr(den), len(den), _ptr(q), _ptr(r), C.byref(rl), p)
    return q[:ql].copy(), r[:rl.value].copy()


def poly_div_i32(num, den):
    num = np.ascontiguousarray(num, dtype=np.int32)
    den = np.ascontiguousarray(den, dtype=np.int32)
    q = np.zeros(max(len(num), 1), dtype=np.int32)
    r = np.zeros(max(len(num), 1), dtype=np.int32)
    rl = C.c_size_t(0)
    ql = lib().orc_poly_div_i32(_ptr(num), len(num), _ptr(den), len(den), _ptr(q), _ptr(r), C.byref(rl))
    return q[:ql].copy(), r[:rl.value].copy()


def fri_coef_fold(coef, beta, p=P):
    coef = _u32(coef)
    out = np.zeros(len(coef) // 2, dtype=np.uint32)
    lib().orc_fri_coef_fold(_ptr(coef), len(coef), beta, _ptr(out), p)
    return out


# ---- transforms ------------------------------------------------------------
def set_threads(n): lib().orc_set_threads(int(n))


def ntt(data, root):
    a = _u32(data).copy()
    lib().orc_ntt(_ptr(a), int(np.log2(len(a))), root)
    return a


def intt(data, root):
    a = _u32(data).copy()
    lib().orc_intt(_ptr(a), int(np.log2(len(a))), root)
    return a


def trace_fibsq(count, a0=1, a1=3141592):
    out = np.zeros(count, dtype=np.uint32)
    lib().orc_trace_fibsq(a0, a1, count, _ptr(out))
    return out


def virtual_point(trace, log_n):
    t = _u32(trace)
    return lib().orc_virtual_point(_ptr(t), log_n)


def lde(trace, log_n, log_b):
    t = _u32(trace)
    assert len(t) == (1 << log_n) - 1
    out = np.zeros(1 << (log_n + log_b), dtype=np.uint32)
    lib().orc_lde(_ptr(t), log_n, log_b, _ptr(out))
    return out


def compose(f_eval, log_n, log_b, alpha_raw, public_last):
    f = _u32(f_eval)
    al = _u32(alpha_raw)
    out = np.zeros(len(f), dtype=np.uint32)
    lib().orc_compose(_ptr(f), log_n, log_b, _ptr(al), public_last, _ptr(out))
    return out


def fri_fold_eval(layer, log_n, log_b, rnd, beta_raw):
    e = _u32(layer)
    out = np.zeros(len(e) // 2, dtype=np.uint32)
    lib().orc_fri_fold_eval(_ptr(e), log_n, log_b, rnd, beta_raw, _ptr(out))
    return out


# ---- sha / merkle ------------------------------------------------------------
HASH_SHA256, HASH_FIELD = 0, 1


def set_hash(kind):
    """0 = SHA-256 (reference), 1 = field-native hash (configs[4]); affects merkle_*, prove, verify."""
    lib().orc_set_hash(int(kind))


def fieldhash_permute(state):
    s = _u32(state).copy()
    assert len(s) == 16
    lib().orc_fieldhash_permute(_ptr(s))
    return s


def sha256(msg: bytes) -> bytes:
    out = C.create_string_buffer(32)
    lib().orc_sha256(msg, len(msg), out)
    return out.raw


def set_fieldhash_batch(on):
    """Field hash in orc_merkle_build: eight hashes at a time in exact double arithmetic on AVX-512 registers (default on
    where the CPU has AVX-512F) or the scalar plain-residue code; the digests are the same (tests pin one on the other)."""
    lib().orc_set_fieldhash_batch(int(bool(on)))


def merkle_build(vals):
    v = _u32(vals)
    nodes = np.zeros((2 * len(v) - 1, 32), dtype=np.uint8)
    rc = lib().orc_merkle_build(_ptr(v), len(v), _ptr(nodes))
    if rc:
        raise ValueError("merkle size must be a power of two")
    return nodes


def node_hash(left: bytes, right: bytes) -> bytes:
    """merkle.rs:42-45 for one pair of 32-byte digests (the hash selected by set_hash)."""
    out = C.create_string_buffer(32)
    lib().orc_node_hash(bytes(left), bytes(right), out)
    return out.raw


def merkle_trace(nodes, leaf):
    m = (len(nodes) + 1) // 2
    path = np.zeros((64, 32), dtype=np.uint8)
    k = lib().orc_merkle_trace(_ptr(nodes), m, leaf, _ptr(path))
    return path[:k].copy()


def compute_root_from_path(element, index, path):
    path = np.ascontiguousarray(path, dtype=np.uint8)
    out = C.create_string_buffer(32)
    lib().orc_compute_root_from_path(element, index, _ptr(path), len(path), out)
    return out.raw


# ---- prover / verifier ---------------------------------------------------------
class ProveResult:
    pass


def prove(log_n=10, log_b=3, a0=1, a1=3141592, mode=MODE_NTT, want_vectors=True, want_roots=False):
    n, N, R = 1 << log_n, 1 << (log_n + log_b), log_n
    cap = lib().orc_proof_data_len(log_n, log_b)
    buf = np.zeros(cap, dtype=np.uint8)
    state = np.zeros(32, dtype=np.uint8)
    plen = C.c_size_t(0)
    dbg = _Debug()
    res = ProveResult()
    if want_vectors:
        res.trace = np.zeros(n - 1, dtype=np.uint32)
        res.f_eval = np.zeros(N, dtype=np.uint32)
        res.cp_flat = np.zeros(2 * N, dtype=np.uint32)
        res.roots = np.zeros((R + 2, 32), dtype=np.uint8)
        dbg.trace, dbg.f_eval = res.trace.ctypes.data, res.f_eval.ctypes.data
        dbg.cp_layers, dbg.roots = res.cp_flat.ctypes.data, res.roots.ctypes.data
    elif want_roots:                 # the R + 2 Merkle roots only (large domains: no layer copies)
        res.roots = np.zeros((R + 2, 32), dtype=np.uint8)
        dbg.roots = res.roots.ctypes.data
    rc = lib().orc_prove(log_n, log_b, a0, a1, mode, _ptr(buf), cap, C.byref(plen), _ptr(state), C.byref(dbg))
    res.rc = rc
    res.proof = bytes(buf[:plen.value])
    res.state = bytes(state)
    res.alpha_raw = list(dbg.alpha_raw)
    res.beta_raw = list(dbg.beta_raw)[:R]
    res.free_term, res.query_raw = dbg.free_term, dbg.query_raw
    res.public_last, res.cp_degree = dbg.public_last, dbg.cp_degree
    if want_vectors:
        res.cp_layers, off = [], 0
        for r in range(R + 1):
            res.cp_layers.append(res.cp_flat[off:off + (N >> r)])
            off += N >> r
    return res


def prove_prefixed(prefix: bytes, log_n=10, log_b=3, a0=1, a1=3141592):
    """generate_proof on a channel that committed `prefix` first: returns (Channel.data, Channel.state)."""
    cap = len(prefix) + lib().orc_proof_data_len(log_n, log_b)
    buf = np.zeros(cap, dtype=np.uint8)
    state = np.zeros(32, dtype=np.uint8)
    plen = C.c_size_t(0)
    rc = lib().orc_prove_prefixed(prefix, len(prefix), log_n, log_b, a0, a1, _ptr(buf), cap, C.byref(plen), _ptr(state))
    if rc:
        raise ValueError(f"orc_prove_prefixed failed: {rc}")
    return bytes(buf[:plen.value]), bytes(state)


def verify(proof: bytes, log_n, log_b, public_last):
    return lib().orc_verify(proof, len(proof), log_n, log_b, public_last)


def set_queries(q):
    """Decommitment queries per proof (1 = the reference); affects prove, verify, proof_data_len."""
    lib().orc_set_queries(int(q))


def proof_size(data_len): return lib().orc_proof_size(data_len)
def proof_data_len(log_n, log_b): return lib().orc_proof_data_len(log_n, log_b)
