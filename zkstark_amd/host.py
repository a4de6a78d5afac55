"""Host-side mirror of the reference interface, written over the C ABI (ctypes).

Names, argument meaning and error behaviour follow the reference so that tests
read like the reference's own (the reference panics; here a ZkError is raised).
"""
import ctypes as C
import struct

import numpy as np

from . import _lib
from ._lib import ZkError, check

P = 3221225473  # main.rs:13
HASHES = {"sha256": 0, "field": 1}   # Merkle hash: the reference's SHA-256, or the field-native one (configs[4])


def _u32arr(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class field:
    """Scalar Gf<P> arithmetic (field.rs:8-211) on canonical residues."""
    P = P

    @staticmethod
    def add(a, b): return _lib.load().zk_field_add(a, b)
    @staticmethod
    def sub(a, b): return _lib.load().zk_field_sub(a, b)
    @staticmethod
    def mul(a, b): return _lib.load().zk_field_mul(a, b)
    @staticmethod
    def neg(a): return _lib.load().zk_field_neg(a)
    @staticmethod
    def inv(a): return _lib.load().zk_field_inv(a)
    @staticmethod
    def pow(a, e): return _lib.load().zk_field_pow(a, e)
    @staticmethod
    def from_u32(v): return _lib.load().zk_field_from_u32(v)
    @staticmethod
    def from_i32(v): return _lib.load().zk_field_from_i32(v)   # field.rs:10-18
    @staticmethod
    def div(a, b):                                             # field.rs:165-177; a zero divisor panics there
        if b % P == 0:
            raise ZeroDivisionError("Gf division by zero (field.rs:165-177)")
        return _lib.load().zk_field_div(a, b)
    @staticmethod
    def rem(a, rhs):                                           # field.rs:89-94
        if rhs == 0:
            raise ZeroDivisionError("Gf % 0 (field.rs:89-94)")
        return _lib.load().zk_field_rem(a, rhs)
    @staticmethod
    def generator(): return _lib.load().zk_field_generator()
    @staticmethod
    def root_of_unity(log_order): return _lib.load().zk_field_root_of_unity(log_order)
    @staticmethod
    def order(a): return _lib.load().zk_field_order(a)     # field.rs:45-49


def host_hash_mode():
    """How the host thread hashes its share of the trees: 'portable' (then everything stays on the device),
    'sha-ni', or 'sha-ni + avx512 x16' (levels of >= 16 nodes sixteen at a time); zk_host_hash_mode."""
    return ("portable", "sha-ni", "sha-ni + avx512 x16")[_lib.load().zk_host_hash_mode()]


def probe_hash_chain(hash="sha256", waves_per_simd=4, hashes=16, launches=10, device=0):
    """Roofline probe (zk_probe_hash_chain): steady-state rate of the compiled inner hash in a dependent chain."""
    r = _lib.ChainProbe()
    check(_lib.load().zk_probe_hash_chain(device, HASHES[hash], waves_per_simd, hashes, launches, C.byref(r)))
    return {"ns_per_hash_per_simd": r.ns_per_hash_per_simd, "clock_ghz": r.clock_ghz, "ms": r.ms,
            "waves_per_simd": r.waves_per_simd, "launches": r.launches, "hashes": r.hashes, "cus": r.cus}


def trace_fibsq(count, a0=1, a1=3141592):
    """prover.rs:32-39."""
    out = np.zeros(count, dtype=np.uint32)
    check(_lib.load().zk_trace_fibsq(a0, a1, count, _ptr(out)))
    return out


def trace_fibsq_batch(a0s, a1s, count, device=0):
    """Many independent traces on the GPU, one lane each: returns [batch, count]."""
    a0s, a1s = _u32arr(a0s), _u32arr(a1s)
    out = np.zeros((len(a0s), count), dtype=np.uint32)
    check(_lib.load().zk_trace_fibsq_batch_host(device, _ptr(a0s), _ptr(a1s), len(a0s), count, _ptr(out)))
    return out


# ---- bincode 1.x default encoding of the types the prover commits ----
def encode(x):
    if isinstance(x, (bytes, bytearray)):          # Hash = [u8; 32]: raw
        return bytes(x)
    if isinstance(x, (int, np.integer)):           # u32
        return struct.pack("<I", int(x))
    if isinstance(x, list):                        # AuthPath = Box<[Hash]>
        return struct.pack("<Q", len(x)) + b"".join(bytes(h) for h in x)
    if isinstance(x, tuple):
        return b"".join(encode(e) for e in x)
    raise TypeError(f"cannot encode {type(x)}")


class Channel:
    """channel.rs:6-37."""

    def __init__(self):                              # channel.rs:12
        self._h = C.c_void_p()
        check(_lib.load().zk_channel_new(C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            _lib.load().zk_channel_free(self._h)
            self._h = None

    def commit(self, data):                          # channel.rs:19
        b = encode(data)
        check(_lib.load().zk_channel_commit(self._h, b, len(b)))

    def get_u32(self):                               # channel.rs:28
        v = C.c_uint32()
        check(_lib.load().zk_channel_get_u32(self._h, C.byref(v)))
        return v.value

    @property
    def state(self):
        out = C.create_string_buffer(32)
        check(_lib.load().zk_channel_state(self._h, out))
        return out.raw

    @property
    def data(self):
        n = _lib.load().zk_channel_data_len(self._h)
        out = C.create_string_buffer(max(n, 1))
        check(_lib.load().zk_channel_data(self._h, out, n))
        return out.raw[:n]

    def finalize(self, log_n=10, log_blowup=3, public_last=2338775057):   # channel.rs:34
        return Proof(self.state, self.data, log_n, log_blowup, public_last)


class Proof:
    """proof.rs:5-154.  verify() raises ZkError where the reference panics."""

    def __init__(self, state, data, log_n=10, log_blowup=3, public_last=2338775057, hash="sha256", queries=1):   # proof.rs:11
        self.state, self.data = bytes(state), bytes(data)
        self.log_n, self.log_blowup, self.public_last = log_n, log_blowup, public_last
        self.hash, self.queries = hash, queries

    def verify(self, strict=False):                  # proof.rs:15
        """strict=True also replays the channel: challenges must come from the transcript and `state`
        must be its final state (the reference trusts the proof for both, proof.rs:22-37)."""
        check(_lib.load().zk_verify_queries(self.data, len(self.data), self.state if strict else None, self.log_n,
                                            self.log_blowup, self.public_last, HASHES[self.hash], self.queries))

    def size(self):                                  # proof.rs:151
        return _lib.load().zk_proof_size(len(self.data))


def compute_root_from_path(element, index, path, hash="sha256"):
    """merkle.rs:82-110."""
    flat = b"".join(bytes(h) for h in path)
    out = C.create_string_buffer(32)
    check(_lib.load().zk_compute_root_from_path_ex(element, index, flat, len(path), out, HASHES[hash]))
    return out.raw


class Merkle:
    """merkle.rs:6-79: SHA-256 heap built on the GPU; merkle[i], merkle.trace(i)."""

    def __init__(self, size, data, device=0, hash="sha256"):        # Merkle::new, merkle.rs:14
        vals = _u32arr(list(data) if not isinstance(data, np.ndarray) else data)
        if len(vals) != size:
            raise ZkError(-1, f"Merkle.new: size {size} != len(data) {len(vals)}")
        self.size = size
        self.nodes = np.zeros((max(2 * size - 1, 1), 32), dtype=np.uint8)
        check(_lib.load().zk_merkle_build_host_ex(device, _ptr(vals), size, _ptr(self.nodes), HASHES[hash]))

    new = classmethod(lambda cls, size, data, device=0, hash="sha256": cls(size, data, device, hash))

    def __getitem__(self, i):                        # merkle.rs:74-79
        return bytes(self.nodes[i])

    def __len__(self):
        return len(self.nodes)

    def trace(self, i):                              # merkle.rs:54-71
        i += len(self.nodes) // 2
        v = []
        while i != 0:
            if i % 2 == 0:
                v.append(self[i - 1]); i -= 2
            else:
                v.append(self[i + 1]); i -= 1
            i >>= 1
        return v


def ntt(data, inverse=False, device=0):
    """Natural-order NTT of size 2^k with root field.root_of_unity(k)."""
    a = _u32arr(data).copy()
    k = int(len(a)).bit_length() - 1
    if len(a) != 1 << k:
        raise ZkError(-1, "ntt: length must be a power of two")
    check(_lib.load().zk_ntt_host(device, _ptr(a), k, 1 if inverse else 0))
    return a


def lde(trace, log_n, log_blowup, device=0):
    """lagrange + solve over the coset (prover.rs:60-70) as one call."""
    t = _u32arr(trace)
    out = np.zeros(1 << (log_n + log_blowup), dtype=np.uint32)
    check(_lib.load().zk_lde_host(device, _ptr(t), log_n, log_blowup, _ptr(out)))
    return out


class Context:
    """Device-resident prover state for one (log_n, log_blowup): zk_ctx."""

    def __init__(self, log_n=10, log_blowup=3, device=0, hash="sha256", queries=1, host_levels=None):
        """host_levels: (top_log, tail_log) of zk_ctx_set_host_levels; None = the library default."""
        self.log_n, self.log_blowup, self.device, self.hash, self.queries = log_n, log_blowup, device, hash, queries
        self.n, self.B = 1 << log_n, 1 << log_blowup
        self.N, self.rounds = self.n * self.B, log_n
        self._h = C.c_void_p()
        check(_lib.load().zk_ctx_create(device, log_n, log_blowup, C.byref(self._h)))
        if hash != "sha256":
            check(_lib.load().zk_ctx_set_hash(self._h, HASHES[hash]))
        if queries != 1:
            check(_lib.load().zk_ctx_set_queries(self._h, queries))
        if host_levels is not None:
            check(_lib.load().zk_ctx_set_host_levels(self._h, host_levels[0], host_levels[1]))

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().zk_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self): return self
    def __exit__(self, *a): self.close()

    @property
    def setup_ms(self): return _lib.load().zk_ctx_setup_ms(self._h)
    @property
    def device_bytes(self): return _lib.load().zk_ctx_device_bytes(self._h)
    @property
    def stream(self): return _lib.load().zk_ctx_stream(self._h)

    @property
    def host_levels(self):
        a, b = C.c_uint32(), C.c_uint32()
        check(_lib.load().zk_ctx_get_host_levels(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def sync(self): check(_lib.load().zk_ctx_sync(self._h))
    def set_profiling(self, classes=()):
        """Time the named kernel classes (_lib.KERNEL_CLASSES) with HIP events; () = off, "all" = every class."""
        if classes == "all":
            classes = _lib.KERNEL_CLASSES
        mask = 0
        for c in classes:
            mask |= 1 << _lib.KERNEL_CLASSES.index(c)
        check(_lib.load().zk_ctx_set_profiling(self._h, mask))

    def kernel_stats(self, reset=True):
        arr = _lib.kernel_stat_array()
        check(_lib.load().zk_kernel_stats(self._h, arr, len(arr), int(reset)))
        return {name: {"launches": int(a.launches), "ms": a.ms, "bytes": a.bytes, "ops": a.ops}
                for name, a in zip(_lib.KERNEL_CLASSES, arr)}

    def layer_size(self, layer): return self.N if layer == 0 else self.N >> (layer - 1)

    def trace_upload(self, trace):
        t = _u32arr(trace)
        check(_lib.load().zk_trace_upload(self._h, _ptr(t), len(t)))

    def lde(self): check(_lib.load().zk_lde(self._h))

    def merkle_commit(self, layer):
        out = C.create_string_buffer(32)
        check(_lib.load().zk_merkle_commit(self._h, layer, out))
        return out.raw

    def compose(self, alpha_raw):
        a = _u32arr(alpha_raw)
        check(_lib.load().zk_compose(self._h, _ptr(a)))

    def fri_fold(self, rnd, beta_raw): check(_lib.load().zk_fri_fold(self._h, rnd, beta_raw))

    def layer_read(self, layer, offset=0, count=None):
        if count is None:
            count = self.layer_size(layer) - offset
        out = np.zeros(count, dtype=np.uint32)
        check(_lib.load().zk_layer_read(self._h, layer, offset, count, _ptr(out)))
        return out

    def layer_write(self, layer, values, offset=0):
        v = _u32arr(values)
        check(_lib.load().zk_layer_write(self._h, layer, offset, len(v), _ptr(v)))

    def merkle_node(self, tree, index):
        out = C.create_string_buffer(32)
        check(_lib.load().zk_merkle_node(self._h, tree, index, out))
        return out.raw

    def merkle_path(self, tree, leaf):
        buf = C.create_string_buffer(32 * 64)
        n = C.c_size_t()
        check(_lib.load().zk_merkle_path(self._h, tree, leaf, buf, C.byref(n)))
        return [buf.raw[32 * i:32 * i + 32] for i in range(n.value)]

    def prove(self, trace=None):
        """generate_proof as one C call (C++ host prover). trace=None: already uploaded."""
        cap = _lib.load().zk_proof_data_len_queries(self.log_n, self.log_blowup, self.queries)
        buf = C.create_string_buffer(cap)
        st = C.create_string_buffer(32)
        n = C.c_size_t()
        if trace is None:
            check(_lib.load().zk_prove_resident(self._h, buf, cap, C.byref(n), st))
        else:
            t = _u32arr(trace)
            check(_lib.load().zk_prove(self._h, _ptr(t), len(t), buf, cap, C.byref(n), st))
        info = self.last_transcript()
        return Proof(st.raw, buf.raw[:n.value], self.log_n, self.log_blowup, info.public_last, self.hash, self.queries)

    def prove_channel(self, channel):
        """generate_proof(channel) (prover.rs:9) in one C call on the caller's Channel (zk_prove_channel): the
        resident trace is proved on top of whatever the channel already holds; returns channel.finalize(...)."""
        check(_lib.load().zk_prove_channel(self._h, channel._h))
        return channel.finalize(self.log_n, self.log_blowup, self.last_transcript().public_last)

    def set_host_levels(self, top_log, tail_log):
        check(_lib.load().zk_ctx_set_host_levels(self._h, top_log, tail_log))

    def set_early_launch(self, on=True):
        """zk_ctx_set_early_launch: the next FRI round's launches are enqueued before the current commitment is waited for."""
        check(_lib.load().zk_ctx_set_early_launch(self._h, int(on)))
        return bool(_lib.load().zk_ctx_get_early_launch(self._h))

    def set_checks(self, on=True):
        """The reference's in-prover assertions (prover.rs:64-66, :148-159/:169, :228-251) inside prove()."""
        check(_lib.load().zk_ctx_set_checks(self._h, int(on)))

    def last_transcript(self):
        info = _lib.TranscriptInfo()
        check(_lib.load().zk_last_transcript(self._h, C.byref(info)))
        return info


def prove_many(ctxs):
    """Context.prove() (resident traces) on several contexts at once, one host thread each inside the
    library (zk_prove_many): returns the proofs in order."""
    ctxs = list(ctxs)
    c0 = ctxs[0]
    stride = max(_lib.load().zk_proof_data_len_queries(c.log_n, c.log_blowup, c.queries) for c in ctxs)
    handles = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    data = np.zeros((len(ctxs), stride), dtype=np.uint8)
    lens = (C.c_size_t * len(ctxs))()
    states = np.zeros((len(ctxs), 32), dtype=np.uint8)
    check(_lib.load().zk_prove_many(handles, len(ctxs), data.ctypes.data_as(C.c_void_p), stride, lens, states.ctypes.data_as(C.c_void_p)))
    out = []
    for i, c in enumerate(ctxs):
        out.append(Proof(states[i].tobytes(), data[i, :lens[i]].tobytes(), c.log_n, c.log_blowup, c.last_transcript().public_last,
                         c.hash, c.queries))
    return out


class BatchContext:
    """2^log_batch proofs of one size in lockstep (zk_batch_*): every stage is one
    launch over the whole batch; each proof has its own channel and is byte-identical to Context.prove()."""

    def __init__(self, log_n=10, log_blowup=3, log_batch=4, device=0, hash="sha256", queries=1):
        self.log_n, self.log_blowup, self.log_batch, self.hash, self.queries = log_n, log_blowup, log_batch, hash, queries
        self.n, self.batch = 1 << log_n, 1 << log_batch
        self._h = C.c_void_p()
        check(_lib.load().zk_batch_create(device, log_n, log_blowup, log_batch, C.byref(self._h)))
        if hash != "sha256":
            check(_lib.load().zk_batch_set_hash(self._h, HASHES[hash]))
        if queries != 1:
            check(_lib.load().zk_batch_set_queries(self._h, queries))

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().zk_batch_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self): return self
    def __exit__(self, *a): self.close()

    @property
    def device_bytes(self): return _lib.load().zk_batch_device_bytes(self._h)

    def set_traces(self, traces):
        """traces: [batch][n-1] canonical residues."""
        t = np.ascontiguousarray(traces, dtype=np.uint32)
        if t.shape != (self.batch, self.n - 1):
            raise ZkError(-1, f"expected traces of shape ({self.batch}, {self.n - 1})")
        check(_lib.load().zk_batch_set_traces(self._h, _ptr(t)))

    def gen_fibsq(self, a0s, a1s):
        """prover.rs:32-39 for every proof, on the device, from the seeds a0s[p], a1s[p]."""
        a0, a1 = _u32arr(a0s), _u32arr(a1s)
        if len(a0) != self.batch or len(a1) != self.batch:
            raise ZkError(-1, f"expected {self.batch} seeds")
        check(_lib.load().zk_batch_gen_fibsq(self._h, _ptr(a0), _ptr(a1)))

    def public_last(self):
        out = np.zeros(self.batch, dtype=np.uint32)
        check(_lib.load().zk_batch_public_last(self._h, _ptr(out)))
        return out

    def prove_raw(self):
        """Returns (proof bytes [batch][len] as a uint8 array, states [batch][32])."""
        plen = _lib.load().zk_proof_data_len_queries(self.log_n, self.log_blowup, self.queries)
        data = np.zeros((self.batch, plen), dtype=np.uint8)
        states = np.zeros((self.batch, 32), dtype=np.uint8)
        check(_lib.load().zk_batch_prove(self._h, data.ctypes.data_as(C.c_void_p), plen, states.ctypes.data_as(C.c_void_p)))
        return data, states

    def prove(self):
        data, states = self.prove_raw()
        last = self.public_last()
        return [Proof(states[p].tobytes(), data[p].tobytes(), self.log_n, self.log_blowup, int(last[p]), self.hash, self.queries)
                for p in range(self.batch)]


def shard_plan(world, log_n, log_blowup, min_layer_log=0, min_chunk_log=0, overlap_min_log=0, force_collectives=False, plain_collectives=False,
               exchange_cp=False, peer_copy=False):
    """zk_shard_plan: the layout zk_shard_create would choose (no GPU needed); a dict of the zk_shard_plan_info fields."""
    opt = _lib.ShardOptions(min_layer_log, min_chunk_log, overlap_min_log, int(force_collectives), 0, int(plain_collectives), 0, 0, int(exchange_cp), 0.0,
                            int(peer_copy))
    pl = _lib.ShardPlan()
    check(_lib.load().zk_shard_plan(world, log_n, log_blowup, C.byref(opt), C.byref(pl)))
    d = {k: v for k, v in pl.fields().items() if k != "piece_log"}
    d["piece_log"] = list(pl.piece_log)[:pl.sharded_layers + 1]
    return d


def shard_unique_id():
    """ncclGetUniqueId (rank 0): the 128 bytes every rank hands to ShardContext."""
    buf = C.create_string_buffer(128)
    check(_lib.load().zk_shard_unique_id(buf))
    return buf.raw


class ShardContext:
    """One proof sharded over `world` GPUs (zk_shard_*): this process is rank `rank`.  Collective: every rank
    constructs it and calls the same methods in the same order.  transport=None uses RCCL (native, inside the
    library) with the shared `unique_id`; a _lib.ShardTransport supplies the caller's own collectives."""

    def __init__(self, log_n, log_blowup, rank, world, unique_id=None, device=0, transport=None, min_layer_log=0, min_chunk_log=0,
                 overlap_min_log=0, force_collectives=False, no_root_board=False, hash="sha256", queries=1, plain_collectives=False,
                 single_build_stream=False, single_communicator=False, timeout_s=0.0, exchange_cp=False, peer_copy=False):
        self.log_n, self.log_blowup, self.rank, self.world = log_n, log_blowup, rank, world
        self.hash, self.queries = hash, queries
        self._transport = transport                      # keeps the callbacks alive
        opt = _lib.ShardOptions(min_layer_log, min_chunk_log, overlap_min_log, int(force_collectives), int(no_root_board),
                                int(plain_collectives), int(single_build_stream), int(single_communicator), int(exchange_cp), float(timeout_s),
                                int(peer_copy))
        self._h = C.c_void_p()
        idb = C.create_string_buffer(bytes(unique_id), 128) if unique_id is not None else None
        check(_lib.load().zk_shard_create(device, rank, world, idb, C.byref(transport) if transport is not None else None,
                                          C.byref(opt), log_n, log_blowup, C.byref(self._h)))
        if hash != "sha256":
            check(_lib.load().zk_shard_set_hash(self._h, HASHES[hash]))
        if queries != 1:
            check(_lib.load().zk_shard_set_queries(self._h, queries))

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().zk_shard_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self): return self
    def __exit__(self, *a): self.close()

    def trace_upload(self, trace):
        t = _u32arr(trace)
        check(_lib.load().zk_shard_trace_upload(self._h, _ptr(t), len(t)))

    def prove(self):
        cap = _lib.load().zk_proof_data_len_queries(self.log_n, self.log_blowup, self.queries)
        buf, st, n = C.create_string_buffer(cap), C.create_string_buffer(32), C.c_size_t()
        check(_lib.load().zk_shard_prove(self._h, buf, cap, C.byref(n), st))
        return Proof(st.raw, buf.raw[:n.value], self.log_n, self.log_blowup, self.last_transcript().public_last, self.hash, self.queries)

    def inject_failure(self, code=-4):
        """Test hook (zk_shard_inject_failure): this rank leaves the protocol with an error; peers must not hang."""
        return _lib.load().zk_shard_inject_failure(self._h, code)

    def self_test(self):
        """zk_shard_self_test: known-pattern all-to-all + all-gather through the transport (collective)."""
        check(_lib.load().zk_shard_self_test(self._h))

    def set_profiling(self, on):
        """zk_shard_set_profiling: HIP events around the exchanges of later proofs (stats(): exchange_ms, ...)."""
        check(_lib.load().zk_shard_set_profiling(self._h, int(bool(on))))

    def prove_channel(self, channel):
        check(_lib.load().zk_shard_prove_channel(self._h, channel._h))
        return channel.finalize(self.log_n, self.log_blowup, self.last_transcript().public_last)

    def lde_commit(self):
        out = C.create_string_buffer(32)
        check(_lib.load().zk_shard_lde_commit(self._h, out))
        return out.raw

    def last_transcript(self):
        info = _lib.TranscriptInfo()
        check(_lib.load().zk_shard_last_transcript(self._h, C.byref(info)))
        return info

    def layer_read(self, layer, offset, count):
        out = np.zeros(count, dtype=np.uint32)
        check(_lib.load().zk_shard_layer_read(self._h, layer, offset, count, _ptr(out)))
        return out

    def stats(self):
        st = _lib.ShardStats()
        check(_lib.load().zk_shard_get_stats(self._h, C.byref(st)))
        return st.fields()


def generate_proof(channel, log_n=10, log_blowup=3, a0=1, a1=3141592, ctx=None):
    """prover.rs:9-293, stage by stage over the C ABI, driven by `channel`.

    The reference literals are the defaults (trace 1023 values, domain 8192).
    Context.prove() is the same flow inside one C call.
    """
    own = ctx is None
    ctx = ctx or Context(log_n, log_blowup)
    try:
        n, B, N, R = ctx.n, ctx.B, ctx.N, ctx.rounds
        a = trace_fibsq(n - 1, a0, a1)                       # prover.rs:32-39
        ctx.trace_upload(a)
        ctx.lde()                                            # prover.rs:60-70
        channel.commit(ctx.merkle_commit(0))                 # prover.rs:81-85
        alphas = [channel.get_u32() for _ in range(3)]       # prover.rs:163-165
        ctx.compose(alphas)                                  # prover.rs:166-173
        channel.commit(ctx.merkle_commit(1))                 # prover.rs:176-180
        for r in range(R):                                   # prover.rs:198-225
            beta = channel.get_u32()
            ctx.fri_fold(r, beta)
            channel.commit(ctx.merkle_commit(2 + r))
        last = ctx.layer_read(1 + R)
        if not (last == last[0]).all():                      # prover.rs:238
            raise ZkError(-7, "last FRI layer is not constant")
        channel.commit(int(last[0]))                         # prover.rs:254
        x = channel.get_u32() % (N - 2 * B)                  # prover.rs:263
        for layer, idx in ((0, x), (0, x + B), (0, x + 2 * B), (1, x)):      # prover.rs:266-277
            channel.commit((int(ctx.layer_read(layer, idx, 1)[0]), ctx.merkle_path(layer, idx)))
        for i in range(R):                                   # prover.rs:280-289
            ln = N >> i
            xi = x % ln
            nx = (xi + ln // 2) % ln
            channel.commit((int(ctx.layer_read(1 + i, xi, 1)[0]), int(ctx.layer_read(1 + i, nx, 1)[0]),
                            ctx.merkle_path(1 + i, xi), ctx.merkle_path(1 + i, nx)))
        return channel.finalize(log_n, log_blowup, int(a[n - 2]))   # prover.rs:292
    finally:
        if own:
            ctx.close()
