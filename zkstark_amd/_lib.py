"""ctypes loader for libzkstark_amd.so (the C ABI declared in include/zkstark_amd.h).

There is no CPU fallback: if the library is missing and cannot be built with
hipcc, importing raises.  Compute entry points need a visible MI355X.
"""
import ctypes as C
import os

from . import build as _build

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libzkstark_amd.so")

ZK_OK = 0
ERR_NAMES = {-1: "ZK_ERR_INVALID", -2: "ZK_ERR_HIP", -3: "ZK_ERR_NOMEM", -4: "ZK_ERR_STATE",
             -5: "ZK_ERR_BUFFER", -6: "ZK_ERR_VERIFY", -7: "ZK_ERR_CHECK"}


class ZkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


ABI_VERSION = 6   # ZK_ABI_VERSION of include/zkstark_amd.h these declarations were written against; load() checks it


class _Sized(C.Structure):
    """A caller-allocated struct of the C ABI: its first field is struct_size = sizeof(the struct), which the library
    checks before it reads or writes anything (include/zkstark_amd.h, "ABI version and caller-allocated structs")."""

    def __init__(self, *args, **kw):
        super().__init__(C.sizeof(type(self)), *args, **kw)

    def fields(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k not in ("struct_size", "reserved")}


class TranscriptInfo(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("alpha_raw", C.c_uint32 * 3), ("beta_raw", C.c_uint32 * 32), ("free_term", C.c_uint32),
                ("query_raw", C.c_uint32), ("public_last", C.c_uint32), ("roots", (C.c_uint8 * 32) * 34)]


class KernelStat(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("reserved", C.c_uint32), ("launches", C.c_uint64), ("ms", C.c_double), ("bytes", C.c_double), ("ops", C.c_double)]


# zk_shard_transport: the two collectives of the sharded prover, supplied by the caller (tests: gloo-staged)
ALL_TO_ALL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p)
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class ShardTransport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_to_all", ALL_TO_ALL_FN), ("all_gather", ALL_GATHER_FN)]


def kernel_stat_array():
    """One zk_kernel_stat per kernel class; out[0].struct_size is the array's stride."""
    arr = (KernelStat * len(KERNEL_CLASSES))()
    for e in arr:
        e.struct_size = C.sizeof(KernelStat)
    return arr


class ShardOptions(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("min_layer_log", C.c_uint32), ("min_chunk_log", C.c_uint32), ("overlap_min_log", C.c_uint32),
                ("force_collectives", C.c_int), ("no_root_board", C.c_int), ("plain_collectives", C.c_int),
                ("single_build_stream", C.c_int), ("single_communicator", C.c_int), ("exchange_cp", C.c_int), ("timeout_s", C.c_double),
                ("peer_copy", C.c_int), ("reserved", C.c_int)]


class ShardStats(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("sharded_layers", C.c_uint32), ("root_board", C.c_uint32), ("chunked_layers", C.c_uint32), ("native_rccl", C.c_uint32),
                ("rccl_nranks", C.c_uint32), ("communicators", C.c_uint32), ("sent_bytes", C.c_double), ("all_to_all_bytes", C.c_double), ("setup_ms", C.c_double), ("device_bytes", C.c_double),
                ("exchange_ms", C.c_double), ("exposed_exchange_ms", C.c_double), ("tail_ms", C.c_double), ("selftest_ms", C.c_double), ("decommit_ms", C.c_double),
                ("exchanges", C.c_uint32), ("selftest_ok", C.c_uint32), ("peer_copy", C.c_uint32), ("reserved", C.c_uint32)]


class ChainProbe(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("reserved", C.c_uint32), ("ns_per_hash_per_simd", C.c_double), ("clock_ghz", C.c_double), ("ms", C.c_double),
                ("waves_per_simd", C.c_uint32), ("launches", C.c_uint32), ("hashes", C.c_uint32), ("cus", C.c_uint32)]


class ShardPlan(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("world", C.c_uint32), ("log_world", C.c_uint32), ("sharded_layers", C.c_uint32), ("tail_rounds", C.c_uint32),
                ("chunked_layers", C.c_uint32), ("chunked_mask", C.c_uint32), ("log_chunks", C.c_uint32), ("min_layer_log", C.c_uint32),
                ("min_chunk_log", C.c_uint32), ("overlap_min_log", C.c_uint32), ("piece_log", C.c_uint32 * 32),
                ("all_to_all_bytes", C.c_double), ("lde_commit_bytes", C.c_double), ("cp_from_f", C.c_uint32), ("reserved", C.c_uint32)]


KERNEL_CLASSES = ("ntt", "merkle_leaf", "merkle_inner", "merkle_top", "compose", "fri_fold", "gather")


# name -> (restype, argtypes); every symbol include/zkstark_amd.h declares
_u32, _sz, _vp, _int, _dbl = C.c_uint32, C.c_size_t, C.c_void_p, C.c_int, C.c_double
_cp = C.c_char_p
SYMBOLS = {
    "zk_last_error": (_cp, []),
    "zk_version": (_cp, []),
    "zk_build_hash": (_cp, []),
    "zk_abi_version": (_u32, []),
    "zk_host_hash_mode": (C.c_int, []),
    "zk_host_set_hash_mode": (C.c_int, [C.c_int]),
    "zk_field_add": (_u32, [_u32, _u32]),
    "zk_field_sub": (_u32, [_u32, _u32]),
    "zk_field_mul": (_u32, [_u32, _u32]),
    "zk_field_neg": (_u32, [_u32]),
    "zk_field_inv": (_u32, [_u32]),
    "zk_field_pow": (_u32, [_u32, _u32]),
    "zk_field_from_u32": (_u32, [_u32]),
    "zk_field_from_i32": (_u32, [C.c_int32]),
    "zk_probe_fieldhash_forms": (_int, [_int, _u32, _u32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "zk_field_div": (_u32, [_u32, _u32]),
    "zk_field_rem": (_u32, [_u32, _u32]),
    "zk_field_generator": (_u32, []),
    "zk_field_root_of_unity": (_u32, [_u32]),
    "zk_field_order": (_u32, [_u32]),
    "zk_dev_trace_fibsq_batch": (_int, [_vp, _vp, _u32, _u32, _vp, _vp]),
    "zk_trace_fibsq_batch_host": (_int, [_int, _vp, _vp, _u32, _u32, _vp]),
    "zk_ctx_create": (_int, [_int, _u32, _u32, C.POINTER(_vp)]),
    "zk_ctx_destroy": (_int, [_vp]),
    "zk_ctx_setup_ms": (_dbl, [_vp]),
    "zk_ctx_device_bytes": (_sz, [_vp]),
    "zk_ctx_sync": (_int, [_vp]),
    "zk_ctx_set_hash": (_int, [_vp, _int]),
    "zk_ctx_set_checks": (_int, [_vp, _int]),
    "zk_ctx_set_host_levels": (_int, [_vp, _u32, _u32]),
    "zk_ctx_get_host_levels": (_int, [_vp, C.POINTER(_u32), C.POINTER(_u32)]),
    "zk_ctx_set_queries": (_int, [_vp, _u32]),
    "zk_ctx_set_early_launch": (_int, [_vp, _int]),
    "zk_ctx_get_early_launch": (_int, [_vp]),
    "zk_verify_queries": (_int, [_vp, _sz, _vp, _u32, _u32, _u32, _int, _u32]),
    "zk_proof_data_len_queries": (_sz, [_u32, _u32, _u32]),
    "zk_ctx_stream": (_vp, [_vp]),
    "zk_ctx_set_profiling": (_int, [_vp, _u32]),
    "zk_kernel_stats": (_int, [_vp, _vp, _sz, _int]),
    "zk_trace_fibsq": (_int, [_u32, _u32, _sz, _vp]),
    "zk_trace_upload": (_int, [_vp, _vp, _sz]),
    "zk_lde": (_int, [_vp]),
    "zk_merkle_commit": (_int, [_vp, _u32, _vp]),
    "zk_compose": (_int, [_vp, _vp]),
    "zk_fri_fold": (_int, [_vp, _u32, _u32]),
    "zk_layer_read": (_int, [_vp, _u32, _sz, _sz, _vp]),
    "zk_layer_write": (_int, [_vp, _u32, _sz, _sz, _vp]),
    "zk_merkle_node": (_int, [_vp, _u32, _sz, _vp]),
    "zk_merkle_path": (_int, [_vp, _u32, _sz, _vp, C.POINTER(_sz)]),
    "zk_prove_resident": (_int, [_vp, _vp, _sz, C.POINTER(_sz), _vp]),
    "zk_prove_channel": (_int, [_vp, _vp]),
    "zk_prove_many": (_int, [_vp, _sz, _vp, _sz, _vp, _vp]),
    "zk_prove": (_int, [_vp, _vp, _sz, _vp, _sz, C.POINTER(_sz), _vp]),
    "zk_last_transcript": (_int, [_vp, C.POINTER(TranscriptInfo)]),
    "zk_verify": (_int, [_vp, _sz, _u32, _u32, _u32]),
    "zk_verify_ex": (_int, [_vp, _sz, _u32, _u32, _u32, _int]),
    "zk_compute_root_from_path_ex": (_int, [_u32, _sz, _vp, _sz, _vp, _int]),
    "zk_merkle_build_host_ex": (_int, [_int, _vp, _sz, _vp, _int]),
    "zk_dev_merkle_build_ex": (_int, [_vp, _u32, _vp, _vp, _int]),
    "zk_dev_merkle_build_interleaved": (_int, [_vp, _u32, _u32, _vp, _vp, _int]),
    "zk_dev_merkle_build_chunk": (_int, [_vp, _u32, _u32, _vp, _u32, _u32, _vp, _int]),
    "zk_batch_create": (_int, [_int, _u32, _u32, _u32, C.POINTER(_vp)]),
    "zk_batch_destroy": (_int, [_vp]),
    "zk_batch_size": (_sz, [_vp]),
    "zk_batch_set_queries": (_int, [_vp, _u32]),
    "zk_batch_set_hash": (_int, [_vp, _int]),
    "zk_batch_set_host_levels": (_int, [_vp, _int]),
    "zk_batch_set_threads": (_int, [_vp, _u32]),
    "zk_batch_device_bytes": (_sz, [_vp]),
    "zk_batch_set_traces": (_int, [_vp, _vp]),
    "zk_batch_gen_fibsq": (_int, [_vp, _vp, _vp]),
    "zk_batch_public_last": (_int, [_vp, _vp]),
    "zk_batch_prove": (_int, [_vp, _vp, _sz, _vp]),
    "zk_committer_create": (_int, [_int, C.POINTER(_vp)]),
    "zk_committer_destroy": (_int, [_vp]),
    "zk_committer_set_top": (_int, [_vp, _u32]),
    "zk_dev_merkle_commit": (_int, [_vp, _vp, _u32, _u32, _vp, _vp, _int, _vp]),
    "zk_dev_merkle_commit_finish": (_int, [_vp, _vp, _u32, _u32, _vp, _int, _vp]),
    "zk_dev_merkle_finish": (_int, [_vp, _u32, _u32, _vp, _int]),
    "zk_verify_strict": (_int, [_vp, _sz, _vp, _u32, _u32, _u32]),
    "zk_proof_size": (_sz, [_sz]),
    "zk_proof_data_len": (_sz, [_u32, _u32]),
    "zk_compute_root_from_path": (_int, [_u32, _sz, _vp, _sz, _vp]),
    "zk_channel_new": (_int, [C.POINTER(_vp)]),
    "zk_channel_free": (_int, [_vp]),
    "zk_channel_import": (_int, [_vp, _vp, _vp, _sz]),
    "zk_channel_commit": (_int, [_vp, _vp, _sz]),
    "zk_channel_get_u32": (_int, [_vp, C.POINTER(_u32)]),
    "zk_channel_state": (_int, [_vp, _vp]),
    "zk_channel_data_len": (_sz, [_vp]),
    "zk_channel_data": (_int, [_vp, _vp, _sz]),
    "zk_merkle_build_host": (_int, [_int, _vp, _sz, _vp]),
    "zk_ntt_host": (_int, [_int, _vp, _u32, _int]),
    "zk_lde_host": (_int, [_int, _vp, _u32, _u32, _vp]),
    "zk_dom_create": (_int, [_int, _u32, _u32, _u32, _int, C.POINTER(_vp)]),
    "zk_dom_destroy": (_int, [_vp]),
    "zk_dev_lde": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "zk_dev_compose": (_int, [_vp, _vp, _vp, _u32, _u32, _vp, _vp]),
    "zk_dev_fri_fold": (_int, [_vp, _vp, _vp, _u32, _u32, _u32, _vp]),
    "zk_dev_interleave": (_int, [_vp, _vp, _u32, _u32, _vp]),
    "zk_dev_gather": (_int, [_vp, _vp, _u32, _u32, _vp, _vp]),
    "zk_tail_create": (_int, [_int, _u32, _u32, _u32, C.POINTER(_vp)]),
    "zk_tail_run": (_int, [_vp, _vp, _vp, _vp, _int, _vp, _vp, C.POINTER(_u32)]),
    "zk_tail_open": (_int, [_vp, _sz, _vp, _vp]),
    "zk_shard_plan": (_int, [_int, _u32, _u32, C.POINTER(ShardOptions), C.POINTER(ShardPlan)]),
    "zk_shard_unique_id": (_int, [_vp]),
    "zk_shard_create": (_int, [_int, _int, _int, _vp, C.POINTER(ShardTransport), C.POINTER(ShardOptions), _u32, _u32, C.POINTER(_vp)]),
    "zk_shard_destroy": (_int, [_vp]),
    "zk_shard_trace_upload": (_int, [_vp, _vp, _sz]),
    "zk_shard_prove_channel": (_int, [_vp, _vp]),
    "zk_shard_prove": (_int, [_vp, _vp, _sz, C.POINTER(_sz), _vp]),
    "zk_shard_lde_commit": (_int, [_vp, _vp]),
    "zk_shard_set_hash": (_int, [_vp, _int]),
    "zk_shard_set_queries": (_int, [_vp, _u32]),
    "zk_shard_inject_failure": (_int, [_vp, _int]),
    "zk_shard_self_test": (_int, [_vp]),
    "zk_shard_set_profiling": (_int, [_vp, _int]),
    "zk_shard_last_transcript": (_int, [_vp, C.POINTER(TranscriptInfo)]),
    "zk_shard_layer_read": (_int, [_vp, _u32, _sz, _sz, _vp]),
    "zk_shard_get_stats": (_int, [_vp, C.POINTER(ShardStats)]),
    "zk_probe_hash_chain": (_int, [_int, _int, _u32, _u32, _u32, C.POINTER(ChainProbe)]),
    "zk_dev_set_profiling": (_int, [_u32]),
    "zk_dev_kernel_stats": (_int, [_vp, _sz, _int]),
    "zk_dev_merkle_build": (_int, [_vp, _u32, _vp, _vp]),
    "zk_dev_merkle_node": (_int, [_vp, _sz, _vp, _vp]),
    "zk_dev_set_merkle_latency_log": (_int, [_u32]),
}

_lib = None


def load():
    """Returns the loaded CDLL; builds it first if hipcc is present and the .so is missing or was built from
    other sources (hash of sources + headers, zk_build_hash); never loads a stale binary silently."""
    global _lib
    if _lib is not None:
        return _lib
    # The authority on what a binary was built from is the hash compiled into it (zk_build_hash); the git-ignored
    # sidecar file only saves a dlopen when deciding whether to rebuild.
    want, build_err = None, None
    try:
        want = _build.source_hash()          # needs csrc/: absent in a deployment that ships only the .so
        _build.build()                       # rebuilds (under a lock) when the library is missing or built from other sources
    except Exception as e:                   # no sources, or no hipcc on this box: only an existing library will do
        build_err = e
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"libzkstark_amd.so is missing and cannot be built here: {e}") from e
    # One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64.so (SONAME
    # libamdhip64.so.7); if it is loaded first the dynamic loader resolves this library's
    # libamdhip64.so.7 to the same object, so torch tensors, streams and RCCL interoperate with
    # the kernels here.  Loaded in the other order the process ends up with two runtimes.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)   # AttributeError if the library does not export it
        fn.restype, fn.argtypes = res, args
    if lib.zk_abi_version() != ABI_VERSION:
        raise ImportError(f"libzkstark_amd.so speaks ABI version {lib.zk_abi_version()}, zkstark_amd/_lib.py was written for {ABI_VERSION}")
    got = lib.zk_build_hash().decode()
    if want is not None and got != want and os.environ.get("ZK_ALLOW_STALE_LIB") != "1":
        why = f" and cannot be rebuilt here: {build_err}" if build_err is not None else " (python -m zkstark_amd.build)"
        raise ImportError(f"libzkstark_amd.so is stale: built from {got}, the tree is {want}{why}")
    _lib = lib
    return lib


def build_hash():
    """Hash of the sources the loaded library was built from (also in the bench line)."""
    return load().zk_build_hash().decode()


def check(code):
    if code != ZK_OK:
        raise ZkError(code, load().zk_last_error().decode())
    return code
