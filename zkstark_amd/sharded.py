"""Caller-side transports for the native sharded prover (zk_shard_*, csrc/shard.hip): the two collectives of
`zk_shard_transport` (include/zkstark_amd.h) implemented over torch.distributed.

The sharded prover itself is native code behind the C ABI (`zkstark_amd.ShardContext`); by default its collectives are
RCCL's, loaded by the library.  A caller with its own communication layer passes a transport instead:

  * `staged_transport`  -- host-staged over any process group (gloo): several ranks sharing ONE GPU (tests, rehearsals);
  * `device_transport`  -- torch.distributed's own RCCL communicator on the library's device pointers and stream: what
                           `bench.py` falls back to when the built-in transport cannot be brought up.

(Rounds 1-3 also kept a torch.distributed MIRROR of the whole protocol here; it is test infrastructure now:
tests/sharded_mirror.py.  The package holds ONE implementation of the sharded protocol: csrc/shard.hip.)
"""
import ctypes as C

from . import _lib


def staged_transport(group=None):
    """A zk_shard_transport (include/zkstark_amd.h) over torch.distributed with the device buffers staged through
    host memory: for process groups without device collectives (gloo), e.g. several ranks sharing ONE GPU, where
    RCCL cannot be used.  Never a measurement configuration.  Returns a _lib.ShardTransport; keep it alive as
    long as the zk_shard that uses it."""
    import os
    import traceback
    import torch
    import torch.distributed as dist
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))   # the runtime already loaded
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    D2H, H2D = 2, 1
    world = dist.get_world_size(group)

    def d2h(ptr, words):
        t = torch.empty(words, dtype=torch.int32)
        if hip.hipMemcpy(t.data_ptr(), ptr, words * 4, D2H) != 0:
            raise RuntimeError("hipMemcpy D2H failed")
        return t

    def h2d(ptr, t):
        if hip.hipMemcpy(ptr, t.data_ptr(), t.numel() * 4, H2D) != 0:
            raise RuntimeError("hipMemcpy H2D failed")

    def all_to_all(user, send, recv, words, stream):
        try:
            if hip.hipStreamSynchronize(stream) != 0:
                raise RuntimeError("hipStreamSynchronize failed")
            s = torch.cat([d2h(send[p], words) for p in range(world)])
            r = torch.empty_like(s)
            dist.all_to_all_single(r, s, group=group)
            for q in range(world):
                h2d(recv[q], r[q * words:(q + 1) * words].contiguous())
            return 0
        except Exception:                                  # a ctypes callback must not raise
            traceback.print_exc()
            return 1

    def all_gather(user, send, recv, words, stream):
        try:
            if hip.hipStreamSynchronize(stream) != 0:
                raise RuntimeError("hipStreamSynchronize failed")
            r = torch.empty(words * world, dtype=torch.int32)
            dist.all_gather_into_tensor(r, d2h(send, words), group=group)
            h2d(recv, r)
            return 0
        except Exception:
            traceback.print_exc()
            return 1

    return _lib.ShardTransport(None, _lib.ALL_TO_ALL_FN(all_to_all), _lib.ALL_GATHER_FN(all_gather))


def device_transport(group):
    """A zk_shard_transport over a torch.distributed process group with DEVICE collectives (backend "nccl" = RCCL on ROCm):
    the library's device pointers are wrapped as tensors (no copy) and exchanged with torch's own RCCL communicator, on
    the library's stream.  bench.py uses it when the built-in transport (RCCL loaded by the library itself) cannot be
    set up on a multi-GPU node: a second, independent way to the same wire.  Keep the returned object alive as long as
    the zk_shard that uses it."""
    import traceback
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)

    class _Raw:                                            # __cuda_array_interface__ view of a raw device pointer
        def __init__(self, ptr, words):
            self.__cuda_array_interface__ = {"shape": (words,), "typestr": "<i4", "data": (int(ptr), False), "version": 2}

    def view(ptr, words):
        return torch.as_tensor(_Raw(ptr, words), device=torch.device("cuda", torch.cuda.current_device()))

    def all_to_all(user, send, recv, words, stream):
        try:
            ext = torch.cuda.ExternalStream(int(stream))
            with torch.cuda.stream(ext):                   # ordered after the producer of the send pieces, before the hashing
                outs = [view(recv[q], words) for q in range(world)]
                ins = [view(send[p], words) for p in range(world)]
                dist.all_to_all(outs, ins, group=group)
            return 0
        except Exception:                                  # a ctypes callback must not raise
            traceback.print_exc()
            return 1

    def all_gather(user, send, recv, words, stream):
        try:
            ext = torch.cuda.ExternalStream(int(stream))
            with torch.cuda.stream(ext):
                dist.all_gather_into_tensor(view(recv, words * world), view(send, words), group=group)
            return 0
        except Exception:
            traceback.print_exc()
            return 1

    return _lib.ShardTransport(None, _lib.ALL_TO_ALL_FN(all_to_all), _lib.ALL_GATHER_FN(all_gather))
