"""TEST INFRASTRUCTURE (moved out of the package in round 4): a torch.distributed MIRROR of the native sharded prover
(csrc/shard.hip, zk_shard_*).  Same protocol, same layout (zk_shard_plan), same bytes -- written a second time in Python
so that the exchange logic can be run on CPU (gloo, world 2/4/8) with a test-double backend, where the native prover
(which needs a GPU for every kernel) cannot run.  Nothing in zkstark_amd/ or bench.py imports this file.

One proof across G GPUs of one node (one process per GPU, RCCL over xGMI).

Layout.  The evaluation domain is distributed CYCLICALLY: rank r holds the
elements i = r (mod G) of every layer.  Rank r's shard of the size-N coset {w h^i} is itself a
coset domain {(w h^r) (h^G)^j, j < N/G} with blow-up B/G, so

  * LDE          rank r evaluates its B/G cosets of the (replicated) size-n interpolant,
  * composition  taps i+B, i+2B are local (G divides B),
  * FRI fold     pairs (i, i+m/2) are local, and the folded layer is again cyclic,

all with no communication, using the same kernels as the single-GPU path on a domain with
shift w h^r.  The only exchange step is the commitment: Merkle leaves are in natural order, so
per committed layer ONE all-to-all of the 4-byte values turns the cyclic layout into contiguous
blocks of m/G leaves, each rank hashes its subtree, the G subtree roots (32 B each) are
all-gathered and the top log2(G) levels are hashed on the host by every rank.  Once a layer has
fewer than 2^min_layer_log values in total (default: zk_shard_plan's, 2^21 or 2^20: a sharded commitment carries ~70 us of
fixed cost -- two collectives, two device-to-host reads, the latency phase of the subtree -- which
is what hashing 2^21 leaves redundantly costs) or fewer than 2^min_chunk_log leaves per (rank, peer)
chunk, it is all-gathered once and the remaining layers are folded and committed on every rank in
one C call (zk_tail_*).  The transcript
(channel.rs) runs identically on every rank, so challenges are never broadcast.  No all-reduce.

Collectives per proof: (number of sharded layers + 1) all-to-alls, as many 256-byte
all-gathers, one all-gather for the replication switch and one for the decommitment.

The protocol logic lives here and is shared by two compute backends with the same interface:
`HipBackend` (the product: gfx950 kernels through the C ABI on torch CUDA tensors) and a CPU
test double that tests inject (tests/test_sharded_gloo.py); there is no CPU fallback in this file.
"""
import ctypes as C
import hashlib
import os
import struct
import time

import numpy as np

from zkstark_amd import _lib
from zkstark_amd._lib import ZkError, check
from zkstark_amd.host import Channel, Proof, P, trace_fibsq

GEN_W = 5


def _pow(a, e):
    return pow(int(a), int(e), P)


def root_of_unity(log_order):
    return _pow(GEN_W, (P - 1) >> log_order)


# merkle.rs:54-71: node indices of the authentication path of `leaf` in a heap of m leaves
def path_nodes(m, leaf):
    i = leaf + (2 * m - 1) // 2
    out = []
    while i != 0:
        if i % 2 == 0:
            out.append(i - 1); i -= 2
        else:
            out.append(i + 1); i -= 1
        i >>= 1
    return out


def host_merkle_top(subroots):
    """merkle.rs:38-47 over G subtree roots (bytes): returns the heap of 2G-1 digests, root first."""
    g = len(subroots)
    heap = [None] * (2 * g - 1)
    heap[g - 1:] = list(subroots)
    for j in range(g - 2, -1, -1):
        heap[j] = hashlib.sha256(heap[2 * j + 1] + heap[2 * j + 2]).digest()
    return heap


def words_to_bytes(words):
    """Digest state words (native u32) -> SHA-256 byte order."""
    return np.ascontiguousarray(words, dtype=np.uint32).astype(">u4").tobytes()


class _Works:
    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()


class Comm:
    """The three collectives the path needs, over torch.distributed.  `staged` moves device tensors
    through host memory (gloo); with the nccl backend (RCCL) tensors are exchanged in place."""

    def __init__(self, group=None, staged=False, force=False, lists=None):
        import torch.distributed as dist
        self.dist, self.group, self.staged = dist, group, staged
        self.lists = lists          # None: list exchanges only where the backend has them (RCCL); True: emulate with send/recv pairs
        self.force = force          # run the collectives even with one rank (exercises RCCL on a one-GPU box)
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def all_to_all(self, send, recv):
        """Equal splits: chunk p of `send` goes to rank p; chunk q of `recv` comes from rank q."""
        if self.staged and send.is_cuda:
            s, r = send.cpu(), recv.cpu()
            self.dist.all_to_all_single(r, s, group=self.group)
            recv.copy_(r)
        else:
            self.dist.all_to_all_single(recv, send, group=self.group)
        return recv

    def all_to_all_async(self, send, recv):
        """As all_to_all, returning an object with .wait() that orders the CURRENT stream after the exchange
        (RCCL runs it on its own stream, so kernels already queued keep the GPU busy meanwhile)."""
        if self.staged and send.is_cuda:
            self.all_to_all(send, recv)
            return None
        return self.dist.all_to_all_single(recv, send, group=self.group, async_op=True)

    def lists_ok(self):
        """True when the backend exchanges lists of (non-contiguous) slices in place: RCCL does; gloo does not
        (there the caller packs and uses all_to_all, unless lists=True asks for the send/recv emulation)."""
        if self.lists is not None:
            return bool(self.lists) and not self.staged
        return not self.staged and self.dist.get_backend(self.group) == "nccl"

    def all_to_all_list_async(self, sends, recvs):
        """sends[p] goes to rank p, recvs[q] comes from rank q (any slices of device tensors, no packing)."""
        if self.dist.get_backend(self.group) == "nccl":
            return self.dist.all_to_all(recvs, sends, group=self.group, async_op=True)
        works = []                                       # backends without alltoall: pairwise, FIFO per pair
        for q in range(self.world):
            if q == self.rank:
                recvs[q].copy_(sends[q])
            else:
                works.append(self.dist.irecv(recvs[q], src=q, group=self.group))
                works.append(self.dist.isend(sends[q], dst=q, group=self.group))
        return _Works(works)

    def all_gather(self, t, out):
        """out: [world * len(t)] flat."""
        if self.staged and t.is_cuda:
            o = out.cpu()
            self.dist.all_gather_into_tensor(o, t.cpu(), group=self.group)
            out.copy_(o)
        else:
            self.dist.all_gather_into_tensor(out, t, group=self.group)
        return out


    def agree(self, ok):
        """True iff `ok` holds on every rank (one tiny all-reduce on the backend's own kind of tensor)."""
        import torch
        on_gpu = self.dist.get_backend(self.group) == "nccl"
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()) if on_gpu else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return bool(int(t.item()))

    def barrier(self):
        self.dist.barrier(group=self.group)


class RootBoard:
    """Exchange of the G subtree roots of a commitment between the ranks of ONE node through shared memory.

    A commitment ends with every rank needing all G roots (32 bytes each) on the host to hash the top of
    the tree and feed the channel (prover.rs:85).  As a device collective that is an all-gather plus a
    device-to-host read, ~100 us of fixed cost for 256 bytes; through a page in /dev/shm it is a store
    and G polled loads.  Slot = [digest 8 words | sequence number]; x86 keeps the two stores and the two
    loads in order.  Built only when every rank can map the file (same node); otherwise the caller keeps
    using the collective."""
    RING = 4                                   # a commit cannot run more than one exchange ahead of the slowest rank

    def __init__(self, comm, tag):
        import mmap
        import os
        self.G, self.rank = comm.world, comm.rank
        self.path = "/dev/shm/zkstark_amd_%s_%s_%s" % (os.environ.get("MASTER_ADDR", "local").replace("/", "_"),
                                                        os.environ.get("MASTER_PORT", "0"), tag)
        size = self.RING * self.G * 64
        self.mm = None
        ok = True
        if self.rank == 0:
            try:
                if os.path.exists(self.path):
                    os.unlink(self.path)
                fd = os.open(self.path, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
                os.ftruncate(fd, size)
                self.mm = mmap.mmap(fd, size)
                os.close(fd)
            except OSError:
                ok = False
        comm.barrier()                           # the file exists (or rank 0 failed) before anybody opens it
        if self.rank != 0:
            try:
                fd = os.open(self.path, os.O_RDWR)
                ok = os.fstat(fd).st_size == size
                if ok:
                    self.mm = mmap.mmap(fd, size)
                os.close(fd)
            except OSError:
                ok = False
        self.ok = comm.agree(ok)                 # also orders "everybody has mapped it" before the unlink
        if self.rank == 0 and os.path.exists(self.path):
            os.unlink(self.path)                 # the mapping keeps the memory alive; no file is left behind
        if not self.ok:
            self.close()
            return
        self.slots = np.ndarray((self.RING, self.G, 16), dtype=np.uint32, buffer=self.mm)

    def close(self):
        self.slots = None
        if self.mm is not None:
            try:
                self.mm.close()
            except BufferError:
                pass
            self.mm = None

    def exchange(self, seq, mine, timeout=120.0):
        """mine: this rank's 32-byte digest for exchange number seq (1, 2, ...): returns the G digests."""
        row = self.slots[seq % self.RING]
        row[self.rank, :8] = np.frombuffer(mine, dtype=np.uint32)
        row[self.rank, 8] = seq                  # after the digest (program order = store order on x86)
        out = []
        t0 = None
        for q in range(self.G):
            while int(row[q, 8]) != seq:
                if t0 is None:
                    t0 = time.perf_counter()
                elif time.perf_counter() - t0 > timeout:
                    raise ZkError(-6, f"rank {q} did not post its subtree root (exchange {seq})")
            out.append(row[q, :8].tobytes())
        return out


class LocalComm:
    """world = 1 (no process group): lets the sharded code path run in a single process."""
    rank, world, force = 0, 1, False


class HipBackend:
    """gfx950 kernels through the C ABI (include/zkstark_amd.h, zk_dev_*), on torch CUDA tensors,
    enqueued on torch's current stream so that RCCL collectives order with them."""

    def __init__(self, device):
        import torch
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.index = device
        self.lib = _lib.load()
        self._doms = []

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def domain(self, log_n, log_b, shift, fold_only=False):
        h = C.c_void_p()
        check(self.lib.zk_dom_create(self.index, log_n, log_b, shift, int(fold_only), C.byref(h)))
        self._doms.append(h)
        return h

    def close(self):
        for h in self._doms:
            self.lib.zk_dom_destroy(h)
        for h in getattr(self, "_tails", []):
            self.lib.zk_ctx_destroy(h)
        if getattr(self, "_committer", None) is not None:
            self.lib.zk_committer_destroy(self._committer)
            self._committer = None
        self._doms, self._tails = [], []

    def empty(self, nwords):
        return self.torch.empty(nwords, dtype=self.torch.int32, device=self.device)

    def upload(self, arr):
        return self.torch.from_numpy(np.ascontiguousarray(arr, dtype=np.uint32).view(np.int32)).to(self.device)

    def lde(self, dom, trace, coef, out):
        check(self.lib.zk_dev_lde(dom, trace.data_ptr(), coef.data_ptr(), out.data_ptr(), self._stream()))

    def compose(self, dom, f, cp, first, last, alphas):
        a = (C.c_uint32 * 3)(*alphas)
        check(self.lib.zk_dev_compose(dom, f.data_ptr(), cp.data_ptr(), first, last, a, self._stream()))

    def fold(self, dom, src, dst, log_m, rnd, beta):
        check(self.lib.zk_dev_fri_fold(dom, src.data_ptr(), dst.data_ptr(), log_m, rnd, beta, self._stream()))

    def interleave(self, src, dst, log_parts, log_cnt):
        check(self.lib.zk_dev_interleave(src.data_ptr(), dst.data_ptr(), log_parts, log_cnt, self._stream()))

    def merkle(self, vals, log_m, nodes):
        check(self.lib.zk_dev_merkle_build(vals.data_ptr(), log_m, nodes.data_ptr(), self._stream()))

    def merkle_chunk(self, recv, log_parts, log_cnt, nodes, log_m, chunk):
        check(self.lib.zk_dev_merkle_build_chunk(recv.data_ptr(), log_parts, log_cnt, nodes.data_ptr(), log_m, chunk, self._stream(), 0))

    def merkle_finish(self, nodes, log_m, log_chunks):
        check(self.lib.zk_dev_merkle_finish(nodes.data_ptr(), log_m, log_chunks, self._stream(), 0))

    def _committer_handle(self):
        if getattr(self, "_committer", None) is None:
            self._committer = C.c_void_p()
            check(self.lib.zk_committer_create(self.index, C.byref(self._committer)))
        return self._committer

    def commit_finish(self, nodes, log_m, log_chunks):
        """merkle_finish with the root returned as bytes (top of the tree on this thread)."""
        root = C.create_string_buffer(32)
        check(self.lib.zk_dev_merkle_commit_finish(self._committer_handle(), nodes.data_ptr(), log_m, log_chunks, self._stream(), 0, root))
        return root.raw

    def commit(self, src, log_parts, log_cnt, nodes):
        """Tree over 2^(log_parts+log_cnt) leaves (log_parts > 0: still in all-to-all order) with the root
        returned as bytes: the device stops at depth 8, this thread hashes the top (zk_dev_merkle_commit)."""
        root = C.create_string_buffer(32)
        check(self.lib.zk_dev_merkle_commit(self._committer_handle(), src.data_ptr(), log_parts, log_cnt, nodes.data_ptr(), self._stream(), 0, root))
        return root.raw

    def merkle_interleaved(self, recv, log_parts, log_cnt, nodes):
        """Tree over leaves still in all-to-all order (interleave fused into the leaf hashing)."""
        check(self.lib.zk_dev_merkle_build_interleaved(recv.data_ptr(), log_parts, log_cnt, nodes.data_ptr(), self._stream(), 0))

    def gather(self, src, offsets, words):
        """Returns a host uint32 array [len(offsets), words]."""
        torch = self.torch
        off = torch.from_numpy(np.asarray(offsets, dtype=np.int64)).to(self.device)
        out = torch.empty(len(offsets) * words, dtype=torch.int32, device=self.device)
        check(self.lib.zk_dev_gather(src.data_ptr(), off.data_ptr(), len(offsets), words, out.data_ptr(), self._stream()))
        return out

    def to_host(self, t):
        return t.cpu().numpy().view(np.uint32)

    def sync(self):
        self.torch.cuda.current_stream(self.device).synchronize()

    def side_begin(self):
        """A second stream, ordered after everything queued so far on the current one."""
        torch = self.torch
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._side.wait_event(ev)
        return self._side

    def on_side(self, fn):
        with self.torch.cuda.stream(self._side):
            return fn()

    # FRI tail in one C call (zk_tail_*): the replicated small layers need no collectives, so the
    # per-round Python and readback cost is dropped for them
    def tail_create(self, log_n_tail, log_b, shift):
        h = C.c_void_p()
        check(self.lib.zk_tail_create(self.index, log_n_tail, log_b, shift, C.byref(h)))
        self._tails = getattr(self, "_tails", []) + [h]
        return h

    def tail_run(self, tail, layer, channel, rounds, hash_kind=0):
        betas = (C.c_uint32 * max(rounds, 1))()
        roots = C.create_string_buffer(32 * (rounds + 1))
        free_term = C.c_uint32()
        check(self.lib.zk_tail_run(tail, layer.data_ptr(), self._stream(), channel._h, hash_kind, betas, roots,
                                   C.byref(free_term)))
        return list(betas)[:rounds], [roots.raw[32 * i:32 * i + 32] for i in range(rounds + 1)], free_term.value

    def tail_open(self, tail, x, rounds, log_size):
        ndig = sum(2 * (log_size - i) for i in range(rounds))
        vals = (C.c_uint32 * max(2 * rounds, 1))()
        paths = C.create_string_buffer(max(32 * ndig, 1))
        check(self.lib.zk_tail_open(tail, x, vals, paths))
        return list(vals)[:2 * rounds], paths.raw[:32 * ndig]


class ShardedProver:
    """generate_proof (prover.rs:9-293) for one proof spread over comm.world ranks."""

    def __init__(self, log_n, log_blowup, comm, backend, min_chunk_log=None, overlap_min_log=0, min_layer_log=None, use_board=True):
        default_layout = min_layer_log is None and min_chunk_log is None      # zk_shard_plan's own thresholds (0 = default)
        if min_layer_log is None:                      # an explicit chunk threshold alone decides (tests shard tiny domains)
            min_layer_log = 0
        if min_chunk_log is None:
            min_chunk_log = 14
        self.log_n, self.log_b, self.comm, self.be = log_n, log_blowup, comm, backend
        G = comm.world
        self.G, self.rank = G, comm.rank
        self.lg = G.bit_length() - 1
        if 1 << self.lg != G or self.lg > log_blowup:
            raise ZkError(-1, f"world size {G} must be a power of two dividing the blow-up {1 << log_blowup}")
        self.L = log_n + log_blowup
        self.R = log_n
        self.n, self.N, self.B = 1 << log_n, 1 << self.L, 1 << log_blowup
        # Which layers stay distributed and which are exchanged in chunks is NOT decided here: zk_shard_plan (csrc/shard.hip)
        # is the one statement of the layout, shared with the native prover (in zk_shard_options 0 selects the default, so the
        # smallest explicit threshold is 1: pieces of two leaves).
        from zkstark_amd.host import shard_plan
        # cp without an exchange (zk_shard_plan_info.cp_from_f) needs the block form of the composition from the backend
        # (OracleBackend.compose_block; the device form is internal to the library): other backends exchange cp as rounds 1-4 did
        plan = shard_plan(G, log_n, log_blowup, min_layer_log=0 if default_layout else max(min_layer_log, 1), min_chunk_log=max(min_chunk_log, 1),
                          overlap_min_log=overlap_min_log, force_collectives=bool(getattr(comm, "force", False)),
                          exchange_cp=not hasattr(backend, "compose_block"))
        self.plan = plan
        self.cp_from_f = bool(plan["cp_from_f"])
        self.n_sharded = plan["sharded_layers"]
        self.chunked_mask = plan["chunked_mask"]
        h = root_of_unity(self.L)
        self.shift = GEN_W * _pow(h, self.rank) % P
        be = self.be
        self.dom_loc = be.domain(log_n, log_blowup - self.lg, self.shift)
        # the replicated tail (layers rho >= n_sharded) runs as one C call when the backend offers it:
        # layer rho0 is layer 0 of the domain with n' = n >> rho0 and shift w^(2^rho0)
        self.tail_rounds = self.R - self.n_sharded
        self.tail = None
        if hasattr(be, "tail_create") and self.tail_rounds >= 1:
            self.tail = be.tail_create(self.tail_rounds, log_blowup, _pow(GEN_W, 1 << self.n_sharded))
        self.dom_glob = be.domain(log_n, log_blowup, GEN_W, fold_only=True) if (self.n_sharded <= self.R and self.tail is None) or self.cp_from_f else None
        # one allocation for layers, one for trees (as the single-GPU context)
        NL = self.N // G
        self.layer_off, self.layer_len, off = [], [], 0
        sizes = [NL] + [self._layer_words(rho) for rho in range(self.R + 1)]
        for s in sizes:
            self.layer_off.append(off); self.layer_len.append(s); off += s
        self.layers = be.empty(off)
        self.tree_off, off = [], 0
        for t in range(self.R + 2):
            self.tree_off.append(off)
            off += (2 * self._tree_leaves(t) - 1) * 8
        self.trees = be.empty(off)
        self.trace = be.empty(self.n)
        self.coef = be.empty(2 * self.n)
        self.recv = be.empty(NL)
        self.block = be.empty(NL)
        self.subroot_all = be.empty(8 * G)
        self.log_chunks = plan["log_chunks"]                     # chunked layers (plan["chunked_mask"]): 2^log_chunks chunks
        # the first replicated layer arrives as G cyclic pieces before it is interleaved
        self.gbuf = be.empty(max(1, self.N >> self.n_sharded)) if self.n_sharded <= self.R else None
        self.have_trace = False
        # subtree roots travel through shared memory when all ranks sit on one node (RootBoard)
        self.board, self.n_exchanges = None, 0
        if (G > 1 or comm.force) and use_board and hasattr(comm, "agree"):
            ShardedProver._instances += 1
            board = RootBoard(comm, ShardedProver._instances)
            self.board = board if board.ok else None

    _instances = 0

    # layer ids: 0 = f, 1 + rho = FRI layer rho
    def _sharded(self, rho):
        return rho < self.n_sharded

    def _layer_words(self, rho):
        m = self.N >> rho
        return m // self.G if self._sharded(rho) else m

    def _tree_leaves(self, t):
        if t == 0:
            return self.N // self.G
        m = self.N >> (t - 1)
        return m // self.G if self._sharded(t - 1) else m

    def _layer(self, lid):
        return self.layers[self.layer_off[lid]:self.layer_off[lid] + self.layer_len[lid]]

    def _tree(self, t):
        return self.trees[self.tree_off[t]:self.tree_off[t] + (2 * self._tree_leaves(t) - 1) * 8]

    def trace_upload(self, trace):
        t = np.ascontiguousarray(trace, dtype=np.uint32)
        if len(t) != self.n - 1:
            raise ZkError(-1, f"expected n-1 = {self.n - 1} trace values")
        self.trace.copy_(self.be.upload(np.concatenate([t, np.zeros(1, dtype=np.uint32)])))
        self.first, self.last = int(t[0]), int(t[-1])
        self.have_trace = True

    # ---- commitments ----------------------------------------------------------------------
    def _commit_sharded(self, lid, m_log):
        """Cyclic layer of 2^m_log values in total -> subtree over this rank's block; returns the root."""
        be, G, lg = self.be, self.G, self.lg
        loc = self._layer(lid)
        cnt = loc.numel()
        recv, block = self.recv[:cnt], self.block[:cnt]
        nodes = self._tree(lid)
        log_cnt = m_log - 2 * lg                                  # words per (rank, peer) chunk
        if (G > 1 or self.comm.force) and hasattr(be, "merkle_chunk") and (self.chunked_mask >> lid) & 1:
            # big layer: exchange and hash in K aligned chunks, so that hashing chunk c overlaps the
            # all-to-all of chunk c+1 (the exchange runs on RCCL's stream)
            K, lk = 1 << self.log_chunks, self.log_chunks
            cc = cnt >> (lg + lk)                                 # words per (peer, chunk)
            per = cnt >> lg                                       # words this rank holds for each peer
            lists = hasattr(self.comm, "lists_ok") and self.comm.lists_ok()
            if not lists:                                         # backends without list exchange: pack chunk-major first
                send = self.block[:cnt]
                send.view(K, G, cc).copy_(loc.view(G, K, cc).transpose(0, 1))
            # the exchanges are issued from a side stream that depends on the layer only, so a chunk's exchange
            # never queues behind the hashing of an earlier chunk; issue order keeps two exchanges ahead
            side = be.side_begin() if hasattr(be, "side_begin") and not getattr(self.comm, "staged", False) else None
            works = [None] * K

            def exchange(c):
                if lists:
                    return self.comm.all_to_all_list_async([loc[g * per + c * cc:g * per + (c + 1) * cc] for g in range(G)],
                                                           [recv[(c * G + q) * cc:(c * G + q + 1) * cc] for q in range(G)])
                sl = slice(c * G * cc, (c + 1) * G * cc)
                return self.comm.all_to_all_async(send[sl], recv[sl])

            def post(c):
                works[c] = be.on_side(lambda: exchange(c)) if side is not None else exchange(c)
            for c in range(min(2, K)):
                post(c)
            for c in range(K):
                if works[c] is not None:
                    works[c].wait()
                be.merkle_chunk(recv[c * G * cc:(c + 1) * G * cc], lg, log_cnt - lk, nodes, m_log - lg, c)
                if c + 2 < K:
                    post(c + 2)
            if hasattr(be, "commit_finish"):
                mine = be.commit_finish(nodes, m_log - lg, lk)
            else:
                be.merkle_finish(nodes, m_log - lg, lk)
                mine = None
        elif G > 1 or self.comm.force:
            self.comm.all_to_all(loc, recv)                       # chunk q: rank q's j in my block
            mine = None
            if hasattr(be, "commit"):                             # leaf u*G + q = recv[q][u], hashed in place; root on the host
                mine = be.commit(recv, lg, m_log - 2 * lg, nodes)
            elif hasattr(be, "merkle_interleaved"):
                be.merkle_interleaved(recv, lg, m_log - 2 * lg, nodes)
            else:
                be.interleave(recv, block, lg, m_log - 2 * lg)    # block[u*G + q] = recv[q][u]
                be.merkle(block, m_log - lg, nodes)
        elif hasattr(be, "commit"):
            mine = be.commit(loc, 0, m_log, nodes)
        else:
            be.merkle(loc, m_log - lg, nodes)
            mine = None
        return self._join_subtrees(lid, nodes, mine)

    def _join_subtrees(self, lid, nodes, mine):
        """The G subtree roots on the host of every rank (shared page or all-gather), the top log2 G levels hashed by every rank."""
        be, G = self.be, self.G
        if G > 1 or self.comm.force:
            if self.board is not None:
                if mine is None:
                    mine = words_to_bytes(be.to_host(nodes[:8]))
                self.n_exchanges += 1
                subroots = self.board.exchange(self.n_exchanges, mine)
            else:
                self.comm.all_gather(nodes[:8], self.subroot_all)
                subroots = [words_to_bytes(w) for w in be.to_host(self.subroot_all).reshape(G, 8)]
        else:
            subroots = [mine if mine is not None else words_to_bytes(be.to_host(nodes[:8]))]
        top = host_merkle_top(subroots)
        self.tops[lid] = top
        return top[0]

    # ---- cp without an exchange (csrc/shard.hip: commit_cp_from_f) -------------------------------------------
    def _halo_pack(self):
        """What the neighbour of every block needs of this rank's cyclic shard of f: the values at the first 2B positions
        after block q are i = (q + 1) M + v, v < 2B; this rank holds those with v = r (mod G)."""
        G, M = self.G, self.N // self.G
        per, h = M // G, 2 * self.B // G
        loc = self.be.to_host(self._layer(0))
        send = np.array([loc[((q + 1) * per + u) % M] for q in range(G) for u in range(h)], dtype=np.uint32)
        return self.be.upload(send)

    def _block_of_f(self):
        """This rank's block of f in natural order, from the receive buffer the commitment of f left behind (all-to-all order,
        in chunks when the layer was exchanged in chunks)."""
        be, G, lg = self.be, self.G, self.lg
        M = self.N // G
        recv = be.to_host(self.recv[:M])
        if (G > 1 or self.comm.force) and hasattr(be, "merkle_chunk") and self.chunked_mask & 1:
            K = 1 << self.log_chunks
            return recv.reshape(K, G, M // (K * G)).transpose(0, 2, 1).reshape(-1)     # chunk c: leaf u G + q = recv[c][q][u]
        return recv.reshape(G, M // G).T.reshape(-1)

    def _commit_cp_from_f(self, halo_send, alphas):
        be, G, lg, B = self.be, self.G, self.lg, self.B
        M, h = self.N // G, 2 * self.B // G
        gath = be.empty(2 * B * G)
        self.comm.all_gather(halo_send, gath)
        gath = be.to_host(gath)
        halo = np.array([gath[(v % G) * 2 * B + self.rank * h + v // G] for v in range(2 * B)], dtype=np.uint32)
        fext = np.concatenate([self._block_of_f(), halo])
        cp_blk = be.empty(M)
        be.compose_block(self.dom_glob, fext, self.rank * M, cp_blk, self.first, self.last, alphas)
        nodes = self._tree(1)
        be.merkle(cp_blk, self.L - lg, nodes)
        return self._join_subtrees(1, nodes, None)

    def _commit_replicated(self, lid, m_log):
        nodes = self._tree(lid)
        self.be.merkle(self._layer(lid), m_log, nodes)
        return words_to_bytes(self.be.to_host(nodes[:8]))

    def lde_commit(self):
        """BASELINE.json configs[3] shape: the sharded LDE (each rank its cosets) followed by the all-to-all
        transpose and the Merkle commitment; returns the root of f_eval (prover.rs:60-85)."""
        if not self.have_trace:
            raise ZkError(-4, "no trace uploaded")
        self.tops = {}
        self.be.lde(self.dom_loc, self.trace, self.coef, self._layer(0))
        return self._commit_sharded(0, self.L)

    # ---- the prover ---------------------------------------------------------------------------
    def prove(self):
        if not self.have_trace:
            raise ZkError(-4, "no trace uploaded")
        be, G, lg, L, R, B, N = self.be, self.G, self.lg, self.L, self.R, self.B, self.N
        ch = Channel()
        self.tops = {}
        roots = []
        be.lde(self.dom_loc, self.trace, self.coef, self._layer(0))                 # prover.rs:60-70
        halo_send = self._halo_pack() if self.cp_from_f else None
        roots.append(self._commit_sharded(0, L)); ch.commit(roots[-1])              # prover.rs:81-85
        alphas = [ch.get_u32() for _ in range(3)]                                   # prover.rs:163-165
        be.compose(self.dom_loc, self._layer(0), self._layer(1), self.first, self.last, alphas)   # :166-173 (cyclic: the folds' input)
        if self.cp_from_f:                                                          # over the block, from the received block of f
            roots.append(self._commit_cp_from_f(halo_send, alphas))
        else:
            roots.append(self._commit_sharded(1, L))
        ch.commit(roots[-1])                                                        # prover.rs:176-180
        betas = []
        for rho in range(R):                                                        # prover.rs:198-225
            beta = ch.get_u32(); betas.append(beta)
            m_log = L - rho
            src, dst_id = self._layer(1 + rho), 2 + rho
            if self._sharded(rho + 1):
                be.fold(self.dom_loc, src, self._layer(dst_id), m_log - lg, rho, beta)
                root = self._commit_sharded(dst_id, m_log - 1)
            elif self._sharded(rho):
                # replication switch: fold locally, all-gather the G cyclic pieces, interleave to natural order
                cnt = (1 << (m_log - 1)) // G
                piece = self.recv[:cnt]
                be.fold(self.dom_loc, src, piece, m_log - lg, rho, beta)
                if G > 1 or self.comm.force:
                    gathered = self.gbuf[:cnt * G]
                    self.comm.all_gather(piece, gathered)
                    be.interleave(gathered, self._layer(dst_id), lg, m_log - 1 - lg)
                else:
                    self._layer(dst_id).copy_(piece)
                if self.tail is not None:
                    # hand the first replicated layer over: its commitment and every later round in one C call
                    tb, tr, free_term = be.tail_run(self.tail, self._layer(dst_id), ch, self.tail_rounds)
                    betas.extend(tb); roots.extend(tr)
                    break
                root = self._commit_replicated(dst_id, m_log - 1)
            else:
                be.fold(self.dom_glob, src, self._layer(dst_id), m_log, rho, beta)
                root = self._commit_replicated(dst_id, m_log - 1)
            roots.append(root); ch.commit(root)                                     # prover.rs:224
        # last layer: B equal values (prover.rs:238, :251, :254)
        if self._sharded(R):
            raise ZkError(-4, "last FRI layer still sharded: lower min_chunk_log only with a tiny world")
        if self.tail is None:
            last = be.to_host(self._layer(1 + R))
            if not (last == last[0]).all():
                raise ZkError(-7, "last FRI layer is not constant (prover.rs:238)")
            free_term = int(last[0])
        ch.commit(free_term)                                                        # prover.rs:254
        x = ch.get_u32() % (N - 2 * B)                                              # prover.rs:263
        self.transcript = {"alpha_raw": alphas, "beta_raw": betas, "roots": roots, "free_term": free_term, "query": x}
        self._decommit(ch, x)
        return Proof(ch.state, ch.data, self.log_n, self.log_b, self.last)

    def _decommit(self, ch, x):
        """prover.rs:266-289.  Every rank gathers the slots it owns (same slot list everywhere), one
        all-gather merges them, every rank assembles the same bytes."""
        be, G, lg, L, R, B, N = self.be, self.G, self.lg, self.L, self.R, self.B, self.N
        val_items, dig_items = [], []      # (owner or -1, offset)
        openings = []                      # (n_local_digests, top_path) per opening; tail openings come later
        rho0 = self.n_sharded
        if self.tail is not None:
            tvals, tpaths = be.tail_open(self.tail, x, self.tail_rounds, L - rho0)

        def add_opening(lid, leaf, m_log, sharded):
            if sharded:
                owner_v, off_v = leaf % G, self.layer_off[lid] + leaf // G
                blk = (1 << m_log) // G
                p, lf = leaf // blk, leaf % blk
                nodes = path_nodes(blk, lf)
                for nd in nodes:
                    dig_items.append((p, self.tree_off[lid] + nd * 8))
                top_path = [self.tops[lid][j] for j in path_nodes(G, p)]
            else:
                owner_v, off_v = -1, self.layer_off[lid] + leaf
                nodes = path_nodes(1 << m_log, leaf)
                for nd in nodes:
                    dig_items.append((-1, self.tree_off[lid] + nd * 8))
                top_path = []
            val_items.append((owner_v, off_v))
            openings.append((len(nodes), top_path))

        for lid, leaf in ((0, x), (0, x + B), (0, x + 2 * B), (1, x)):               # prover.rs:266-277
            add_opening(lid, leaf, L, True)
        n_gathered_layers = rho0 if self.tail is not None else R
        for i in range(n_gathered_layers):                                            # prover.rs:280-289
            ln = N >> i
            xi = x % ln
            nx = (xi + ln // 2) % ln
            add_opening(1 + i, xi, L - i, self._sharded(i))
            add_opening(1 + i, nx, L - i, self._sharded(i))
        me = self.rank
        voff = [off if o in (-1, me) else 0 for o, off in val_items]
        doff = [off if o in (-1, me) else 0 for o, off in dig_items]
        vals = be.gather(self.layers, voff, 1)
        digs = be.gather(self.trees, doff, 8)
        nv, nd = len(voff), len(doff)
        if G > 1 or self.comm.force:
            import torch
            mine = torch.cat([vals, digs])
            allc = be.empty(mine.numel() * G)
            self.comm.all_gather(mine, allc)
            allh = be.to_host(allc).reshape(G, nv + 8 * nd)
        else:
            allh = np.concatenate([be.to_host(vals), be.to_host(digs)]).reshape(1, nv + 8 * nd)
        # pick every slot from its owner's row (vectorised), digests straight to SHA-256 byte order
        vo = np.array([me if o == -1 else o for o, _ in val_items], dtype=np.int64)
        V = allh[vo, np.arange(nv)]
        do = np.array([me if o == -1 else o for o, _ in dig_items], dtype=np.int64)
        cols = nv + 8 * np.arange(nd)[:, None] + np.arange(8)[None, :]
        dbytes = allh[do[:, None], cols].astype(">u4").tobytes()       # nd * 32 bytes
        dpos = 0
        blobs = []
        for nloc, top_path in openings:
            plen = nloc + len(top_path)
            blobs.append(struct.pack("<Q", plen) + dbytes[32 * dpos:32 * (dpos + nloc)] + b"".join(top_path))
            dpos += nloc
        lib, h = _lib.load(), ch._h
        for k in range(4):                                                            # prover.rs:274-277
            b = struct.pack("<I", int(V[k])) + blobs[k]
            check(lib.zk_channel_commit(h, b, len(b)))
        for i in range(n_gathered_layers):                                            # prover.rs:288
            b = struct.pack("<II", int(V[4 + 2 * i]), int(V[5 + 2 * i])) + blobs[4 + 2 * i] + blobs[5 + 2 * i]
            check(lib.zk_channel_commit(h, b, len(b)))
        if self.tail is not None:                                                     # tail layers, from zk_tail_open
            tp = 0
            for j in range(self.tail_rounds):
                pl = L - rho0 - j
                b = (struct.pack("<II", tvals[2 * j], tvals[2 * j + 1])
                     + struct.pack("<Q", pl) + tpaths[32 * tp:32 * (tp + pl)]
                     + struct.pack("<Q", pl) + tpaths[32 * (tp + pl):32 * (tp + 2 * pl)])
                tp += 2 * pl
                check(lib.zk_channel_commit(h, b, len(b)))

    def close(self):
        if self.board is not None:
            self.board.close()
            self.board = None
        self.be.close()
