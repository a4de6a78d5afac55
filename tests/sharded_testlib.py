"""Test double for zkstark_amd.sharded: a CPU compute backend with the HipBackend interface, written
directly from the definitions (generic coset shift) with numpy / Python ints and the oracle's NTT and
Merkle routines.  TEST INFRASTRUCTURE ONLY: the product never imports this."""
import numpy as np
import torch

import oracle

P = oracle.P


def _w2(nodes_bytes):
    """oracle nodes (bytes, SHA order) -> big-endian state words as the device stores them."""
    return np.frombuffer(np.ascontiguousarray(nodes_bytes).tobytes(), dtype=">u4").astype(np.uint32)


class OracleBackend:
    def __init__(self):
        self.calls = {"lde": 0, "compose": 0, "fold": 0, "merkle": 0, "interleave": 0, "gather": 0}

    def domain(self, log_n, log_b, shift, fold_only=False):
        L = log_n + log_b
        return {"log_n": log_n, "log_b": log_b, "L": L, "shift": shift, "h": pow(5, (P - 1) >> L, P),
                "g": pow(5, (P - 1) >> log_n, P)}

    def close(self):
        pass

    def empty(self, nwords):
        return torch.zeros(nwords, dtype=torch.int32)

    def upload(self, arr):
        return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.uint32).view(np.int32).copy())

    @staticmethod
    def _np(t):
        return t.numpy().view(np.uint32)

    def lde(self, dom, trace, coef, out):
        self.calls["lde"] += 1
        n, N = 1 << dom["log_n"], 1 << dom["L"]
        a = self._np(trace)[:n - 1]
        y = np.concatenate([a, [oracle.virtual_point(a, dom["log_n"])]]).astype(np.uint32)
        c = oracle.intt(y, dom["g"])                       # unique interpolant through the n-1 points
        assert c[n - 1] == 0
        s, sk = dom["shift"], 1
        pad = np.zeros(N, dtype=np.uint32)
        for k in range(n):
            pad[k] = int(c[k]) * sk % P
            sk = sk * s % P
        self._np(out)[:] = oracle.ntt(pad, dom["h"])     # f(shift * h^i)

    def compose(self, dom, f, cp, first, last, alphas):
        self.calls["compose"] += 1
        n, N, B = 1 << dom["log_n"], 1 << dom["L"], 1 << dom["log_b"]
        g, h, s = dom["g"], dom["h"], dom["shift"]
        fv = [int(v) for v in self._np(f)]
        a0, a1, a2 = [int(a) % P for a in alphas]
        inv = lambda v: pow(v, P - 2, P)
        gm1 = inv(g); gm2 = gm1 * gm1 % P; gm3 = gm2 * gm1 % P
        out = self._np(cp)
        x = s
        for i in range(N):
            f0, f1, f2 = fv[i], fv[(i + B) % N], fv[(i + 2 * B) % N]
            p0 = (f0 - first) * inv((x - 1) % P) % P
            p1 = (f0 - last) * inv((x - gm2) % P) % P
            num = (f2 - f1 * f1 - f0 * f0) % P
            den = (pow(x, n, P) - 1) * inv((x - gm3) * (x - gm2) * (x - gm1) % P) % P
            out[i] = (a0 * p0 + a1 * p1 + a2 * num * inv(den)) % P
            x = x * h % P

    def compose_block(self, dom, fext, e0, out, first, last, alphas):
        """cp at the positions e0 .. e0 + M - 1 of the GLOBAL domain `dom` (prover.rs:101-166 pointwise), from f at those
        positions and the 2B after them (`fext`: M + 2B values in natural order).  What ComposeBlockSrc computes on the device."""
        self.calls["compose_block"] = self.calls.get("compose_block", 0) + 1
        n, B = 1 << dom["log_n"], 1 << dom["log_b"]
        g, h, s = dom["g"], dom["h"], dom["shift"]
        fv = [int(v) for v in np.asarray(fext, dtype=np.uint32)]
        M = len(fv) - 2 * B
        a0, a1, a2 = [int(a) % P for a in alphas]
        inv = lambda v: pow(v, P - 2, P)
        gm1 = inv(g); gm2 = gm1 * gm1 % P; gm3 = gm2 * gm1 % P
        res = self._np(out)
        x = s * pow(h, e0, P) % P
        for t in range(M):
            f0, f1, f2 = fv[t], fv[t + B], fv[t + 2 * B]
            p0 = (f0 - first) * inv((x - 1) % P) % P
            p1 = (f0 - last) * inv((x - gm2) % P) % P
            num = (f2 - f1 * f1 - f0 * f0) % P
            den = (pow(x, n, P) - 1) * inv((x - gm3) * (x - gm2) * (x - gm1) % P) % P
            res[t] = (a0 * p0 + a1 * p1 + a2 * num * inv(den)) % P
            x = x * h % P

    def fold(self, dom, src, dst, log_m, rnd, beta):
        self.calls["fold"] += 1
        assert log_m + rnd == dom["L"]
        half = 1 << (log_m - 1)
        e = [int(v) for v in self._np(src)[:2 * half]]
        out = self._np(dst)
        inv2 = pow(2, P - 2, P)
        b = int(beta) % P
        x = pow(dom["shift"], 1 << rnd, P)
        step = pow(dom["h"], 1 << rnd, P)
        for i in range(half):
            out[i] = ((e[i] + e[i + half]) * inv2 + b * (e[i] - e[i + half]) % P * pow(2 * x % P, P - 2, P)) % P
            x = x * step % P

    def interleave(self, src, dst, log_parts, log_cnt):
        self.calls["interleave"] += 1
        parts, cnt = 1 << log_parts, 1 << log_cnt
        self._np(dst)[:parts * cnt] = self._np(src)[:parts * cnt].reshape(parts, cnt).T.reshape(-1)

    def merkle(self, vals, log_m, nodes):
        self.calls["merkle"] += 1
        m = 1 << log_m
        self._np(nodes)[:(2 * m - 1) * 8] = _w2(oracle.merkle_build(self._np(vals)[:m]))

    def gather(self, src, offsets, words):
        self.calls["gather"] += 1
        s = self._np(src)
        out = np.zeros(len(offsets) * words, dtype=np.uint32)
        for k, o in enumerate(offsets):
            out[k * words:(k + 1) * words] = s[o:o + words]
        return torch.from_numpy(out.view(np.int32))

    def to_host(self, t):
        return t.numpy().view(np.uint32).copy()

    def sync(self):
        pass


class ChunkingOracleBackend(OracleBackend):
    """Adds the chunked commitment of HipBackend (merkle_chunk / merkle_finish), so that the chunked exchange of
    zkstark_amd.sharded runs with real multi-rank collectives on CPU."""

    def merkle_chunk(self, recv, log_parts, log_cnt, nodes, log_m, chunk):
        import hashlib  # noqa: F401
        self.calls["merkle_chunk"] = self.calls.get("merkle_chunk", 0) + 1
        parts, cnt = 1 << log_parts, 1 << log_cnt
        log_s = log_parts + log_cnt
        leaves = self._np(recv)[:parts * cnt].reshape(parts, cnt).T.reshape(-1)      # leaf u*parts + q = recv[q][u]
        sub = _w2(oracle.merkle_build(np.ascontiguousarray(leaves))).reshape(-1, 8)    # heap of the chunk alone
        heap = self._np(nodes)
        top = log_m - log_s                                                           # depth of the chunk root in the block tree
        for d in range(log_s + 1):
            src = sub[(1 << d) - 1:(2 << d) - 1]
            first = (1 << (top + d)) - 1 + (chunk << d)
            heap[first * 8:(first + (1 << d)) * 8] = src.reshape(-1)

    def merkle_finish(self, nodes, log_m, log_chunks):
        import hashlib
        self.calls["merkle_finish"] = self.calls.get("merkle_finish", 0) + 1
        heap = self._np(nodes)
        for d in range(log_chunks - 1, -1, -1):
            for i in range(1 << d):
                j = (1 << d) - 1 + i
                kids = heap[(2 * j + 1) * 8:(2 * j + 3) * 8].astype(">u4").tobytes()
                heap[j * 8:(j + 1) * 8] = np.frombuffer(hashlib.sha256(kids).digest(), dtype=">u4").astype(np.uint32)


# ---- a caller-supplied zk_shard_transport for tests: gloo, staged through host memory ----------------------
def gloo_transport(dist_group=None):
    """zkstark_amd.sharded.staged_transport: lets several ranks share ONE GPU (RCCL needs a GPU per rank)."""
    from zkstark_amd import sharded
    return sharded.staged_transport(dist_group)
