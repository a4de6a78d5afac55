// transcript.hpp -- host-only Fiat-Shamir channel, proof wire format and verifier.
//
// Channel mirrors channel.rs:6-37; the byte encoding is bincode 1.x defaults
// (little-endian fixed-width ints, [u8;32] raw, Box<[T]> = u64 count + items)
// as read back by proof.rs:16-46.  The reference holds no golden bytes for the
// transcript, so this encoding is "parity unpinned" (SURVEY.md section 8c).
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

#include "field.hpp"
#include "fieldhash.hpp"
#include "sha256.hpp"

namespace zk {

struct Channel {
    uint8_t state[32];
    std::vector<uint8_t> data;
    Channel() { memset(state, 0, 32); }   // channel.rs:12-17
    // channel.rs:19-26
    void commit_bytes(const uint8_t* b, size_t n) {
        Sha256 h;
        h.update(state, 32);
        h.update(b, n);
        h.finalize(state);
        data.insert(data.end(), b, b + n);
    }
    void commit_hash(const uint8_t h[32]) { commit_bytes(h, 32); }
    void commit_u32(uint32_t v) {
        uint8_t b[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)};
        commit_bytes(b, 4);
    }
    // channel.rs:28-32
    uint32_t get_u32() {
        uint32_t f = ((uint32_t)state[0] << 24) | ((uint32_t)state[1] << 16) | ((uint32_t)state[2] << 8) | state[3];
        commit_u32(f);
        return f;
    }
    static void put32(std::vector<uint8_t>& v, uint32_t x) { for (int i = 0; i < 4; ++i) v.push_back((uint8_t)(x >> (8 * i))); }
    static void put64(std::vector<uint8_t>& v, uint64_t x) { for (int i = 0; i < 8; ++i) v.push_back((uint8_t)(x >> (8 * i))); }
    // (u32, AuthPath): prover.rs:274-277
    void commit_val_path(uint32_t val, const uint8_t* path, size_t plen) {
        std::vector<uint8_t> b;
        put32(b, val); put64(b, plen);
        b.insert(b.end(), path, path + 32 * plen);
        commit_bytes(b.data(), b.size());
    }
    // (u32, u32, AuthPath, AuthPath): prover.rs:288
    void commit_pair_paths(uint32_t v0, uint32_t v1, const uint8_t* p0, const uint8_t* p1, size_t plen) {
        std::vector<uint8_t> b;
        put32(b, v0); put32(b, v1);
        put64(b, plen); b.insert(b.end(), p0, p0 + 32 * plen);
        put64(b, plen); b.insert(b.end(), p1, p1 + 32 * plen);
        commit_bytes(b.data(), b.size());
    }
};

// q = number of decommitment queries (1 = the reference's format, prover.rs:263; q > 1: SURVEY.md 8f
// item 1 -- the q raw indices are drawn in a row, then each query's openings are committed in turn).
inline size_t proof_data_len(uint32_t log_n, uint32_t log_b, uint32_t q = 1) {
    size_t L = log_n + log_b, R = log_n;
    size_t per_query = 4 + 4 * (4 + 8 + 32 * L);
    for (size_t i = 0; i < R; ++i) per_query += 8 + 2 * (8 + 32 * (L - i));
    return 32 + 12 + 32 + R * 36 + 4 + (size_t)q * per_query;
}

// Merkle hash on the host (verifier): hash 0 = SHA-256 (merkle.rs:30-34, :42-45), 1 = field-native (fieldhash.hpp)
inline const FieldHashConsts& host_fieldhash_consts() {
    static const FieldHashConsts c = [] { FieldHashConsts t; fieldhash_make_consts(t); return t; }();
    return c;
}
inline void bytes_to_digest(const uint8_t* b, Digest& d) {
    for (int i = 0; i < 8; ++i) d.w[i] = ((uint32_t)b[4 * i] << 24) | ((uint32_t)b[4 * i + 1] << 16) | ((uint32_t)b[4 * i + 2] << 8) | b[4 * i + 3];
}
inline void host_leaf_hash(uint32_t element, uint8_t out[32], int hash) {
    if (hash) { digest_words_to_bytes(fieldhash_leaf(element, host_fieldhash_consts()).w, out); return; }
    uint8_t be[4] = {(uint8_t)(element >> 24), (uint8_t)(element >> 16), (uint8_t)(element >> 8), (uint8_t)element};
    Sha256 h; h.update(be, 4); h.finalize(out);
}
inline void host_node_hash(const uint8_t* l, const uint8_t* r, uint8_t out[32], int hash) {
    if (hash) {
        Digest dl, dr;
        bytes_to_digest(l, dl); bytes_to_digest(r, dr);
        digest_words_to_bytes(fieldhash_inner(dl, dr, host_fieldhash_consts()).w, out);
        return;
    }
    Sha256 h; h.update(l, 32); h.update(r, 32); h.finalize(out);
}

// merkle.rs:82-110
inline void compute_root_from_path(uint32_t element, size_t index, const uint8_t* path, size_t plen, uint8_t out[32], int hash = 0) {
    index += ((size_t)1 << plen) - 1;
    uint8_t cur[32], nxt[32];
    host_leaf_hash(element, cur, hash);
    for (size_t k = 0; k < plen; ++k) {
        if (index % 2 == 0) { host_node_hash(path + 32 * k, cur, nxt, hash); index -= 2; }
        else { host_node_hash(cur, path + 32 * k, nxt, hash); index -= 1; }
        memcpy(cur, nxt, 32);
        index >>= 1;
    }
    memcpy(out, cur, 32);
}

// proof.rs:15-149 with the literals generalised.  Returns 0 or the negative index of the failed check.
inline int verify_proof(const uint8_t* data, size_t len, uint32_t log_n, uint32_t log_b, uint32_t public_last, int hash = 0, uint32_t q = 1) {
    if (log_n < 2 || log_b < 1 || log_n + log_b > 30 || q < 1 || q > 64) return -1;
    const size_t n = (size_t)1 << log_n, B = (size_t)1 << log_b, N = n << log_b, R = log_n, L = log_n + log_b;
    const uint8_t* p = data;
    size_t left = len;
    bool bad = false;
    auto take = [&](size_t k) -> const uint8_t* {
        if (left < k) { bad = true; return nullptr; }
        const uint8_t* q = p; p += k; left -= k; return q;
    };
    auto take32 = [&]() -> uint32_t {
        const uint8_t* q = take(4);
        return q ? ((uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24)) : 0;
    };
    auto take_path = [&](size_t& plen) -> const uint8_t* {
        const uint8_t* q = take(8);
        if (!q) return nullptr;
        uint64_t c = 0;
        for (int i = 0; i < 8; ++i) c |= (uint64_t)q[i] << (8 * i);
        if (c > 64) { bad = true; return nullptr; }
        plen = (size_t)c;
        return take(32 * plen);
    };
    // proof.rs:20-46
    const uint8_t* f_root = take(32);
    uint32_t alpha[3] = {take32(), take32(), take32()};
    const uint8_t* roots[40]; uint32_t betas[40];
    roots[0] = take(32); betas[0] = 0;
    for (size_t i = 0; i < R; ++i) { betas[i + 1] = take32(); roots[i + 1] = take(32); }
    uint32_t free_term = take32();
    uint32_t test_raws[64];
    for (uint32_t k = 0; k < q; ++k) test_raws[k] = take32();
    const uint32_t g = root_of_unity(log_n), h = root_of_unity((uint32_t)L);
    auto fsub = [](uint32_t a, uint32_t b) { return sub(a, b); };
    for (uint32_t qk = 0; qk < q; ++qk) {
    const uint32_t test_raw = test_raws[qk];
    uint32_t fv[4]; const uint8_t* fp[4]; size_t fpl[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) { fv[i] = take32(); fp[i] = take_path(fpl[i]); }
    uint32_t lx[40], lnx[40]; const uint8_t *lpx[40], *lpnx[40]; size_t plx[40], plnx[40];
    for (size_t i = 0; i < R; ++i) {
        lx[i] = take32(); lnx[i] = take32(); plx[i] = plnx[i] = 0;
        lpx[i] = take_path(plx[i]); lpnx[i] = take_path(plnx[i]);
    }
    if (bad) return -1;
    // proof.rs:49-60
    const size_t tp = (size_t)test_raw % (N - 2 * B);
    const uint32_t x = mulmod(GEN_W, powmod(h, tp));
    {   // proof.rs:63-77
        uint32_t f_x = fv[0] % P, f_gx = fv[1] % P, f_ggx = fv[2] % P;
        uint32_t gm1 = invmod(g), gm2 = mulmod(gm1, gm1), gm3 = mulmod(gm2, gm1);
        uint32_t p0 = mulmod(fsub(f_x, 1), invmod(fsub(x, 1)));
        uint32_t p1 = mulmod(fsub(f_x, public_last % P), invmod(fsub(x, gm2)));
        uint32_t num = fsub(fsub(f_ggx, mulmod(f_gx, f_gx)), mulmod(f_x, f_x));
        uint32_t den = mulmod(fsub(powmod(x, n), 1), invmod(mulmod(mulmod(fsub(x, gm3), fsub(x, gm2)), fsub(x, gm1))));
        uint32_t p2 = mulmod(num, invmod(den));
        uint32_t cp0 = add(add(mulmod(alpha[0] % P, p0), mulmod(alpha[1] % P, p1)), mulmod(alpha[2] % P, p2));
        if (cp0 != fv[3]) return -2;
    }
    uint8_t root[32];
    if (fpl[0] != L || fpl[1] != L || fpl[2] != L || fpl[3] != L) return -3;
    // proof.rs:80-95
    compute_root_from_path(fv[0], tp, fp[0], fpl[0], root, hash);         if (memcmp(root, f_root, 32)) return -4;
    compute_root_from_path(fv[1], tp + B, fp[1], fpl[1], root, hash);     if (memcmp(root, f_root, 32)) return -5;
    compute_root_from_path(fv[2], tp + 2 * B, fp[2], fpl[2], root, hash); if (memcmp(root, f_root, 32)) return -6;
    compute_root_from_path(fv[3], tp, fp[3], fpl[3], root, hash);         if (memcmp(root, roots[0], 32)) return -7;
    // proof.rs:101-126
    const uint32_t inv2 = invmod(2);
    for (size_t k = 0; k < R; ++k) {
        uint32_t xk = powmod(x, (uint64_t)1 << k);
        uint32_t gx = mulmod(add(lx[k] % P, lnx[k] % P), inv2);
        uint32_t hx = mulmod(fsub(lx[k] % P, lnx[k] % P), invmod(mulmod(xk, 2)));
        uint32_t calc = add(gx, mulmod(betas[k + 1] % P, hx));
        uint32_t expect = (k + 1 < R) ? lx[k + 1] : free_term;
        if (calc != expect) return -(int)(100 + k);
    }
    // proof.rs:129-148
    for (size_t k = 0; k < R; ++k) {
        size_t size = N >> k;
        if (plx[k] != L - k || plnx[k] != L - k) return -(int)(200 + k);
        compute_root_from_path(lx[k], tp % size, lpx[k], plx[k], root, hash);
        if (memcmp(root, roots[k], 32)) return -(int)(300 + k);
        compute_root_from_path(lnx[k], (tp + size / 2) % size, lpnx[k], plnx[k], root, hash);
        if (memcmp(root, roots[k], 32)) return -(int)(400 + k);
    }
    }
    if (left != 0) return -8;
    return 0;
}

// the reference verifier reads the challenges out of the proof
// (proof.rs:22-37) and never checks Proof.state (proof.rs:6); the author flags this as unfinished
// (readme.md:1).  This replays the Fiat-Shamir channel over the proof bytes in the prover's commit
// order (prover.rs:85, :163-165, :180, :200, :224, :254, :263, :274-277, :288), checks that every
// challenge equals the one the transcript yields at that point and that the final state matches.
// Returns 0, or -(1000 + k) for the k-th challenge / -1999 for the state.
inline int verify_transcript(const uint8_t* data, size_t len, const uint8_t state[32], uint32_t log_n, uint32_t log_b, uint32_t q = 1) {
    if (log_n < 2 || log_b < 1 || log_n + log_b > 30) return -1;
    const size_t R = log_n, L = log_n + log_b;
    if (q < 1 || q > 64 || len != proof_data_len(log_n, log_b, q)) return -1;
    Channel ch;
    const uint8_t* p = data;
    int k = 0;
    auto commit = [&](size_t n) { ch.commit_bytes(p, n); p += n; };
    auto challenge = [&]() -> bool {
        uint32_t expect = ((uint32_t)ch.state[0] << 24) | ((uint32_t)ch.state[1] << 16) | ((uint32_t)ch.state[2] << 8) | ch.state[3];
        uint32_t got = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
        ++k;
        if (got != expect) return false;
        commit(4);
        return true;
    };
    commit(32);                                             // f_eval root
    for (int i = 0; i < 3; ++i) if (!challenge()) return -(1000 + k);
    commit(32);                                             // cp root
    for (size_t r = 0; r < R; ++r) {
        if (!challenge()) return -(1000 + k);               // beta
        commit(32);                                         // layer root
    }
    commit(4);                                              // free term
    for (uint32_t j = 0; j < q; ++j) if (!challenge()) return -(1000 + k);   // queries
    for (uint32_t j = 0; j < q; ++j) {
        for (int i = 0; i < 4; ++i) commit(4 + 8 + 32 * L);
        for (size_t i = 0; i < R; ++i) commit(8 + 2 * (8 + 32 * (L - i)));
    }
    if (memcmp(ch.state, state, 32)) return -1999;
    return 0;
}

}  // namespace zk
