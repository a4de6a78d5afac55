"""Pins the CPU oracle to every known answer the reference holds:
its five unit tests and the assert_eq! checkpoints of prover.rs, then to the committed
golden fixtures.  CPU only."""
import hashlib
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), "golden")
KA = json.load(open(os.path.join(GOLD, "reference_known_answers.json")))
CANON = json.load(open(os.path.join(GOLD, "stark101_canonical.json")))
P = 3221225473


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u4").tobytes()).hexdigest()


# ---- the reference's unit tests ----------------------------------------------------
def test_generator_test(orc):
    """field.rs:213-226: Gf<4391>::generator() has order 4390 and hits every non-zero element once."""
    p = KA["generator_test"]["p"]
    g = orc.generator(p)
    assert orc.order(g, p) == KA["generator_test"]["order"]
    seen = set()
    x = 1
    for _ in range(p - 1):
        assert x not in seen and x != 0
        seen.add(x)
        x = orc.mul(x, g, p)
    assert len(seen) == p - 1


def test_fri_test(orc):
    """polynomial.rs:402-426 over Gf<101>."""
    t = KA["fri_test"]
    p = t["p"]
    coef = t["coef_high_first"][::-1]                 # Polynomial::from reverses (polynomial.rs:38-40)
    folded = orc.fri_coef_fold(coef, t["beta"], p)
    assert list(folded) == t["folded_high_first"][::-1]
    x = t["x"]
    nx = orc.neg(x, p)
    px, pnx = orc.poly_solve_naive(coef, x, p), orc.poly_solve_naive(coef, nx, p)
    g_xx = orc.mul(orc.add(px, pnx, p), orc.inv(2, p), p)
    h_xx = orc.mul(orc.sub(px, pnx, p), orc.inv(orc.mul(x, 2, p), p), p)
    lhs = orc.add(g_xx, orc.mul(t["beta"], h_xx, p), p)
    assert lhs == orc.poly_solve_naive(folded, orc.mul(x, x, p), p)


def test_lagrange_test(orc):
    """polynomial.rs:428-454 over Gf<7>: the interpolant does not depend on which 4 points of the cubic are used."""
    p = 7
    pts = {0: 2, 1: 4, 2: 3, 3: 1}
    poly = orc.lagrange_naive(list(pts), list(pts.values()), p)
    for x in (4, 5, 6):
        pts[x] = orc.poly_solve_naive(poly, x, p)
    for xs in ([0, 3, 5, 6], [1, 6, 3, 2], [3, 2, 1, 0], [6, 5, 4, 3]):
        assert list(orc.lagrange_naive(xs, [pts[x] for x in xs], p)) == list(poly)


def test_div_test(orc):
    """polynomial.rs:456-490 (i32 long division); coefficients low degree first here."""
    q, r = orc.poly_div_i32([-10, -3, 1], [2, 1])
    assert list(q) == [-5, 1] and len(r) == 0
    q, r = orc.poly_div_i32([-1, -5, 2], [-3, 1])
    assert list(q) == [1, 2] and list(r) == [2]
    q, r = orc.poly_div_i32([-9, 6, 0, 0, 2, 0, 1], [3, 0, 0, 1])
    assert list(q) == [-3, 2, 0, 1] and len(r) == 0


def test_merkle_test(orc):
    """merkle.rs:112-182."""
    t = KA["merkle_test"]
    nodes = orc.merkle_build(t["leaves"])
    assert [bytes(n).hex() for n in nodes] == t["nodes"]
    for leaf, want in enumerate(t["traces"]):
        assert [bytes(h).hex() for h in orc.merkle_trace(nodes, leaf)] == [t["nodes"][i] for i in want]
    assert orc.compute_root_from_path(1, 0, orc.merkle_trace(nodes, 0)).hex() == t["nodes"][0]


def test_sha256_against_hashlib(orc):
    for n in (0, 1, 4, 55, 56, 63, 64, 65, 119, 120, 1000):
        msg = bytes((i * 37 + n) & 0xFF for i in range(n))
        assert orc.sha256(msg) == hashlib.sha256(msg).digest()


# ---- prover.rs checkpoints ------------------------------------------------------------
def test_prover_checkpoints(orc):
    k = KA["prover"]
    a = orc.trace_fibsq(1023)
    assert a[1022] == k["trace_1022"]                                   # prover.rs:42
    assert orc.generator() == k["generator"]                            # prover.rs:44-45
    g, h = orc.pow_(5, k["g_exponent"]), orc.pow_(5, k["h_exponent"])   # prover.rs:48-49
    assert orc.order(g) == k["g_order"] and orc.order(h) == k["h_order"]  # prover.rs:52-53
    assert g == orc.gen_of_order_log(10) and h == orc.gen_of_order_log(13)
    f = orc.lde(a, 10, 3)
    assert list(f[:3]) == k["f_eval_head"] and list(f[-3:]) == k["f_eval_tail"]   # prover.rs:73-78


def test_constraint_quotients(orc):
    """prover.rs:101-159: exact divisions, degrees and the three evaluation checkpoints."""
    k = KA["prover"]
    n = 1024
    a = orc.trace_fibsq(n - 1)
    g = orc.gen_of_order_log(10)
    gp = [orc.pow_(g, i) for i in range(n)]
    # the interpolant: iNTT with the virtual last point == Lagrange through n-1 points
    y = np.concatenate([a, [orc.virtual_point(a, 10)]]).astype(np.uint32)
    f = orc.intt(y, g)
    assert f[n - 1] == 0 and f[n - 2] != 0
    f = f[:n - 1]

    def psub(u, v):
        m = max(len(u), len(v))
        w = [orc.sub(int(u[i]) if i < len(u) else 0, int(v[i]) if i < len(v) else 0) for i in range(m)]
        while w and w[-1] == 0:
            w.pop()
        return w
    c0, r0 = orc.poly_div(psub(f, [int(a[0])]), [orc.neg(gp[0]), 1])
    c1, r1 = orc.poly_div(psub(f, [int(a[n - 2])]), [orc.neg(gp[n - 2]), 1])
    assert len(r0) == 0 and len(r1) == 0                                # prover.rs:148-149
    assert len(c0) - 1 == k["c0_degree"] and len(c1) - 1 == k["c1_degree"]
    assert orc.poly_solve_naive(c0, 2718) == k["c0_at_2718"]            # prover.rs:157
    assert orc.poly_solve_naive(c1, 5772) == k["c1_at_5772"]            # prover.rs:158
    # c2(31415) through the pointwise formula of proof.rs:63-77 (the full division runs in test_naive_mode)
    x = 31415
    fx = orc.poly_solve_naive(f, x)
    fgx = orc.poly_solve_naive(f, orc.mul(g, x))
    fggx = orc.poly_solve_naive(f, orc.mul(orc.mul(g, g), x))
    num = orc.sub(orc.sub(fggx, orc.mul(fgx, fgx)), orc.mul(fx, fx))
    den = orc.mul(orc.sub(orc.pow_(x, n), 1),
                  orc.inv(orc.mul(orc.mul(orc.sub(x, gp[n - 3]), orc.sub(x, gp[n - 2])), orc.sub(x, gp[n - 1]))))
    assert orc.mul(num, orc.inv(den)) == k["c2_at_31415"]               # prover.rs:159


def test_naive_mode_equals_ntt_mode_at_reference_size(orc):
    """The literal polynomial.rs path (Lagrange, per-point solve, long division, coefficient fold)
    and the NTT restatement give the same evaluations, roots and proof bytes (prover.rs:169, :228-251)."""
    a = orc.prove(10, 3, mode=orc.MODE_NAIVE)
    b = orc.prove(10, 3, mode=orc.MODE_NTT)
    assert a.rc == 0 and b.rc == 0
    assert a.cp_degree == KA["prover"]["cp_degree"]
    assert np.array_equal(a.f_eval, b.f_eval)
    for la, lb in zip(a.cp_layers, b.cp_layers):
        assert np.array_equal(la, lb)
    assert [len(l) for l in a.cp_layers] == [8192 >> i for i in range(11)]   # prover.rs:241-251
    assert a.proof == b.proof and a.state == b.state
    assert orc.verify(a.proof, 10, 3, 2338775057) == 0


# ---- golden fixtures ---------------------------------------------------------------------
def test_canonical_golden(orc):
    r = orc.prove(10, 3)
    pin, der = CANON["pinned"], CANON["derived"]
    assert int(r.trace[1022]) == pin["trace_last"]
    assert orc.virtual_point(r.trace, 10) == pin["virtual_point"]
    assert sha(r.f_eval) == pin["f_eval_sha256"]
    assert bytes(r.roots[0]).hex() == pin["f_eval_root"]
    assert r.alpha_raw == der["alpha_raw"] and r.beta_raw == der["beta_raw"]
    assert [bytes(x).hex() for x in r.roots] == der["roots"]
    assert [sha(l) for l in r.cp_layers] == der["layer_sha256"]
    assert r.free_term == der["free_term"] and r.query_raw == der["query_raw"]
    assert r.proof.hex() == der["proof_hex"] and r.state.hex() == der["final_state"]
    assert len(r.proof) == 7836 and orc.proof_size(len(r.proof)) == 7884      # proof.rs:151-154
    assert orc.proof_data_len(10, 3) == 7836


def test_golden_matches_survey_appendix_c():
    """SURVEY.md Appendix C was derived by an independent restatement; the fixtures agree with it."""
    pin, der = CANON["pinned"], CANON["derived"]
    assert pin["virtual_point"] == 1822662890
    assert pin["f_eval_sha256"] == "018ef71acb9e15864301375a04955c1d401bff37d3ad8869293e8afee99cb81d"
    assert pin["f_eval_root"] == "e7090678303730d51aee399664256de5f6476ec86fb4d45fbf0556535fb09f48"
    assert der["alpha_raw"] == [361545003, 3235878091, 2708123352]
    assert der["roots"][1] == "3607a328263e286599ab2e932debf372d39c88b0fdb349d58846e187ab7fb55d"
    assert der["beta_raw"] == [4195595581, 3610452991, 724415084, 3295998851, 738561939, 3410211472,
                               11579057, 583424291, 2291229637, 890278089]
    assert der["free_term"] == 1478590336 and der["query_index"] == 7267
    assert der["opened"] == [3140394059, 3145853881, 3178419263, 1250535774]
    assert der["proof_sha256"] == "b956f69349dfb74d2facd9f886efa8b983fb61f17bd95cab2e3449fc57b4bb2e"
    assert der["final_state"] == "d7eec91544f72a592145e7d505a2f274de740e0319ede8c983fd84c7736f6712"


def test_other_sizes_golden(orc):
    for row in json.load(open(os.path.join(GOLD, "prover_sizes.json"))):
        if row["log_n"] > 12:
            continue            # the larger rows are checked on the GPU side
        r = orc.prove(row["log_n"], row["log_blowup"], 1, row["a1"])
        assert r.rc == 0
        assert hashlib.sha256(r.proof).hexdigest() == row["proof_sha256"]
        assert orc.verify(r.proof, row["log_n"], row["log_blowup"], row["public_last"]) == 0


def test_verifier_rejects_tampering(orc):
    r = orc.prove(6, 2)
    assert orc.verify(r.proof, 6, 2, r.public_last) == 0
    for pos in (0, 40, 100, len(r.proof) // 2, len(r.proof) - 1):
        bad = bytearray(r.proof)
        bad[pos] ^= 1
        assert orc.verify(bytes(bad), 6, 2, r.public_last) != 0, pos
    assert orc.verify(r.proof, 6, 2, (r.public_last + 1) % P) != 0
    assert orc.verify(r.proof[:-1], 6, 2, r.public_last) != 0


def test_degenerate_n8_is_refused(orc):
    assert orc.prove(3, 3).rc == -103
