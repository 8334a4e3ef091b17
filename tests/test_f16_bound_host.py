"""CPU: the default (fp16 main-product) filter and the data-dependent rounding bound the exact re-rank trusts, restated in numpy
(tests/f16_bound_model.py) and attacked with constructed operands.  What must hold for the indices to be the reference's
(pit/quantization/gaussian.py:142-150) -- DESIGN.md section 3 maps each inequality to the case below that attacks it:

  (I1) |f~(j) - f(j)| <= k u T_j + E_abs                              for EVERY code j           (representation error)
  (I2) T_j <= T_old,  T_j <= T_norm,  T_j <= Cr - 3 f(j)             for EVERY code j           (the three bounds on T_j)
  (I3) f~(j) >= F - (Ea + Eb + 2 E_r)   for every j with f(j) >= f(j^) - 2 E_r                   (what the margin assumes of j*)
"""
import numpy as np
import pytest

import f16_bound_model as M


def _check(name, A, B, rs, cb, beta, mode, dim):
    r = M.analyse(A, B, rs, cb, beta, mode)
    f, ft, Tj = r["f"], r["ft"], r["T_j"]
    # (I1) with the representation share of the coefficient only (the accumulation here is exact)
    E1 = M.K_F16_REPR * M.U * Tj + r["E_abs"][:, None]
    err = np.abs(ft - f)
    tight1 = float((err / np.maximum(E1, 1e-300)).max())
    assert (err <= E1).all(), (name, "I1", tight1)
    # (I2): 1e-6 relative slack for the fp32 roundings of the sums themselves (f32_up rounds them up; T_old is fp64 of the true mu, sd)
    big = np.maximum(np.abs(r["Cr"]), np.abs(f).max(1))[:, None]
    assert (Tj <= r["T_old"][:, None] * (1 + 1e-6)).all(), (name, "I2 T_old")
    assert (Tj <= r["T_norm"][:, None] * (1 + 1e-6)).all(), (name, "I2 T_norm")
    assert (Tj <= r["Cr"][:, None] - 3.0 * f + 1e-6 * big).all(), (name, "I2 Cr - 3 f")
    # (I3)
    jh = ft.argmax(1)
    rows = np.arange(f.shape[0])
    contenders = f >= (f[rows, jh] - 2.0 * r["Er"])[:, None]
    need = (r["F"] - (r["Ea"] + r["Eb"] + 2.0 * r["Er"]))[:, None]
    short = np.where(contenders, need - ft, -np.inf)          # > 0: a possible j* below the window
    denom = (r["Ea"] + r["Eb"] + 2.0 * r["Er"])[:, None]
    with np.errstate(invalid="ignore", divide="ignore"):
        used = np.where(contenders, (r["F"][:, None] - ft) / denom, 0.0)
    tight3 = float(np.nanmax(used))
    assert (short <= 0).all(), (name, "I3", tight3)
    return tight1, tight3, int(r["well"].sum()), int((~r["well"]).sum()), float(np.median(r["Cr"] - 3 * r["F"]))


@pytest.mark.parametrize("dim", [8, 16, 32])
@pytest.mark.parametrize("scale", [2.0 ** -8, 1.0, 16.0, 254.9])
@pytest.mark.parametrize("beta", [1.0, 0.0, 2.0])
def test_f16_filter_bound_holds_on_constructed_gaussian_rows(dim, scale, beta):
    rng = np.random.default_rng(1000 * dim + int(scale * 7) + int(beta * 13))
    n, rows = 2048, 96
    cb = M.codebooks(rng, n, dim, scale)
    assert np.abs(cb).max() == np.float32(scale) and scale <= M.N1_LIMIT
    worst1 = worst3 = 0.0
    for name, (A, B) in M.coefficient_sets(rng, rows, dim, beta, cb).items():
        # (a) the rows exactly as constructed: coefficients given, sums from the (mu, sd) they are the image of
        t1, t3, nw, nn, slack = _check(name, A, B, M.sums_from_coefficients(A, B, beta), cb, beta, "gq", dim)
        # (b) through the library's own path: fp32 (mu, sd) rows -> gq_prep's coefficients and sums
        mu, sd = M.rows_from_coefficients(A, B, beta)
        A2, B2, rs2 = M.coefficients(mu, sd, beta)
        ok = np.isfinite(A2).all(1) & np.isfinite(B2).all(1) & np.isfinite(rs2).all(1)
        u1, u3, _, _, _ = _check(name + " via (mu, sd)", A2[ok], B2[ok], rs2[ok], cb, beta, "gq", dim)
        worst1, worst3 = max(worst1, t1, u1), max(worst3, t3, u3)
        print(f"dim {dim:2d} max|cb| {scale:8.4f} beta {beta}: {name:70s} I1 {max(t1, u1):.3f}  I3 {max(t3, u3):.3f} of the bound; "
              f"wells {nw} / others {nn}; median Cr - 3F {slack:.3g}")
    assert worst1 <= 1.0 and worst3 <= 1.0


@pytest.mark.parametrize("dim", [8, 16, 32])
@pytest.mark.parametrize("scale", [2.0 ** -8, 1.0, 16.0, 254.9])
def test_f16_filter_bound_holds_on_constructed_vq_rows(dim, scale):
    """VQ (vq.py:58-73) through the same filter: A = -1, B = 2 z."""
    rng = np.random.default_rng(77 * dim + int(scale * 5))
    n, rows = 2048, 128
    cb = M.codebooks(rng, n, dim, scale)
    zsets = {
        "z on codes": cb[rng.integers(0, n, rows)].astype(np.float64),
        "z = fp16 ties": M._tie_values(rng, (rows, dim), -8, 4) * min(scale, 1.0),
        "|z| = 6 (class boundary |B| = 12 |A|)": 6.0 * (1 + rng.choice([0.0, 2.0 ** -23, -2.0 ** -23, 2.0 ** -10, -2.0 ** -10], (rows, dim))) * rng.choice([-1.0, 1.0], (rows, dim)),
        "z = 0 / tiny": rng.standard_normal((rows, dim)) * rng.choice([0.0, 2.0 ** -30, 2.0 ** -16], (rows, 1)),
        "z far outside the codebook": rng.standard_normal((rows, dim)) * 1000.0,
    }
    for name, z in zsets.items():
        A, B, rs = M.coefficients(z.astype(np.float32), None, 0.0, "vq")
        t1, t3, nw, nn, slack = _check(name, A, B, rs, cb, 0.0, "vq", dim)
        print(f"VQ dim {dim:2d} max|cb| {scale:8.4f}: {name:45s} I1 {t1:.3f}  I3 {t3:.3f} of the bound; wells {nw} / others {nn}")
        assert t1 <= 1.0 and t3 <= 1.0


def test_f16_bound_restatement_matches_the_study_on_trained_like_rows():
    """Sanity of the restatement itself on ordinary rows (the synthetic trained-VAE-like recipe of SURVEY.md section 8d): the bound
    holds with room, and the data-dependent bound is the one that wins (that is its point)."""
    rng = np.random.default_rng(5)
    rows, dim, n = 256, 16, 8192
    mu = (0.9 * rng.standard_normal((rows, dim))).astype(np.float32)
    sd = np.exp(0.5 * (-1.5 + 0.3 * rng.standard_normal((rows, dim)))).astype(np.float32)
    cb = np.clip(rng.standard_normal((n, dim)), -4.6, 4.6).astype(np.float32)
    A, B, rs = M.coefficients(mu, sd, 1.0)
    t1, t3, nw, nn, slack = _check("trained-like", A, B, rs, cb, 1.0, "gq", dim)
    r = M.analyse(A, B, rs, cb, 1.0)
    assert t1 < 0.6 and t3 < 0.6
    worst_case = 2.0 * (M.K_F16 * M.U * r["T_old"] + r["E_abs"])          # what Ea + Eb would be under the worst-case T alone
    assert np.median(r["Ea"] + r["Eb"]) < 0.25 * np.median(worst_case)


def _rec_gap(m1, mk):
    """numpy restatement of csrc/gq_common.h:rec_gap -- fp32 difference, x (1 - 2^-22) in fp32, fp16 towards zero; -inf -> +inf."""
    m1, mk = np.asarray(m1, np.float32), np.asarray(mk, np.float32)
    with np.errstate(invalid="ignore", over="ignore"):
        g = ((m1 - mk).astype(np.float32) * np.float32(0.99999976158142090)).astype(np.float32)
        h = g.astype(np.float16)                                          # round to nearest ...
        up = h.astype(np.float32) > g                                     # ... then one step back where that went up
        h = np.where(up, np.nextafter(h, np.float16(0.0)), h).astype(np.float16)
    h = np.where(np.isfinite(g) & ~np.isfinite(h), np.float16(65504.0), h)   # finite never rounds to inf towards zero
    return np.where(mk > -np.inf, h, np.float16(np.inf))


def test_record_gap_encoding_never_reads_below_the_filters_value():
    """The candidate records carry m2..m4 as fp16 gaps below m1 (16-byte records).  Whatever the pair, the value the re-rank
    reconstructs, fp64(m1) - fp64(gap), is >= mk (a group is never missed) and above it by at most the gap's fp16 resolution."""
    rng = np.random.default_rng(11)
    n = 200000
    m1 = (rng.standard_normal(n) * 10.0 ** rng.integers(-3, 6, n)).astype(np.float32)
    gaps = [rng.random(n) * 10.0 ** rng.integers(-9, 6, n),                    # anything from far below an ulp to overflow
            M._tie_values(rng, (n,), -20, 14).astype(np.float64) * rng.choice([1.0, 1 + 2.0 ** -10, 1 - 2.0 ** -11], n),   # fp16 ties / exact values
            np.abs(m1) * 2.0 ** -24 * rng.integers(0, 4, n),                   # a few fp32 ulps of m1
            np.zeros(n)]
    for gtrue in gaps:
        mk = (m1.astype(np.float64) - np.abs(gtrue)).astype(np.float32)
        mk = np.minimum(mk, m1)
        h = _rec_gap(m1, mk)
        rec = m1.astype(np.float64) - h.astype(np.float64)
        real_gap = m1.astype(np.float64) - mk.astype(np.float64)
        assert (rec >= mk.astype(np.float64)).all()
        fits = real_gap <= 65504.0                                             # larger gaps read as 65504 (still above mk)
        assert ((rec - mk)[fits] <= (real_gap * 2.0 ** -10 + 2.0 ** -24 + real_gap * 2.0 ** -21)[fits]).all()
        assert (h.astype(np.float64)[real_gap > 65600.0] == 65504.0).all()
    none = _rec_gap(m1, np.full(n, -np.inf, np.float32))
    assert np.isinf(none).all() and ((m1.astype(np.float64) - none.astype(np.float64)) == -np.inf).all()
