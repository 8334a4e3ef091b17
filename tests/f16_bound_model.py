"""Test infrastructure (never imported by the product): a numpy restatement of the DEFAULT filter's arithmetic and of the
rounding bound the exact re-rank trusts, plus the constructed operand sets that attack it.

Restated, line for line:
  * ``gq_prep_kernel<.., FK = 2>`` (csrc/gq_prep.h): A = beta/2 - 1/(2 sd^2), B = mu / sd^2 in fp64, rounded once to fp32; the
    per-row power of two 2^-e_r that puts the largest |coefficient| into [2^13, 2^14) (``mixed_row_scale``); fp16 of the
    normalised [A | B]; fp16 of the code features [fp32(n n) | n]; the four sums S0..S3 (``rowsum``) and the seven sums of the
    data-dependent bound (``rowaux``: M_well, P, Q, Rb, |B|^2, max|A|, max(|A|, |B|), each rounded UP to fp32 by ``f32_up``),
    including the well / non-well classification ``A < 0 and |B| <= 12 |A|`` (vertex |mu'| <= 6).
  * ``gq_filter_bf16_kernel<.., FK = 2>`` (csrc/gq_filter_bf16.h): f~(r, j) = 2^e_r sum_slots fp16(a^) fp16(s) -- products of
    two fp16 values are exact in fp32; the accumulation is done here in fp64 (exact to 2^-53: the MFMA's own fp32
    accumulation is what the 4 u per step of ``kF16EfCoeff`` is charged for, csrc/gqhip.hip).
  * ``row_bound`` / ``f16_bound`` and the margin of ``rerank_block`` (csrc/gq_rerank.h:207-257, :373-382).

The reference score these protect: pit/quantization/gaussian.py:142-150 (VQ: vq.py:58-73).
"""
import numpy as np

U = 2.0 ** -24
HALF_LOG_2PI = float(np.float32(0.91893853320467274178))
K_F16 = 16700.0               # csrc/gqhip.hip: kF16EfCoeff
K_F16_REPR = 16384.0 + 4 + 3  # its representation share: two fp16 roundings + the fp32 roundings of A / B / n^2
N1_LIMIT = 255.0              # csrc/gqhip.hip: kF16N1Limit


def f32(x):
    return np.asarray(x, dtype=np.float64).astype(np.float32)


def f16_of_f32(x32):
    """fp32 -> fp16 (RNE, subnormals kept), returned as float64."""
    with np.errstate(over="ignore"):
        return np.asarray(x32, dtype=np.float32).astype(np.float16).astype(np.float64)


def f32_up(v):
    """csrc/gq_prep.h:f32_up -- fp64 -> fp32, never below the argument."""
    return (np.asarray(v, dtype=np.float64) * 1.0000002384185791).astype(np.float32)


def coefficients(mu32, sd32, beta, mode="gq"):
    """gq_prep_kernel: fp32 filter coefficients [A | B] and the four bound sums (fp64)."""
    mu = np.asarray(mu32, np.float32).astype(np.float64)
    rows, dim = mu.shape
    if mode == "vq":
        A = np.full((rows, dim), -1.0, np.float32)
        B = (np.float32(2.0) * np.asarray(mu32, np.float32)).astype(np.float32)
        rs = np.zeros((rows, 4))
        rs[:, 1] = np.abs(mu).sum(1)
        return A, B, rs
    sd = np.asarray(sd32, np.float32).astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = 1.0 / (sd * sd)
        A = (0.5 * float(np.float32(beta)) - 0.5 * inv).astype(np.float32)
        B = (mu * inv).astype(np.float32)
        lsd = np.log(sd).astype(np.float32).astype(np.float64)
        inv = np.where(sd > 0.0, inv, np.nan)
    rs = np.stack([inv.sum(1), (np.abs(mu) * inv).sum(1), (mu * mu * inv).sum(1), np.abs(lsd).sum(1)], 1)
    return A, B, rs


def sums_from_coefficients(A32, B32, beta):
    """The four sums of rows given directly by their coefficients: the (mu, sd) they are the exact image of."""
    A, B = A32.astype(np.float64), B32.astype(np.float64)
    inv = float(beta) - 2.0 * A
    assert (inv > 0).all(), "A must be below beta / 2"
    mu = B / inv
    lsd = -0.5 * np.log(inv)
    return np.stack([inv.sum(1), (np.abs(mu) * inv).sum(1), (mu * mu * inv).sum(1), np.abs(lsd).sum(1)], 1)


def row_scale(coef32):
    """mixed_row_scale: s = 2^-e_r with max|c| s in [2^13, 2^14); NaN when there is no usable normalisation."""
    amax = np.abs(coef32.astype(np.float64)).max(1)
    bad = ~np.isfinite(amax) | ~(amax >= 7.8886090522101181e-31) | ~(amax <= 1.2676506002282294e30)
    _, ex = np.frexp(np.where(bad, 1.0, amax))
    return np.where(bad, np.nan, np.ldexp(1.0, 14 - ex))


def filter_f16(A32, B32, cb32):
    """The filter value f~(r, j) in true units, [rows, n] float64."""
    coef = np.concatenate([A32, B32], 1).astype(np.float32)
    sc = row_scale(coef)
    with np.errstate(over="ignore", under="ignore"):
        ch = f16_of_f32((coef * sc[:, None].astype(np.float32)).astype(np.float32))        # fp32 product, then fp16
        c32 = np.asarray(cb32, np.float32)
        feat = np.concatenate([(c32 * c32).astype(np.float32), c32], 1)                     # the square is an fp32 op
    sh = f16_of_f32(feat)
    return (ch @ sh.T) / sc[:, None]


def rowaux(A32, B32):
    """gq_prep_kernel: the sums of the data-dependent bound, each rounded up to fp32 (returned as float64)."""
    A, B = A32.astype(np.float64), B32.astype(np.float64)
    a, b = np.abs(A), np.abs(B)
    well = (A < 0.0) & (b <= 12.0 * a)
    with np.errstate(divide="ignore", invalid="ignore"):
        Mw = np.where(well, B * B / (4.0 * a), 0.0).sum(1)
    P = np.where(~well, np.maximum(A, 0.0), 0.0).sum(1)
    Q = np.where(~well, a, 0.0).sum(1)
    Rb = np.where(~well, b, 0.0).sum(1)
    B2 = (B * B).sum(1)
    Amax = a.max(1)
    cmax = np.maximum(a.max(1), b.max(1))
    out = np.stack([f32_up(Mw), f32_up(P), f32_up(Q), f32_up(Rb), f32_up(B2), f32_up(Amax), f32_up(cmax)], 1).astype(np.float64)
    return out, well


def row_bound(rs, N1, dim, beta, mode="gq"):
    """gq_rerank.h:row_bound -> T (filter side), E_r (the reference's own rounding noise)."""
    N2 = N1 * N1
    if mode == "vq":
        T = dim * N2 + 2.0 * rs[:, 1] * N1
        return T, 1e-12 * T
    b, c = abs(float(np.float32(beta))), HALF_LOG_2PI
    T = (0.5 * b * dim + 0.5 * rs[:, 0]) * N2 + rs[:, 1] * N1
    G = 0.5 * (N2 * rs[:, 0] + 2.0 * N1 * rs[:, 1] + rs[:, 2]) + rs[:, 3] + dim * (c + b * (0.5 * N2 + c))
    return T, (dim + 16.0) * U * G


def f16_bound(aux, T_old, Er, N1, R2, F, dim, k=K_F16):
    """gq_rerank.h:f16_bound, vectorised over rows.  Returns a dict with Ea, Eb, E_unif, E_abs, Cr, T_norm."""
    ku = k * U
    N2 = N1 * N1
    U_wc = aux[:, 1] * N2 + aux[:, 3] * N1
    T_wc = aux[:, 2] * N2 + aux[:, 3] * N1
    Cr = 8.0 * aux[:, 0] + 3.0 * U_wc + T_wc
    T_norm = aux[:, 5] * R2 + np.sqrt(aux[:, 4] * R2)
    NN = max(N2, N1)
    E_abs = 2.0 * dim * (2.0 ** -25 * NN + 2.0 ** -11) * aux[:, 6] * 2.0 ** -13
    Tu = np.minimum(T_old, T_norm)
    E_unif = ku * Tu + E_abs
    slack = np.maximum(Cr - 3.0 * F, 0.0)
    Ea = np.minimum((ku * slack + E_abs) / (1.0 - 3.0 * ku), E_unif)
    Eb = np.minimum(ku * (slack + 3.0 * Ea + 6.0 * Er) + E_abs, E_unif)
    return dict(Ea=Ea, Eb=Eb, E_unif=E_unif, E_abs=E_abs, Cr=Cr, T_norm=T_norm)


def codebook_norms(cb32):
    """max|cb| and the largest squared code norm as the first launch leaves them (fp32 fma chain, inflated by 64 u)."""
    c = np.asarray(cb32, np.float32).astype(np.float64)
    return float(np.abs(c).max()), float((c * c).sum(1).max() * (1.0 + 64.0 * U))


def analyse(A32, B32, rs, cb32, beta, mode="gq", k=K_F16):
    """Everything the assertions need for a set of rows against a codebook (all float64, [rows] or [rows, n])."""
    dim = A32.shape[1]
    coef = np.concatenate([A32, B32], 1).astype(np.float64)
    c = np.asarray(cb32, np.float32).astype(np.float64)
    feat = np.concatenate([c * c, c], 1)
    f_true = coef @ feat.T
    T_j = np.abs(coef) @ np.abs(feat).T
    ft = filter_f16(A32, B32, cb32)
    N1, R2 = codebook_norms(cb32)
    T_old, Er = row_bound(rs, N1, dim, beta, mode)
    aux, well = rowaux(A32, B32)
    F = ft.max(1)
    b = f16_bound(aux, T_old, Er, N1, R2, F, dim, k)
    return dict(f=f_true, ft=ft, T_j=T_j, N1=N1, R2=R2, T_old=T_old, Er=Er, aux=aux, well=well, F=F, **b)


# ------------------------------------------------------------------ constructed operand sets

def _tie_values(rng, shape, klo, khi):
    """Values 2^k (1 + (2 m + 1) 2^-11): exactly half way between two fp16 neighbours (the largest fp16 rounding error,
    direction by the parity of m), signs random."""
    k = rng.integers(klo, khi, shape)
    m = rng.integers(0, 1024, shape)
    return np.ldexp(1.0 + (2.0 * m + 1.0) * 2.0 ** -11, k) * rng.choice([-1.0, 1.0], shape)


def _sqrt_tie_values(rng, shape, klo, khi):
    """n whose fp32 SQUARE sits just below such a tie (the square is what the filter converts)."""
    t = np.abs(_tie_values(rng, shape, 2 * klo, 2 * khi)) * (1.0 - 2.0 ** -21)
    return np.sqrt(t) * rng.choice([-1.0, 1.0], shape)


def codebooks(rng, n, dim, scale):
    """Code sets with max|cb| == scale (<= 255): Gaussian-like, fp16 ties in the values, fp16 ties in the squares, tiny codes
    (fp16-subnormal values and squares), a zero code, and codes at +-scale in every coordinate."""
    base = rng.standard_normal((n, dim)) * (scale / 4.6)
    q = n // 8
    hi = int(np.floor(np.log2(scale))) if scale >= 2.0 ** -20 else -20
    base[q:2 * q] = _tie_values(rng, (q, dim), hi - 6, hi)
    base[2 * q:3 * q] = _sqrt_tie_values(rng, (q, dim), hi - 6, hi)
    base[3 * q:3 * q + q // 2] = rng.standard_normal((q // 2, dim)) * 2.0 ** -8 * min(scale, 1.0)      # squares below 2^-14
    base[3 * q + q // 2:4 * q] = rng.standard_normal((q - q // 2, dim)) * 2.0 ** -16 * min(scale, 1.0)  # values below 2^-14
    base[4 * q] = 0.0
    base = np.clip(base, -scale, scale)
    base[4 * q + 1] = scale
    base[4 * q + 2] = -scale
    base[4 * q + 3] = scale * rng.choice([-1.0, 1.0], dim)
    return base.astype(np.float32)


def coefficient_sets(rng, rows, dim, beta, cb32):
    """name -> (A32, B32): rows given directly by their fp32 coefficients (A < beta / 2)."""
    n = cb32.shape[0]
    half = 0.5 * beta
    out = {}
    # half-ulp ties in every coefficient (after the power-of-two normalisation a tie stays a tie), 8 binades inside a row
    A = -np.abs(_tie_values(rng, (rows, dim), -3, 5))
    B = _tie_values(rng, (rows, dim), -3, 5)
    out["fp16 ties, wells and non-wells mixed"] = (A, B)
    # the class boundary |B| = 12 |A| (vertex |mu'| = 6), from both sides and exactly on it
    A = -np.abs(rng.standard_normal((rows, dim))) * np.ldexp(1.0, rng.integers(-6, 6, (rows, 1))) - 2.0 ** -12
    eps = rng.choice([0.0, 2.0 ** -23, -2.0 ** -23, 2.0 ** -16, -2.0 ** -16, 2.0 ** -9, -2.0 ** -9], (rows, dim))
    out["class boundary |B| = 12 |A| (1 + eps)"] = (A, 12.0 * np.abs(A) * (1.0 + eps) * rng.choice([-1.0, 1.0], (rows, dim)))
    # A -> 0 from both sides (sd -> 1 at beta = 1), A == 0
    if beta > 0:
        A = rng.choice([0.0, 2.0 ** -40, -2.0 ** -40, 2.0 ** -24, -2.0 ** -24, 1e-3, -1e-3, 0.25 * beta], (rows, dim))
        out["A -> 0 from both sides"] = (A, rng.standard_normal((rows, dim)))
    # slack -> 0: every coordinate a well with its vertex (almost) on a code, so the best code scores F ~ M_well and
    # Cr - 3 F -> 5 M_well; with B == 0 and the zero code, F -> 0 and Cr == 0
    A = -np.abs(rng.standard_normal((rows, dim))) * 4.0 - 0.01
    B = np.zeros((rows, dim))
    B[rows // 2:] = rng.standard_normal((rows - rows // 2, dim)) * 2.0 ** -20
    out["B = 0: best code is the zero code, Cr - 3 F -> 0"] = (A, B)
    pick = cb32[rng.integers(0, n, rows)].astype(np.float64)
    A = -np.abs(rng.standard_normal((rows, dim))) * 8.0 - 0.5
    out["vertex on a code: F ~ M_well"] = (A, 2.0 * np.abs(A) * pick)
    # coefficients spread over 40 binades in one row: everything 2^27 below the row's largest is fp16-subnormal after
    # the normalisation (absolute error, E_abs)
    A = -np.ldexp(np.abs(rng.standard_normal((rows, dim))) + 0.5, rng.integers(-30, 10, (rows, dim)))
    B = np.ldexp(rng.standard_normal((rows, dim)), rng.integers(-30, 10, (rows, dim)))
    out["40 binades inside a row (fp16-subnormal coefficients)"] = (A, B)
    # one dominant coordinate, the rest just at / below the subnormal threshold 2^-27 of it
    A = -np.ldexp(1.0 + rng.random((rows, dim)), -27 + rng.integers(-2, 3, (rows, dim)))
    A[:, 0] = -(1.0 + rng.random(rows))
    B = np.ldexp(rng.standard_normal((rows, dim)), -27)
    B[:, 0] = rng.standard_normal(rows)
    out["one dominant coordinate, the others at the fp16-subnormal threshold"] = (A, B)
    res = {}
    for name, (A, B) in out.items():
        A = np.minimum(A, half - 2.0 ** -30 if beta > 0 else -2.0 ** -60)
        res[name] = (A.astype(np.float32), B.astype(np.float32))
    return res


def rows_from_coefficients(A32, B32, beta):
    """(mu, sd) in fp32 whose gq_prep coefficients are (as close as fp32 rows allow to) the given ones."""
    A, B = A32.astype(np.float64), B32.astype(np.float64)
    inv = float(beta) - 2.0 * A
    sd = (1.0 / np.sqrt(inv)).astype(np.float32)
    s = sd.astype(np.float64)
    mu = (B * s * s).astype(np.float32)
    return mu, sd
