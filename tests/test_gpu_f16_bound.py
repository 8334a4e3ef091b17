"""-m gpu: the constructed operand sets of tests/f16_bound_model.py (fp16 rounding ties, the well / non-well class boundary,
A -> 0, slack -> 0, fp16-subnormal coefficients and code features, max|cb| from 2^-8 to 254.9) through the C ABI:
  * indices equal the oracle's (the reference's arg-max, pit/quantization/gaussian.py:142-150; VQ: vq.py:58-73) with
    every filter selection;
  * behind the default filter, every candidate record the filter kernel left is within the bound the re-rank charges
    for it -- evaluated by the numpy restatement of gq_rerank.h:f16_bound on the same operands."""
import numpy as np
import pytest
import torch

import f16_bound_model as M
from oracle import gq_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["auto", "bf16", "fp32", "mixed"])
def filter_kind(request):
    from pit_hip import _lib

    _lib.set_filter(request.param)
    yield request.param
    _lib.set_filter("auto")


def _records_within_bound(ws, rows, n, dim, A, B, rs, cb, beta, mode):
    """max over (set, row) of |m1 - max_group f| / max_group E(j), E(j) = k u T_j + E_abs (the charged coefficient).  The second
    and third values of a record travel as fp16 gaps below m1, rounded towards zero (csrc/gq_common.h:Rec): what the re-rank
    reconstructs must never be BELOW the filter's value -- checked here against the group's true maximum minus the same bound --
    and not above it by more than the gap's fp16 resolution."""
    from pit_hip import _lib

    pl = _lib.debug_plan(rows, n, dim)
    assert pl["bf16"] == 3, pl
    m, ids = _lib.debug_records(ws, rows, n, dim)
    m = m.cpu().numpy().astype(np.float64)
    ids = ids.cpu().numpy()
    r = M.analyse(A, B, rs, cb, beta, mode)
    E = M.K_F16 * M.U * r["T_j"] + r["E_abs"][:, None]
    gt = pl["gt"]
    q = np.arange(16 * gt)
    worst = 0.0
    ar = np.arange(rows)
    for s in range(pl["nsplit"]):
        for k in range(3):
            mk = m[s, :, k]
            ok_rec = np.isfinite(mk)
            gid = ids[s, :, k]
            tile = (gid >> 1)[:, None] * gt + (q >> 4)[None, :]
            code = tile * 32 + (q & 3)[None, :] + 8 * ((q & 15) >> 2)[None, :] + 4 * (gid & 1)[:, None]
            inside = code < n
            cc = np.minimum(code, n - 1)
            fg = np.where(inside, r["f"][ar[:, None], cc], -np.inf).max(1)
            Eg = np.where(inside, E[ar[:, None], cc], 0.0).max(1)
            sel = ok_rec & np.isfinite(fg)
            if k == 0:
                ratio = np.abs(mk - fg) / np.maximum(Eg, 1e-300)
                worst = max(worst, float(ratio[sel].max(initial=0.0)))
            else:
                gap = m[s, :, 0] - mk
                low = (fg - mk) / np.maximum(Eg, 1e-300)                   # <= 1: never below the filter's own value - bound
                assert float(low[sel].max(initial=0.0)) <= 1.0, (s, k, float(low[sel].max()))
                over = mk - fg - Eg - gap * 2.0 ** -9 - 2.0 ** -22 * np.abs(m[s, :, 0]) - 2.0 ** -24     # gap: fp16, 2^-24 floor
                assert float(over[sel].max(initial=-1.0)) <= 0.0, (s, k, float(over[sel].max()))
                assert (gap[ok_rec] >= 0).all()
    return worst


@pytest.mark.parametrize("dim,scale,beta", [(16, 1.0, 1.0), (16, 254.9, 1.0), (16, 2.0 ** -8, 1.0), (16, 16.0, 0.0), (8, 16.0, 1.0),
                                            (8, 254.9, 2.0), (32, 1.0, 1.0), (32, 16.0, 2.0), (32, 254.9, 0.0), (8, 2.0 ** -8, 0.0)])
def test_constructed_gaussian_rows_indices_and_records(dim, scale, beta, filter_kind):
    from pit_hip import _lib

    rng = np.random.default_rng(31 * dim + int(scale * 3) + int(beta * 7))
    n, rows = 2048, 256
    cb = M.codebooks(rng, n, dim, scale)
    dev = torch.device("cuda:0")
    cbd = torch.from_numpy(cb).to(dev)
    for name, (A0, B0) in M.coefficient_sets(rng, rows, dim, beta, cb).items():
        mu, sd = M.rows_from_coefficients(A0, B0, beta)
        A, B, rs = M.coefficients(mu, sd, beta)
        ok = np.isfinite(A).all(1) & np.isfinite(B).all(1) & np.isfinite(rs).all(1) & (sd > 0).all(1)
        mu, sd, A, B, rs = mu[ok], sd[ok], A[ok], B[ok], rs[ok]
        lsd = O.torch_log(sd)
        ws = _lib.Workspace()
        idx, zhat = _lib.gq_argmax(torch.from_numpy(mu).to(dev), torch.from_numpy(sd).to(dev), cbd, beta,
                                   logsd=torch.from_numpy(lsd).to(dev), ws=ws)
        torch.cuda.synchronize()
        ref, _ = O.argmax_rows(mu, sd, cb, beta, logstd=lsd)
        got = idx.cpu().numpy()
        assert np.array_equal(got, ref), (name, filter_kind, int((got != ref).sum()))
        assert np.array_equal(zhat.cpu().numpy(), cb[got])
        if filter_kind == "auto":
            w = _records_within_bound(ws, len(mu), n, dim, A, B, rs, cb, beta, "gq")
            fb, _ = _lib.debug_counters(ws)
            print(f"dim {dim} max|cb| {scale} beta {beta}: {name:70s} records within {w:.3f} of the charged bound; listed rows {fb}")
            assert w <= 1.0, (name, w)


@pytest.mark.parametrize("dim,scale", [(16, 1.0), (16, 254.9), (8, 16.0), (32, 2.0 ** -8), (32, 16.0)])
def test_constructed_vq_rows_indices_and_records(dim, scale, filter_kind):
    from pit_hip import _lib

    rng = np.random.default_rng(17 * dim + int(scale * 3))
    n, rows = 2048, 256
    cb = M.codebooks(rng, n, dim, scale)
    dev = torch.device("cuda:0")
    zsets = {
        "z on codes": cb[rng.integers(0, n, rows)].astype(np.float64),
        "z = fp16 ties": M._tie_values(rng, (rows, dim), -8, 4) * min(scale, 1.0),
        "|z| = 6": 6.0 * (1 + rng.choice([0.0, 2.0 ** -23, -2.0 ** -23, 2.0 ** -10, -2.0 ** -10], (rows, dim))) * rng.choice([-1.0, 1.0], (rows, dim)),
        "z = 0 / tiny": rng.standard_normal((rows, dim)) * rng.choice([0.0, 2.0 ** -30, 2.0 ** -16], (rows, 1)),
        "z far outside": rng.standard_normal((rows, dim)) * 1000.0,
    }
    for name, z in zsets.items():
        z32 = z.astype(np.float32)
        ws = _lib.Workspace()
        idx, zq = _lib.vq_argmin(torch.from_numpy(z32).to(dev), torch.from_numpy(cb).to(dev), ws=ws)
        torch.cuda.synchronize()
        ref = O.vq_argmin_rows(z32, cb)
        ref = ref[0] if isinstance(ref, tuple) else ref
        got = idx.cpu().numpy()
        assert np.array_equal(got, ref), (name, filter_kind, int((got != ref).sum()))
        if filter_kind == "auto":
            A, B, rs = M.coefficients(z32, None, 0.0, "vq")
            w = _records_within_bound(ws, rows, n, dim, A, B, rs, cb, 0.0, "vq")
            print(f"VQ dim {dim} max|cb| {scale}: {name:20s} records within {w:.3f} of the charged bound")
            assert w <= 1.0, (name, w)
