"""-m gpu: randomized / adversarial parity sweep of the fused arg-max against the oracle.
Stresses the exactness machinery (rounding margins, candidate records, in-block fp64 scan of undecided rows, partial
tiles, code splits) with heavy-tailed sigmas, large means, odd sizes and several betas."""
import os

import numpy as np
import pytest
import torch

from oracle import gq_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["auto", "bf16", "fp32", "mixed"])
def filter_kind(request):
    """Every test of this module runs with all four filter selections ("auto": the fp16 main-product filter + data-dependent bound;
    "mixed": round 2's fp16 + fp8 at dim 16, split-bf16 at the other
    MFMA dims; "bf16": split-bf16 everywhere; "fp32": the fp32 MFMA filter); same indices."""
    from pit_hip import _lib

    _lib.set_filter(request.param)
    yield request.param
    _lib.set_filter("auto")


def _case(seed):
    rng = np.random.default_rng(seed)
    dim = int(rng.choice([4, 8, 16, 32]))
    n = int(rng.choice([33, 100, 1000, 4096, 5000, 20000, 65536]))
    rows = int(rng.integers(1, 1500))
    beta = float(rng.choice([0.0, 0.5, 1.0, 2.0]))
    kind = int(rng.integers(0, 5))
    g = torch.Generator().manual_seed(seed)
    mu = torch.randn(rows, dim, generator=g)
    if kind == 0:      # trained-model-like
        mu, sd = 0.9 * mu, torch.exp(0.5 * (-1.5 + 0.3 * torch.randn(rows, dim, generator=g)))
    elif kind == 1:    # the reference smoke loop: |randn| sigmas (tiny values included)
        sd = torch.abs(torch.randn(rows, dim, generator=g)) + 1e-4
    elif kind == 2:    # log-uniform sigma over 5 decades, large means
        mu, sd = 3.0 * mu, torch.exp(torch.rand(rows, dim, generator=g) * 11.5 - 9.2)
    elif kind == 3:    # nearly flat posteriors (sigma >> 1): scores almost tie
        sd = 5.0 + 20.0 * torch.rand(rows, dim, generator=g)
    else:              # rows sitting exactly on codewords with small sigma
        sd = torch.full((rows, dim), 0.02)
    cb = O.codebook(n, dim, 42)
    if kind == 4:
        mu = torch.from_numpy(cb[rng.integers(0, n, rows)])
    return dim, n, rows, beta, kind, mu.contiguous(), sd.contiguous(), cb


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("GQ_STRESS_SEEDS", "24")))))
def test_randomized_parity(seed):
    from pit_hip import _lib

    dim, n, rows, beta, kind, mu, sd, cb = _case(seed)
    dev = torch.device("cuda:0")
    lsd = O.torch_log(sd.numpy())
    idx, zhat = _lib.gq_argmax(mu.to(dev), sd.to(dev), torch.from_numpy(cb).to(dev), beta,
                               logsd=torch.from_numpy(lsd).to(dev))
    idx = idx.cpu().numpy()
    budget = 40_000_000  # (row, code) pairs the oracle evaluates per case
    step = max(1, int(np.ceil(rows * n / budget)))
    sel = np.arange(0, rows, step)
    ref, _ = O.argmax_rows(mu.numpy()[sel], sd.numpy()[sel], cb, beta, logstd=lsd[sel])
    assert np.array_equal(idx[sel], ref), f"seed {seed}: dim {dim} n {n} rows {rows} beta {beta} kind {kind}"
    assert np.array_equal(zhat.cpu().numpy(), cb[idx])
