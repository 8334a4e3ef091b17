"""-m gpu: the module-level eval forwards that run as ONE library call (ABI 8) -- GaussianQuantRegularizer2.forward
(gq_quantize_z_gauss_f32: pit/quantization/gaussian.py:211-271, 273-345) and VQQuantizer.forward (vq_quantize_z_f32:
pit/quantization/vq.py:39-96) -- against goldens captured from the reference (g19, g20: tests/golden/make_golden_r6.py), the oracle,
and the torch-glue path of the same modules (what runs when autograd is on)."""
import copy
import os

import numpy as np
import pytest
import torch

from oracle import gq_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"
REF_INFO_KEYS = {"kl_loss", "bits-mean", "bits-min", "bits-max", "lam-min", "lam-max", "lam", "mu", "std", "zhat_noquant",
                 "indices", "zhat_quant"}          # gaussian.py:268-269, :329, :344


def _close(a, b, rel=2e-5):
    a, b = (float(v.detach()) if isinstance(v, torch.Tensor) else float(v) for v in (a, b))
    return abs(a - b) <= rel * max(1.0, abs(b))


# ------------------------------------------------------------------------------------------ GQ2
@pytest.mark.parametrize("tag,dim,n,dim_idx", [("a", 4, 1024, 1), ("b", 16, 4096, -1)])
@pytest.mark.parametrize("channels_last", [False, True])
def test_g19_gq2_eval_forward_vs_reference_golden(tag, dim, n, dim_idx, channels_last):
    """Three consecutive eval forwards (the lambda state machine moves): scalars of info within 2e-5, the lambdas EXACTLY the
    reference's Python floats, indices / zhat / zhat_quant bit-equal, std within an ulp (libm), every info key present."""
    from pit_hip.quantization.gaussian import GaussianQuantRegularizer2

    d = np.load(os.path.join(G, "g19_gq2_eval_forward.npz"))
    z = torch.from_numpy(d[f"{tag}_z"]).to(DEV)
    if channels_last:
        if z.dim() != 4:
            pytest.skip("channels_last is a 4-d layout")
        z = z.contiguous(memory_format=torch.channels_last)
    m = GaussianQuantRegularizer2(dim, n, dim_idx=dim_idx).eval().to(DEV)
    with torch.no_grad():
        for it in range(3):
            zi = z + 0.1 * it
            zhat, info = m(zi)
            assert set(info) == REF_INFO_KEYS
            for k, want in zip(("kl_loss", "bits-mean", "bits-min", "bits-max"), d[f"{tag}_scalars_{it}"]):
                assert _close(info[k], want), (it, k, float(info[k]), want)
                assert info[k].dim() == 0 and info[k].dtype == torch.float32
            lams = [float(info["lam"]), float(info["lam-min"]), float(info["lam-max"])]
            assert lams == list(d[f"{tag}_lams_{it}"]), (it, lams)
            assert np.array_equal(info["indices"].cpu().numpy(), d[f"{tag}_indices_{it}"])
            assert np.array_equal(info["zhat_quant"].cpu().numpy(), d[f"{tag}_zhat_quant_{it}"])
            assert torch.equal(zhat, info["zhat_quant"])                 # finite zhat_noquant: (g - g) + v == v
            np.testing.assert_allclose(info["std"].cpu().numpy(), d[f"{tag}_std_{it}"], rtol=2.4e-7)
            c = z.shape[dim_idx] // 2
            assert torch.equal(info["mu"], zi.narrow(dim_idx % z.dim(), 0, c))
            assert info["zhat_noquant"].shape == zhat.shape == info["mu"].shape
    assert [m.lam, m.lam_min, m.lam_max] == list(d[f"{tag}_lams_2"])       # the attributes pull the device state


def test_gq2_fused_forward_equals_the_torch_glue_path_at_config_shape():
    """BASELINE configs[3]'s gq2_0.25 shape (dim 16, n 65536, bs 4): fused call vs quant_gaussian + quant_vq of the same module
    (autograd path): same indices, same statistics, same lambda trajectory; zhat_noquant = mu + noise * std for the noise the
    fused path drew from torch's generator."""
    from pit_hip.quantization.gaussian import GaussianQuantRegularizer2

    g = torch.Generator().manual_seed(3)
    z = torch.cat([0.9 * torch.randn(4, 16, 32, 32, generator=g), -1.5 + 0.3 * torch.randn(4, 16, 32, 32, generator=g)], 1).to(DEV)
    fused = GaussianQuantRegularizer2(16, 65536).eval().to(DEV)
    glue = copy.deepcopy(fused)
    for it in range(3):
        zi = z * (1.0 + 0.2 * it)
        with torch.no_grad():
            torch.manual_seed(11 + it)
            zf, inf = fused(zi)
            torch.manual_seed(11 + it)
            noise = torch.randn(4, 16, 32 * 32, device=DEV).view(4, 16, 32, 32)     # the draw of _forward_fused ("bchw": [outer, C, inner])
        zg, ing = glue(zi.clone().requires_grad_(True))
        assert torch.equal(inf["indices"], ing["indices"]) and torch.equal(zf, zg.detach())
        for k in ("kl_loss", "bits-mean", "bits-min", "bits-max"):
            assert _close(inf[k], ing[k], 1e-5), (it, k, float(inf[k]), float(ing[k]))
        assert [float(inf[k]) for k in ("lam", "lam-min", "lam-max")] == [ing["lam"], ing["lam-min"], ing["lam-max"]]
        want = inf["mu"] + noise * inf["std"]
        assert torch.equal(inf["zhat_noquant"], want)
        np.testing.assert_allclose(inf["std"].cpu().numpy(), ing["std"].detach().cpu().numpy(), rtol=2.4e-7)
    assert (fused.lam, fused.lam_min, fused.lam_max) == (glue.lam, glue.lam_min, glue.lam_max)


def test_gq2_lambda_state_moves_between_host_and_device():
    """Writing an attribute makes the host copy authoritative (the next fused forward uploads it); the torch-glue path (train /
    autograd) continues from where the device left off; use_ste False returns the codes in eval."""
    from pit_hip.quantization.gaussian import GaussianQuantRegularizer2

    g = torch.Generator().manual_seed(4)
    z = torch.randn(2, 32, 8, 8, generator=g).to(DEV)
    m = GaussianQuantRegularizer2(4, 1024).eval().to(DEV)
    with torch.no_grad():
        _, i0 = m(z)
        m.lam = 2.0
        assert (m.lam, m.lam_min, m.lam_max) == (2.0, float(i0["lam-min"]), float(i0["lam-max"]))
        _, i1 = m(z)
    st, _ = O.gq2_quant_gaussian_stats(z.cpu().numpy(), 4, 1024, (2.0, float(i0["lam-min"]), float(i0["lam-max"])))
    assert _close(i1["kl_loss"], st["kl_loss"])
    host = (m.lam, m.lam_min, m.lam_max)
    m.train()
    _, i2 = m.quant_gaussian(z)                                   # torch glue: starts from the pulled device state
    _, want = O.gq2_quant_gaussian_stats(z.cpu().numpy(), 4, 1024, host)
    assert (m.lam, m.lam_min, m.lam_max) == want
    m2 = GaussianQuantRegularizer2(4, 1024, use_ste=False).eval().to(DEV)
    with torch.no_grad():
        zh, inf = m2(z)
    assert torch.equal(zh, inf["zhat_quant"]) and torch.equal(zh, m2.dequant(inf["indices"]))


def test_gq2_straight_through_value_is_nan_where_the_sample_is_not_finite():
    """zhat = zhat_g - zhat_g.detach() + zhat_v (gaussian.py:337-338): inf - inf = NaN in the reference; zhat_quant stays the code."""
    from pit_hip.quantization.gaussian import GaussianQuantRegularizer2

    g = torch.Generator().manual_seed(5)
    z = torch.randn(1, 32, 4, 4, generator=g)
    z[0, 3, 1, 2] = float("inf")
    m = GaussianQuantRegularizer2(16, 1024).eval().to(DEV)
    with torch.no_grad():
        zh, inf = m(z.to(DEV))
    bad = torch.zeros(1, 16, 4, 4, dtype=torch.bool)
    bad[0, 3, 1, 2] = True
    assert torch.isnan(zh.cpu()[bad]).all() and torch.isfinite(zh.cpu()[~bad]).all()
    assert torch.isfinite(inf["zhat_quant"]).all() and torch.equal(inf["zhat_quant"], m.dequant(inf["indices"]))


# ------------------------------------------------------------------------------------------ VQ
@pytest.mark.parametrize("tag,n,dim,K,legacy", [("k1", 4096, 16, 1, True), ("k2", 1024, 8, 2, True), ("nl", 2048, 8, 2, False)])
@pytest.mark.parametrize("channels_last", [False, True])
def test_g20_vq_eval_forward_vs_reference_golden(tag, n, dim, K, legacy, channels_last):
    from pit_hip.quantization.vq import VQQuantizer

    d = np.load(os.path.join(G, "g20_vq_eval_forward.npz"))
    vq = VQQuantizer("bchw", n, dim, codebook_num=K, legacy=legacy).eval().to(DEV)
    with torch.no_grad():
        vq.embedding.weight.copy_(torch.from_numpy(d[f"{tag}_emb"]))
        z = torch.from_numpy(d[f"{tag}_z"]).to(DEV)
        if channels_last:
            z = z.contiguous(memory_format=torch.channels_last)
        zq, info = vq(z)
    ind = info["indices"].cpu().numpy()
    clear = d[f"{tag}_gap"] > 1e-4
    assert ind.dtype == np.int64 and ind.shape == d[f"{tag}_indices"].shape
    assert np.array_equal(ind[clear], d[f"{tag}_indices"][clear])
    assert np.array_equal(ind, d[f"{tag}_indices"])                        # (no near-tie rows in these fixtures)
    assert np.array_equal(zq.cpu().numpy(), d[f"{tag}_zq"])                # z + (e - z), bit for bit
    assert _close(info["codebook_loss"], d[f"{tag}_loss"], 2e-6) and info["codebook_loss"].dim() == 0
    assert np.array_equal(vq.dequant(info["indices"]).detach().cpu().numpy(), O.vq_dequant(ind, d[f"{tag}_emb"], K))


@pytest.mark.parametrize("fmt,K,dim,n", [("bchw", 1, 16, 65536), ("blc", 2, 8, 4096), ("bchw", 4, 4, 65536), ("bchw", 1, 6, 512)])
def test_vq_fused_forward_equals_the_autograd_path_and_the_oracle(fmt, K, dim, n):
    """Fused call (no_grad) vs the torch-glue path (autograd on) of the same module, and both vs the oracle's restatement of
    vq.py:39-96 (fp64 arbiter): the MFMA filter dims, the dim-4 search, a dim without a filter (6: exhaustive kernel), "blc"
    (which the reference itself cannot run: its rearrange pattern is rejected by einops, vq.py:49)."""
    from pit_hip.quantization.vq import VQQuantizer

    g = torch.Generator().manual_seed(21 + dim)
    c = dim * K
    z = torch.randn(2, c, 16, 16, generator=g) if fmt == "bchw" else torch.randn(2, 256, c, generator=g)
    vq = VQQuantizer(fmt, n, dim, codebook_num=K).to(DEV)
    with torch.no_grad():
        vq.embedding.weight.copy_(torch.randn(n, dim, generator=g))
        zq, info = vq(z.to(DEV))
    zq_a, info_a = vq(z.to(DEV).requires_grad_(True))
    ozq, oind, oloss, gap = O.vq_forward_eval(z.numpy(), vq.embedding.weight.detach().cpu().numpy(), K, fmt, vq.beta, True)
    assert np.array_equal(info["indices"].cpu().numpy(), oind), f"min gap {gap.min():.2e}"
    assert torch.equal(info["indices"], info_a["indices"])
    assert np.array_equal(zq.cpu().numpy(), ozq) and torch.equal(zq, zq_a.detach())
    assert _close(info["codebook_loss"], oloss, 2e-6) and _close(info_a["codebook_loss"], oloss, 2e-6)
    assert zq_a.requires_grad and not zq.requires_grad


def test_vq_fused_forward_is_bit_reproducible_and_graph_capturable():
    """The loss is summed in a fixed order (ticketed partials): five runs, one value; the whole forward replays from a HIP graph."""
    from pit_hip.quantization.vq import VQQuantizer

    g = torch.Generator().manual_seed(33)
    z = torch.randn(4, 16, 64, 64, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    vq = VQQuantizer("bchw", 65536, 16).eval().to(DEV)
    with torch.no_grad():
        vq.embedding.weight.copy_(torch.randn(65536, 16, generator=g))
        zq0, i0 = vq(z)
        for _ in range(4):
            zq, i = vq(z)
            assert torch.equal(i["codebook_loss"], i0["codebook_loss"]) and torch.equal(zq, zq0) and torch.equal(i["indices"], i0["indices"])
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            vq(z)
        torch.cuda.current_stream().wait_stream(s)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            zq_g, i_g = vq(z)
        z.mul_(-1.0)
        gr.replay()
        zq1, i1 = vq(z)
    assert torch.equal(zq_g, zq1) and torch.equal(i_g["indices"], i1["indices"]) and torch.equal(i_g["codebook_loss"], i1["codebook_loss"])
    assert not torch.equal(i1["indices"], i0["indices"])


def test_gq2_fused_forward_replays_from_a_hip_graph_and_the_lambda_state_keeps_moving():
    """The lambda state lives on the device, so a captured forward advances it on every replay with no host in the loop: three
    replays leave the module where three eager forwards of a twin leave it (and the replayed outputs are the eager ones)."""
    from pit_hip.quantization.gaussian import GaussianQuantRegularizer2

    g = torch.Generator().manual_seed(8)
    z = torch.cat([0.9 * torch.randn(2, 16, 16, 16, generator=g), -1.5 + 0.3 * torch.randn(2, 16, 16, 16, generator=g)], 1).to(DEV)
    z = z.contiguous(memory_format=torch.channels_last)
    a = GaussianQuantRegularizer2(16, 4096).eval().to(DEV)
    b = copy.deepcopy(a)
    with torch.no_grad():
        a(z)                                        # warm-up: workspace, codebook cache, the device copy of the lambdas
        b(z)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            a(z)
        torch.cuda.current_stream().wait_stream(s)
        b(z)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            zh_g, info_g = a(z)
        for _ in range(3):
            gr.replay()
            zh_e, info_e = b(z)
        torch.cuda.synchronize()
    assert torch.equal(zh_g, zh_e) and torch.equal(info_g["indices"], info_e["indices"])
    for k in ("kl_loss", "bits-mean", "bits-min", "bits-max", "lam", "lam-min", "lam-max"):
        assert float(info_g[k]) == float(info_e[k]), k
    assert (a.lam, a.lam_min, a.lam_max) == (b.lam, b.lam_min, b.lam_max) and a.lam != 1.0


@pytest.mark.parametrize("dim,n,c", [(4, 65536, 16), (6, 512, 12), (8, 4096, 16), (32, 2048, 32)])
def test_gq2_fused_forward_on_every_launch_path_vs_the_oracle(dim, n, c):
    """The statistics block runs as one extra block of the re-rank launch at dims 8 / 16 / 32 and dim 4 without a search index, and as
    its own launch behind the dim-4 search (n >= 2^14) and the exhaustive kernel (a dim without a filter): same function, same order
    of additions -- every path against the oracle's restatement of quant_gaussian's statistics and of quant_vq, two forwards each."""
    from pit_hip.quantization.gaussian import GaussianQuantRegularizer2

    g = torch.Generator().manual_seed(40 + dim)
    z = torch.cat([1.2 * torch.randn(2, c, 16, 16, generator=g), -1.5 + 0.3 * torch.randn(2, c, 16, 16, generator=g)], 1)
    m = GaussianQuantRegularizer2(dim, n).eval().to(DEV)
    cb = m.prior_samples.cpu().numpy()
    state = (1.0, 1.0, 1.0)
    with torch.no_grad():
        for it in range(2):
            zi = z * (1.0 + 0.5 * it)
            zh, info = m(zi.to(DEV))
            want, state = O.gq2_quant_gaussian_stats(zi.numpy(), dim, n, state)
            for k in ("kl_loss", "bits-mean", "bits-min", "bits-max"):
                assert _close(info[k], want[k], 2e-6), (it, k, float(info[k]), float(want[k]))
            assert (float(info["lam"]), float(info["lam-min"]), float(info["lam-max"])) == state
            # operands as the kernels derive them (fp64 exp / log, rounded once) differ from numpy's fp32 chain by an ulp: strict on indices
            # only where the oracle's own top-2 gap is clear of that
            ozq, oind = O.gq2_quant_vq(zi.numpy(), cb, dim, 1)
            same = info["indices"].cpu().numpy() == oind
            assert same.mean() > 0.999, same.mean()
            el = np.repeat(same, dim, axis=1)
            assert np.array_equal(zh.cpu().numpy()[el], ozq[el])


# ------------------------------------------------------------------------------------------ randomized sweep of both fused forwards
def _fused_case(seed):
    rng = np.random.default_rng(1000 + seed)
    dim = int(rng.choice([4, 8, 16, 32, 6]))
    K = int(rng.choice([1, 2, 3])) if dim <= 8 else 1
    n = int(rng.choice([64, 1000, 4096, 16384, 65536] if dim != 6 else [64, 500]))
    B, h, w = int(rng.integers(1, 4)), int(rng.integers(1, 20)), int(rng.integers(1, 20))
    layout = str(rng.choice(["nchw", "channels_last", "last"]))
    kind = int(rng.integers(0, 3))
    return dim, K, n, B, h, w, layout, kind


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("GQ_FUSED_SEEDS", "16")))))
def test_randomized_fused_forwards_vs_the_oracle(seed):
    """Random shapes (ragged h x w, K sub-codebooks, every filter dim + a dim without one), layouts (NCHW, channels_last, channel axis
    last) and three kinds of z, both fused forwards.  GQ2: indices identical to gq_quantize_z_f32's on the same z AND to the oracle on
    the operands those kernels derived (strict), statistics and lambdas against the oracle's restatement; VQ: indices, straight-through
    value and loss against the oracle's fp64 arbiter."""
    from pit_hip import _lib
    from pit_hip.quantization.gaussian import GaussianQuantRegularizer2
    from pit_hip.quantization.vq import VQQuantizer

    dim, K, n, B, h, w, layout, kind = _fused_case(seed)
    c = dim * K
    g = torch.Generator().manual_seed(seed)
    mu = torch.randn(B, c, h, w, generator=g) * (0.9, 3.0, 0.3)[kind]
    lv = (-1.5 + 0.3 * torch.randn(B, c, h, w, generator=g), 12.0 * torch.rand(B, c, h, w, generator=g) - 9.0,
          0.1 * torch.randn(B, c, h, w, generator=g))[kind]
    z = torch.cat([mu, lv], 1)
    dim_idx = 1
    zd = z.to(DEV)
    if layout == "channels_last":
        zd = zd.contiguous(memory_format=torch.channels_last)
    elif layout == "last":
        zd, dim_idx = zd.permute(0, 2, 3, 1).contiguous(), -1
    m = GaussianQuantRegularizer2(dim, n, dim_idx=dim_idx).eval().to(DEV)
    with torch.no_grad():
        zh, info = m(zd)
    # the same kernels through the plain module-level entry, with the operands they derived
    zrows = z.permute(0, 2, 3, 1).reshape(1, -1, 2 * c).to(DEV)
    idx2, _, mu_r, sd_r = _lib.gq_quantize_z(zrows, m.prior_samples, dim, "blc", _lib.GQHIP_GROUP_CONTIGUOUS, return_operands=True)
    ind_rows = torch.movedim(info["indices"], dim_idx, -1).reshape(-1)
    assert torch.equal(ind_rows, idx2.reshape(-1))
    sd_np = sd_r.cpu().numpy()
    oi, _ = O.argmax_rows(mu_r.cpu().numpy(), sd_np, m.prior_samples.cpu().numpy(), 1.0,
                          logstd=np.log(sd_np.astype(np.float64)).astype(np.float32))
    assert np.array_equal(ind_rows.cpu().numpy(), oi), (seed, dim, K, n, layout, kind)
    assert torch.equal(zh, m.dequant(info["indices"])) and torch.equal(zh, info["zhat_quant"])
    want, state = O.gq2_quant_gaussian_stats(z.numpy(), dim, n, (1.0, 1.0, 1.0))
    for k in ("kl_loss", "bits-mean", "bits-min", "bits-max"):
        assert _close(info[k], want[k], 3e-6), (seed, k, float(info[k]), float(want[k]))
    assert (float(info["lam"]), float(info["lam-min"]), float(info["lam-max"])) == state

    vq = VQQuantizer("bchw", n, dim, codebook_num=K).eval().to(DEV)
    zv = mu if layout != "channels_last" else mu
    with torch.no_grad():
        vq.embedding.weight.copy_(torch.randn(n, dim, generator=g))
        zvd = zv.to(DEV).contiguous(memory_format=torch.channels_last) if layout == "channels_last" else zv.to(DEV)
        zq, vinfo = vq(zvd)
    ozq, oind, oloss, gap = O.vq_forward_eval(zv.numpy(), vq.embedding.weight.detach().cpu().numpy(), K, "bchw", vq.beta, True)
    assert np.array_equal(vinfo["indices"].cpu().numpy(), oind), (seed, float(gap.min()))
    assert np.array_equal(zq.cpu().numpy(), ozq) and _close(vinfo["codebook_loss"], oloss, 3e-6)
