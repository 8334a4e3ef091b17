"""-m gpu: the drop-in modules (pit_hip.*, gq_cuda) on a real MI355X against the golden
vectors captured from the reference and against the oracle on the same operands."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import gq_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
META = json.load(open(os.path.join(G, "meta.json")))
DEV = "cuda:0"


def load(name):
    return np.load(os.path.join(G, name))


def _rows_from_bchw(ind):
    return ind.transpose(0, 2, 3, 1).reshape(-1)


@pytest.mark.parametrize("name,dim,n", [("g2_dim16_n1024", 16, 1024), ("g2_dim8_n1024", 8, 1024),
                                        ("g2_dim4_n1024", 4, 1024), ("g2_dim16_n65536", 16, 65536)])
def test_g2_kernel_boundary_goldens_bit_exact(name, dim, n):
    """(mu, std, log std) exactly as the reference's CPU path derived them -> identical indices."""
    from pit_hip import _lib

    d = load(name + ".npz")
    cb = torch.from_numpy(O.codebook(n, dim, 42)).to(DEV)
    idx, zhat = _lib.gq_argmax(torch.from_numpy(d["mu"]).to(DEV), torch.from_numpy(d["std"]).to(DEV), cb, 1.0,
                               logsd=torch.from_numpy(d["logstd"]).to(DEV))
    assert np.array_equal(idx.cpu().numpy(), d["indices"])
    assert torch.equal(zhat, cb[idx])


@pytest.mark.parametrize("name", ["randn_seed0", "realistic_seed0"])
def test_g3_module_forward_matches_reference(name):
    from pit_hip import _lib
    from pit_hip.quantization.gaussian import GaussianQuantRegularizer

    d = load(f"g3_{name}.npz")
    q = GaussianQuantRegularizer("bchw", 65536, group=16, backend="hip").eval().to(DEV)
    z = torch.from_numpy(d["z"]).to(DEV)
    zhat, info = q(z)
    ind = info["indices"].cpu().numpy()
    assert ind.dtype == np.int64 and ind.shape == (1, 1, 32, 32)
    assert info["zhat_noquant"].shape == zhat.shape == (1, 16, 32, 32)
    # (1) strict: the operands the kernels derived (fp64 exp/log, rounded once) through the oracle
    idx2, _, mu_r, sd_r = _lib.gq_quantize_z(z, q.prior_samples, 16, "bchw", _lib.GQHIP_GROUP_STRIDED,
                                             return_operands=True)
    sd_np = sd_r.cpu().numpy()
    lsd = np.log(sd_np.astype(np.float64)).astype(np.float32)
    cb = q.prior_samples.cpu().numpy()
    oi, _, best, second = O.argmax_rows(mu_r.cpu().numpy(), sd_np, cb, 1.0, logstd=lsd, with_gap=True)
    assert np.array_equal(_rows_from_bchw(ind), oi)
    assert torch.equal(idx2, info["indices"])
    # (2) vs the reference's stored indices: exp/log are libm specific (<= 1 ulp); a differing
    # index is only legitimate where the reference's own top-2 gap is a rounding tie
    diff = _rows_from_bchw(ind) != _rows_from_bchw(d["indices"])
    assert diff.mean() < 1e-3 and np.all((best - second)[diff] < 1e-4)
    # std: ours is the correctly rounded exp; torch-CPU's differs by at most 1 ulp
    ref_std = O.torch_exp_half(np.clip(d["z"][:, 16:], -30, 20)).transpose(0, 2, 3, 1).reshape(-1, 16)
    ulp = np.abs(sd_np.view(np.int32).astype(np.int64) - ref_std.view(np.int32).astype(np.int64))
    assert ulp.max() <= 1
    # (3) dequant round trip, bit exact
    assert torch.equal(q.dequant(info["indices"]), zhat)
    assert np.array_equal(zhat.cpu().numpy(), O.gq1_dequant(ind, cb, 16))


def test_g4_layouts_strided_contiguous_blc():
    from pit_hip.quantization.gaussian import GaussianQuantRegularizer, GaussianQuantRegularizer2

    for name, group, fmt in (("g4_gq1_group4", 4, "bchw"), ("g4_gq1_group8", 8, "bchw"), ("g4_gq1_blc_group4", 4, "blc")):
        d = load(name + ".npz")
        q = GaussianQuantRegularizer(fmt, 2048, group=group, backend="hip").eval().to(DEV)
        zhat, info = q(torch.from_numpy(d["z"]).to(DEV))
        assert np.array_equal(info["indices"].cpu().numpy(), d["indices"]), name
        assert np.array_equal(zhat.cpu().numpy(), d["zhat"])
        assert torch.equal(q.dequant(info["indices"]), zhat)
    for dim_idx in (1, 2):
        d = load(f"g4_gq2_dimidx{dim_idx}.npz")
        q2 = GaussianQuantRegularizer2(4, 2048, dim_idx=dim_idx, backend="hip").eval().to(DEV)
        zv, iv = q2.quant_vq(torch.from_numpy(d["z"]).to(DEV))
        assert np.array_equal(iv["indices"].cpu().numpy(), d["indices"])
        assert np.array_equal(zv.cpu().numpy(), d["zhat"])
        assert torch.equal(q2.dequant(iv["indices"]), zv)
        zhat, info = q2(torch.from_numpy(d["z"]).to(DEV))  # STE mix: numerically the VQ output
        assert torch.allclose(zhat, zv, atol=1e-5)
        assert {"indices", "zhat_quant", "kl_loss", "mu", "std", "zhat_noquant"} <= set(info)


def test_g5_edge_rows():
    from pit_hip.quantization.gaussian import GaussianQuantRegularizer

    d = load("g5_edges.npz")
    q = GaussianQuantRegularizer("bchw", 2048, group=16, backend="hip").eval().to(DEV)
    q.prior_samples.copy_(torch.from_numpy(d["cb"]))   # in-place edit: nothing cached from the old codebook exists
    zhat, info = q(torch.from_numpy(d["z"]).to(DEV))
    ind = info["indices"].cpu().numpy()
    assert np.array_equal(ind, d["indices"])
    assert ind[0, 0, 0, 3] == 7  # duplicated codeword -> first index


def test_backend_cuda_is_the_fused_path_and_cuda_compat_the_reference_call_sequence(monkeypatch):
    """backend="cuda" (what every shipped GQ YAML says, configs/sd3unet_gq_0.25.yaml:33): the fused kernels, no score matrix, the
    reference golden's indices; backend="cuda-compat" (or "cuda" + GQHIP_COMPAT=1): gq_cuda.ops.gq_cuda -> argmax -> index_select
    (gaussian.py:124-133)."""
    import gq_cuda
    from pit_hip.quantization.gaussian import GaussianQuantRegularizer

    d = load("g3_realistic_seed0.npz")
    zdev = torch.from_numpy(d["z"]).to(DEV)
    monkeypatch.delenv("GQHIP_COMPAT", raising=False)
    qf = GaussianQuantRegularizer("bchw", 65536, group=16, backend="cuda").eval().to(DEV)
    _, info_f = qf(zdev)
    assert qf.perturbed is None                                   # no 4.29 GB matrix on the zero-edit route
    qh = GaussianQuantRegularizer("bchw", 65536, group=16, backend="hip").eval().to(DEV)
    zh_h, info_h = qh(zdev)                                       # pinned against the reference golden by test_g3_*
    assert torch.equal(info_f["indices"], info_h["indices"]) and torch.equal(qf(zdev)[0], zh_h)
    assert (info_f["indices"].cpu().numpy() != d["indices"]).mean() < 1e-3
    monkeypatch.setenv("GQHIP_COMPAT", "1")
    _, info_e = qf(zdev)
    assert qf.perturbed is not None and qf.perturbed.shape == (1024, 65536)
    monkeypatch.delenv("GQHIP_COMPAT")
    q = GaussianQuantRegularizer("bchw", 65536, group=16, backend="cuda-compat").eval().to(DEV)
    zhat, info = q(zdev)
    assert q.perturbed.shape == (1024, 65536) and torch.equal(info["indices"], info_e["indices"])
    ind = info["indices"].cpu().numpy()
    cb = q.prior_samples.cpu().numpy()
    zf = d["z"]
    _, oind = O.gq1_forward(zf, cb, 16)
    agree = (ind == oind).mean()
    assert agree > 0.995  # different (CUDA-kernel) formula: same arg-max except rounding ties
    # schema / error behaviour of the op
    with pytest.raises(RuntimeError):
        gq_cuda.ops.gq_cuda(torch.zeros(2, 16), torch.ones(2, 16), q.prior_samples, q.perturbed, 16, 2, 65536, 1.0)
    with pytest.raises(RuntimeError):
        gq_cuda.ops.gq_cuda(torch.zeros(2, 16, device=DEV), torch.ones(3, 16, device=DEV), q.prior_samples,
                            q.perturbed, 16, 2, 65536, 1.0)
    assert "extension_cpp::gq" in str(torch.ops.extension_cpp.gq.default._schema)


def test_g6_vq_and_lfq():
    from pit_hip.quantization.lfq import LFQQuantizer
    from pit_hip.quantization.vq import VQQuantizer

    for name, n, dim, k in (("g6_vq_k1", 4096, 16, 1), ("g6_vq_k2", 1024, 8, 2)):
        d = load(name + ".npz")
        vq = VQQuantizer("bchw", n, dim, codebook_num=k).eval().to(DEV)
        vq.embedding.weight.data.copy_(torch.from_numpy(d["emb"]))
        zq, info = vq(torch.from_numpy(d["z"]).to(DEV))
        ind = info["indices"].cpu().numpy()
        _, oind, gap = O.vq_forward(d["z"], d["emb"], k, with_gap=True)
        assert np.array_equal(ind, oind)                     # fp64 arbiter on both sides: exact
        clear = d["gap"] > 1e-4
        assert np.array_equal(ind[clear], d["indices"][clear])  # the reference's fp32 einsum path
        assert np.array_equal(vq.dequant(info["indices"]).detach().cpu().numpy(), O.vq_dequant(oind, d["emb"], k))
        assert torch.allclose(zq.detach(), vq.dequant(info["indices"]).detach(), atol=1e-6)
    d = load("g6_lfq.npz")
    lfq = LFQQuantizer("bchw", codebook_size=256, num_codebooks=2).eval().to(DEV)
    q, info = lfq(torch.from_numpy(d["x"]).to(DEV))
    assert np.array_equal(info["indices"].cpu().numpy(), d["indices"])
    assert np.array_equal(q.cpu().numpy(), d["q"])
    assert np.array_equal(lfq.dequant(info["indices"]).cpu().numpy(), d["q"])


def test_vq_lfq_512_shapes():
    """BASELINE config 5 shapes: 512x512 inputs -> 4096 positions per image."""
    from pit_hip import _lib

    g = torch.Generator().manual_seed(7)
    emb = torch.randn(65536, 16, generator=g)
    z = torch.randn(2 * 4096, 16, generator=g)
    idx, zq = _lib.vq_argmin(z.to(DEV), emb.to(DEV))
    sel = np.arange(0, z.shape[0], 16)
    oi = O.vq_argmin_rows(z.numpy()[sel], emb.numpy())
    assert np.array_equal(idx.cpu().numpy()[sel], oi)
    assert torch.equal(zq, emb.to(DEV)[idx])
    x = torch.randn(2 * 4096, 16, generator=g)
    li, lq = _lib.lfq_pack(x.to(DEV))
    _, oi = O.lfq_forward(x.numpy().reshape(2, 4096, 16), fmt="blc")
    assert np.array_equal(li.cpu().numpy(), oi.reshape(-1))
    assert torch.equal(_lib.lfq_unpack(li, 16), lq)


@pytest.mark.e2e
def test_engine_end_to_end_full_config():
    """x -> encode -> indices -> dequant/decode on the GPU vs the reference's CPU end-to-end golden.
    Gates (fp32, different conv/GN/SDPA kernels): |z_enc diff| <= 5e-5; at most 2 of 1024 indices differ end to end and
    only where the reference's top-2 gap < 1e-3 (golden z through the GPU quantiser: none, or gap < 1e-4);
    reconstruction max-abs <= 5e-3 when the indices agree and PSNR(gpu recon, cpu recon) >= 60 dB."""
    from pit_hip.models.autoencoder import AutoencodingEngine

    unet = dict(attn_type="vanilla", double_z=True, z_channels=16, resolution=256, in_channels=3, out_ch=3, ch=128,
                ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[32], dropout=0.0)
    torch.manual_seed(1234)
    vae = AutoencodingEngine(
        encoder_config={"target": "pit.modules.unet.Encoder", "params": unet},
        decoder_config={"target": "pit.modules.unet.Decoder", "params": unet},
        regularizer_config={"target": "pit.quantization.gaussian.GaussianQuantRegularizer",
                            "params": {"format": "bchw", "group": 16, "n_samples": 65536, "backend": "cuda"}},
    ).eval()
    assert vae.regularization.backend == "cuda"      # the shipped YAML's value, untouched: it lands on the fused path
    vae = vae.to(DEV)
    d = load("g7_full_e2e.npz")
    gx = torch.Generator().manual_seed(1000)
    x = (torch.rand(1, 3, 256, 256, generator=gx) * 2 - 1).to(DEV)
    with torch.no_grad():
        z_enc = vae.encode(x, unregularized=True)[0]
        z, ind = vae.quant(x)
        rec = vae.dequant(ind)
        z2, rec2, log = vae(x)
    from bench import GATES     # the ONE definition of the end-to-end gates (bench.py prints the same numbers)

    dz = float((z_enc.cpu() - torch.from_numpy(d["z_enc"])).abs().max())
    assert dz <= GATES["z_enc_max_abs"], dz   # measured 3-4e-6 (fp32 conv rounding); 2e-3 in round 1 was far too loose
    got, want = ind.cpu().numpy(), d["indices"]
    diff = _rows_from_bchw(got) != _rows_from_bchw(want)
    # GPU encoder + GPU quantiser vs the reference end to end: what README/DESIGN claim is "equal"; the gate allows the
    # encoder's rounding to flip at most 2 of the 1024 rows, and only where the reference's own top-2 gap is < 1e-3
    print(f"e2e 256 NCHW: |dz| {dz:.2e}, {int(diff.sum())} of 1024 indices differ"
          f"{' (gaps ' + str(d['gap'][diff]) + ')' if diff.any() else ''}")
    assert diff.sum() <= GATES["indices_differing_per_1024"] and np.all(d["gap"][diff] < GATES["near_tie_gap"]), (diff.sum(), d["gap"][diff])
    ref = torch.from_numpy(d["x_rec"].astype(np.float32))
    if not diff.any():
        assert float((rec.cpu() - ref).abs().max()) <= GATES["recon_max_abs_if_indices_equal"]    # the golden is stored in fp16 (ulp 4.9e-4 at |x| ~ 1)
    mse = float(((rec.cpu() - ref) ** 2).mean())
    assert 10 * np.log10(4.0 / max(mse, 1e-20)) >= (GATES["recon_psnr_db"] if diff.any() else GATES["recon_psnr_db_if_indices_equal"])
    # two separate forward passes of the NCHW module (ATen / MIOpen convolutions, whose picks we do not control): the
    # second pass may differ at rounding level and an index only at a near-tie
    assert float((z2 - z).abs().max()) <= 1e-4 or int((log["indices"] != ind).sum()) <= 2
    assert int((log["indices"] != ind).sum()) <= 2
    assert torch.allclose(rec, rec2, atol=1e-3)
    # the bench configuration: channels_last conv stack (NHWC fused kernels, Winograd encoder) -- same gate
    vae_cl = vae.to(memory_format=torch.channels_last)
    with torch.no_grad():
        z_cl, ind_cl = vae_cl.quant(x.contiguous(memory_format=torch.channels_last))
        rec_cl = vae_cl.dequant(ind_cl)
        # the product path (every kernel ours or a pinned library GEMM): a second pass gives the SAME bits
        z_cl2, ind_cl2 = vae_cl.quant(x.contiguous(memory_format=torch.channels_last))
        assert torch.equal(z_cl2, z_cl) and torch.equal(ind_cl2, ind_cl) and torch.equal(vae_cl.dequant(ind_cl2), rec_cl)
    diff_cl = _rows_from_bchw(ind_cl.cpu().numpy()) != _rows_from_bchw(want)
    print(f"e2e 256 channels_last: {int(diff_cl.sum())} of 1024 indices differ")
    assert diff_cl.sum() <= GATES["indices_differing_per_1024"] and np.all(d["gap"][diff_cl] < GATES["near_tie_gap"]), (diff_cl.sum(), d["gap"][diff_cl])
    mse_cl = float(((rec_cl.cpu() - ref) ** 2).mean())
    assert 10 * np.log10(4.0 / max(mse_cl, 1e-20)) >= (GATES["recon_psnr_db"] if diff_cl.any() else GATES["recon_psnr_db_if_indices_equal"])
    vae = vae_cl.to(memory_format=torch.contiguous_format)
    # golden z_enc fed straight to the GPU quantiser: no conv rounding in between, so the indices must be the
    # reference's (a row may differ only where its top-2 gap is below the 1-ulp libm difference of exp / log: < 1e-4)
    zhat, info = vae.regularization(torch.from_numpy(d["z_enc"]).to(DEV))
    diff2 = _rows_from_bchw(info["indices"].cpu().numpy()) != _rows_from_bchw(want)
    assert diff2.sum() == 0 or np.all(d["gap"][diff2] < GATES["same_z_gap"]), (diff2.sum(), d["gap"][diff2])


def test_histogram_and_u16_wire_format():
    from pit_hip import _lib

    g = torch.Generator().manual_seed(3)
    idx = torch.randint(0, 65536, (16, 1, 32, 32), generator=g)
    di = idx.to(DEV)
    h = _lib.index_histogram(di, 65536).cpu().numpy()
    assert np.array_equal(h, np.bincount(idx.reshape(-1).numpy(), minlength=65536))
    u = _lib.indices_to_u16(di)
    assert u.dtype == torch.uint16 and u.shape == idx.shape
    assert torch.equal(_lib.indices_from_u16(u).cpu(), idx)


@pytest.mark.e2e
def test_eval_loop_single_rank_on_gpu():
    from pit_hip.eval_dist import evaluate_sharded
    from pit_hip.models.autoencoder import AutoencodingEngine

    unet = dict(attn_type="vanilla", double_z=True, z_channels=16, resolution=32, in_channels=3, out_ch=3, ch=32,
                ch_mult=[1, 2, 2], num_res_blocks=1, attn_resolutions=[8], dropout=0.0)
    torch.manual_seed(1234)
    vae = AutoencodingEngine(
        encoder_config={"target": "pit.modules.unet.Encoder", "params": unet},
        decoder_config={"target": "pit.modules.unet.Decoder", "params": unet},
        regularizer_config={"target": "pit.quantization.gaussian.GaussianQuantRegularizer",
                            "params": {"format": "bchw", "group": 16, "n_samples": 1024}},
    ).eval().to(DEV)

    def images_for(ids):
        g = torch.Generator().manual_seed(5)
        bank = torch.rand(10, 3, 32, 32, generator=g) * 2 - 1
        return bank[ids]

    out = evaluate_sharded(vae, images_for, 10, 4, 0, 1, torch.device(DEV), tokens_per_image=64)
    assert out["indices"].shape == (8, 64) and out["psnr"].shape == (8,)
    with torch.no_grad():
        _, ind = vae.quant(images_for([0, 1, 2, 3]).to(DEV))
    # (a second forward pass of the NCHW conv stack -- ATen / MIOpen kernels: equal up to their non-reproducibility at near-ties)
    assert int((out["indices"][:4].cpu() != ind.reshape(4, -1).cpu()).sum()) <= 2


@pytest.mark.e2e
def test_smoke_entry():
    import __graft_entry__ as ge

    ge.smoke()


def test_g10_bsq_and_fsq_modules():
    """BSQ / FSQ eval + dequant through the same boundary."""
    from pit_hip.quantization.bsq import BSQQuantizer
    from pit_hip.quantization.fsq import FSQQuantizer

    d = load("g10_bsq.npz")
    bsq = BSQQuantizer("bchw", codebook_size=2, num_codebooks=16).eval().to(DEV)
    q, info = bsq(torch.from_numpy(d["x"]).to(DEV))
    assert np.array_equal(info["indices"].cpu().numpy(), d["indices"])
    np.testing.assert_allclose(q.cpu().numpy(), d["q"], atol=1e-6)      # F.normalize on device: fp32 rounding
    assert np.array_equal(bsq.dequant(info["indices"]).cpu().numpy(), d["deq"])
    assert info["indices"].shape == (2, 1, 8, 8) and info["indices"].dtype == torch.int64

    d = load("g10_fsq.npz")
    levels = d["levels"].tolist()
    fsq = FSQQuantizer(levels, "bchw").eval().to(DEV)
    zhat, info = fsq(torch.from_numpy(d["x"]).to(DEV))
    got = info["indices"].cpu().numpy()
    assert got.dtype == np.int32 and got.shape == d["indices"].shape
    # tanh/atanh are libm specific: an index may differ only where the bounded value sits on a
    # rounding boundary (|distance| < 1e-5); everywhere else bit-equal.
    diff = got != d["indices"]
    assert np.all(d["margin"][diff] < 1e-5) and diff.mean() < 0.01
    same = ~np.broadcast_to(diff, d["zhat"].shape[:1] + (1,) + d["zhat"].shape[2:]).repeat(len(levels), 1)
    assert np.array_equal(zhat.cpu().numpy()[same], d["zhat"][same])
    assert np.array_equal(fsq.dequant(info["indices"]).cpu().numpy(), O.fsq_dequant(got, levels))
    assert abs(float(info["bits"]) - float(np.sum(np.log2(levels)) * 6 * 64)) < 1e-2


@pytest.mark.convstack
@pytest.mark.parametrize("shape", [(2, 128, 64, 64), (3, 256, 16, 16), (1, 512, 32, 32), (2, 64, 6, 10)])
def test_fused_groupnorm_silu_matches_torch(shape):
    """gn_silu_f32 vs ATen GroupNorm -> SiLU (fp32; tolerance 2e-5 abs / 1e-5 rel: different
    reduction order and v_exp-based sigmoid)."""
    import torch.nn.functional as F
    from pit_hip import _lib

    g = torch.Generator().manual_seed(shape[1])
    x = (torch.randn(*shape, generator=g) * 2.0 + 0.3).to(DEV)
    gn = torch.nn.GroupNorm(32, shape[1], eps=1e-6).to(DEV)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(shape[1], generator=g))
        gn.bias.copy_(torch.randn(shape[1], generator=g))
        want_n = gn(x)
        want_s = F.silu(want_n)
        got_s = _lib.gn_silu(x, gn.weight, gn.bias, 32, 1e-6, silu=True)
        got_n = _lib.gn_silu(x, gn.weight, gn.bias, 32, 1e-6, silu=False)
    torch.testing.assert_close(got_n, want_n, atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(got_s, want_s, atol=2e-5, rtol=1e-5)


@pytest.mark.convstack
def test_gn_prebias_and_add_bias_kernels():
    from pit_hip import _lib

    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 16, 24, generator=g).to(DEV)
    pb = torch.randn(64, generator=g).to(DEV)
    gn = torch.nn.GroupNorm(32, 64, eps=1e-6).to(DEV)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(64, generator=g)); gn.bias.copy_(torch.randn(64, generator=g))
        want = torch.nn.functional.silu(gn(x + pb[None, :, None, None]))
        got = _lib.gn_silu(x, gn.weight, gn.bias, 32, 1e-6, silu=True, pre_bias=pb)
        torch.testing.assert_close(got, want, atol=2e-5, rtol=1e-5)
        y = torch.randn(2, 64, 16, 24, generator=g).to(DEV)
        assert torch.equal(_lib.add_bias(x, y, None), x + y)
        torch.testing.assert_close(_lib.add_bias(x, y, pb), x + y + pb[None, :, None, None], atol=1e-6, rtol=1e-6)


@pytest.mark.convstack
@pytest.mark.parametrize("shape", [(2, 128, 32, 32), (1, 256, 16, 24), (2, 512, 8, 8)])
def test_fused_groupnorm_nhwc_and_add_bias_nhwc(shape):
    import torch.nn.functional as F
    from pit_hip import _lib

    g = torch.Generator().manual_seed(shape[1] + 1)
    x = (torch.randn(*shape, generator=g) * 1.5 - 0.2).to(DEV).contiguous(memory_format=torch.channels_last)
    pb = torch.randn(shape[1], generator=g).to(DEV)
    gn = torch.nn.GroupNorm(32, shape[1], eps=1e-6).to(DEV)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(shape[1], generator=g)); gn.bias.copy_(torch.randn(shape[1], generator=g))
        want = F.silu(gn(x + pb[None, :, None, None]))
        got = _lib.gn_silu(x, gn.weight, gn.bias, 32, 1e-6, silu=True, pre_bias=pb)
        assert got.is_contiguous(memory_format=torch.channels_last)
        torch.testing.assert_close(got, want, atol=2e-5, rtol=1e-5)
        torch.testing.assert_close(_lib.gn_silu(x, gn.weight, gn.bias, 32, 1e-6, silu=False), gn(x), atol=2e-5, rtol=1e-5)
        y = torch.randn(*shape, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
        torch.testing.assert_close(_lib.add_bias(x, y, pb), x + y + pb[None, :, None, None], atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize("z_channels,K", [(16, 1), (32, 2)])
def test_gq2_shipped_config_and_flagged_k2_variant(z_channels, K):
    """BASELINE config 4: configs/sd3unet_gq2_0.25.yaml (dim 16, codebook 65536, z_channels 16 -> K = 1
    as shipped) and the flagged K = 2 variant (z_channels 32); module vs oracle restatement of
    GaussianQuantRegularizer2.quant_vq (contiguous channel groups) on 2 images, bit-exact operands path."""
    from pit_hip import _lib
    from pit_hip.quantization.gaussian import GaussianQuantRegularizer2

    q2 = GaussianQuantRegularizer2(dim=16, codebook_size=65536, backend="cuda").eval().to(DEV)   # as shipped (sd3unet_gq2_0.25.yaml)
    g = torch.Generator().manual_seed(40 + K)
    c = z_channels
    z = torch.cat([0.9 * torch.randn(2, c, 32, 32, generator=g), -1.5 + 0.3 * torch.randn(2, c, 32, 32, generator=g)], 1)
    zhat, info = q2(z.to(DEV))
    ind = info["indices"]
    assert ind.shape == (2, K, 32, 32) and zhat.shape == (2, c, 32, 32)
    assert torch.equal(q2.dequant(ind), info["zhat_quant"])
    # operands the kernels derived (position-major rows, contiguous groups) through the oracle
    zf = z.permute(0, 2, 3, 1).reshape(1, -1, 2 * c).contiguous()
    idx2, _, mu_r, sd_r = _lib.gq_quantize_z(zf.to(DEV), q2.prior_samples, 16, "blc", _lib.GQHIP_GROUP_CONTIGUOUS,
                                             return_operands=True)
    assert torch.equal(idx2.reshape(2, 32, 32, K).permute(0, 3, 1, 2), ind)
    sdn = sd_r.cpu().numpy()
    sel = np.arange(0, sdn.shape[0], 4)
    oi, _ = O.argmax_rows(mu_r.cpu().numpy()[sel], sdn[sel], q2.prior_samples.cpu().numpy(), 1.0,
                          logstd=np.log(sdn[sel].astype(np.float64)).astype(np.float32))
    assert np.array_equal(idx2.reshape(-1).cpu().numpy()[sel], oi)
    # contiguous grouping: sub-codebook k <- channels [k*16, (k+1)*16)
    mu_ref = z[:, :c].permute(0, 2, 3, 1).reshape(-1, K, 16).reshape(-1, 16)
    assert torch.equal(mu_r.cpu(), mu_ref)


def test_quantizer_is_hip_graph_capturable():
    """The C-ABI entry points never synchronise or allocate, so the fused quantiser (prep -> filter -> re-rank ->
    fallback) can be captured in a HIP graph and replayed on new data."""
    from pit_hip.quantization.gaussian import GaussianQuantRegularizer

    q = GaussianQuantRegularizer("bchw", 65536, group=16, backend="hip").eval().to(DEV)
    g = torch.Generator().manual_seed(77)
    mk = lambda: torch.cat([0.9 * torch.randn(2, 16, 32, 32, generator=g), -1.5 + 0.3 * torch.randn(2, 16, 32, 32, generator=g)], 1)
    z_static = mk().to(DEV)
    from pit_hip import _lib

    run = lambda: _lib.gq_quantize_z(z_static, q.prior_samples, 16, "bchw", _lib.GQHIP_GROUP_STRIDED,
                                     ws=q._ws)
    run()  # warm-up: sizes the workspace outside the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        idx_g, zhat_g = run()
    for _ in range(3):
        z_new = mk().to(DEV)
        z_static.copy_(z_new)
        graph.replay()
        torch.cuda.synchronize()
        idx_e, zhat_e = _lib.gq_quantize_z(z_new, q.prior_samples, 16, "bchw", _lib.GQHIP_GROUP_STRIDED)
        assert torch.equal(idx_g, idx_e) and torch.equal(zhat_g, zhat_e)


@pytest.mark.convstack
def test_upsample2x_nhwc_matches_interpolate():
    import torch.nn.functional as F
    from pit_hip import _lib

    x = torch.randn(2, 64, 5, 7).to(DEV).contiguous(memory_format=torch.channels_last)
    y = _lib.upsample2x_nhwc(x)
    assert y.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(y, F.interpolate(x, scale_factor=2.0, mode="nearest"))


def test_rccl_single_rank_gather_of_packed_record():
    """RCCL (backend "nccl") end to end on this box with world_size 1: process-group init bound to
    the device, the packed int32 record through all_gather_into_tensor, MAX all-reduce of the
    step time and a barrier -- the exact calls bench.py / evaluate_sharded make at N > 1."""
    import socket
    import torch.distributed as dist
    from pit_hip.eval_dist import StepRecord, gather_step

    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device(DEV))
    try:
        bs, tokens = 4, 1024
        layout = StepRecord(bs, tokens, n_metrics=1)
        g = torch.Generator().manual_seed(5)
        idx = torch.randint(0, 65536, (bs, 1, 32, 32), generator=g).to(DEV)
        psnr = torch.rand(bs, 1, generator=g).to(DEV)
        rec = layout.pack(idx, psnr)
        out = gather_step(rec, 1, always_collective=True)
        assert out.shape == (1, layout.words) and out.is_cuda
        i2, m2 = layout.unpack(out[:, None])
        assert torch.equal(i2.reshape(-1), idx.reshape(-1)) and torch.equal(m2.reshape(-1), psnr.reshape(-1))
        t = torch.tensor([1.25], dtype=torch.float64, device=DEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        assert float(t.item()) == 1.25
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("filt", ["auto", "fp32"])
@pytest.mark.parametrize("rows,n,dim", [(256, 1024, 8), (1000, 4096, 16), (3000, 65536, 16), (500, 5000, 32)])
def test_vq_multi_candidate(rows, n, dim, filt):
    """VQ arg-min on random data, where the split-bf16 filter's margin leaves several candidate groups per row
    for the exact fp64 arbiter: indices must equal the brute-force fp64 arg-min wherever it is not a rounding tie."""
    from pit_hip import _lib

    g = torch.Generator().manual_seed(7 + rows)
    emb = torch.randn(n, dim, generator=g)
    z = torch.randn(rows, dim, generator=g)
    zz, ee = (z.double() ** 2).sum(1, keepdim=True), (emb.double() ** 2).sum(1)[None]
    d = zz + ee - 2 * z.double() @ emb.double().T
    ref = d.argmin(1)
    _lib.set_filter(filt)
    try:
        idx, zq = _lib.vq_argmin(z.to(DEV), emb.to(DEV))
    finally:
        _lib.set_filter("auto")
    idx = idx.cpu()
    gap = d.gather(1, idx[:, None])[:, 0] - d.gather(1, ref[:, None])[:, 0]
    assert float(gap.max()) <= 1e-9, f"{int((gap > 1e-9).sum())} rows picked a farther codeword (max gap {float(gap.max()):.3g})"
    assert torch.equal(zq.cpu(), emb[idx])


@pytest.mark.convstack
def test_add_bias_stats_and_gn_apply_match_the_unfused_pair():
    """Residual add that leaves the next GroupNorm's statistics behind + apply-only GroupNorm == add, then GN."""
    import torch.nn.functional as F
    from pit_hip import _lib

    g = torch.Generator().manual_seed(3)
    for C, H in ((128, 24), (256, 10), (512, 7)):
        a = torch.randn(3, C, H, H + 1, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
        b = torch.randn(3, C, H, H + 1, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
        bias = torch.randn(C, generator=g).to(DEV)
        gamma, beta = torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
        y, stats = _lib.add_bias_stats(a, b, bias, 32)
        ref = a + b + bias[None, :, None, None]
        assert torch.equal(y, ref) and y.is_contiguous(memory_format=torch.channels_last)
        for silu in (True, False):
            out = _lib.gn_apply(y, gamma, beta, 32, 1e-6, silu, stats)
            exp = F.group_norm(ref, 32, gamma, beta, 1e-6)
            exp = F.silu(exp) if silu else exp
            assert torch.allclose(out, exp, atol=2e-5, rtol=1e-5)
            assert torch.allclose(out, _lib.gn_silu(y, gamma, beta, 32, 1e-6, silu=silu), atol=1e-6, rtol=1e-6)


@pytest.mark.convstack
def test_upsample_fallback_route_matches_upsample_then_conv():
    """Upsample at shapes the direct sub-pixel kernel does not tile (H % 8, W % 32, or no statistics on the input): libgqhip's
    NHWC upsample copy + the ordinary convolution routes == interpolate followed by the 3x3 conv.  (The tiled shapes:
    tests/test_gpu_convstack_kernels.py::test_upconv2x_direct_matches_fp64.)"""
    import torch.nn.functional as F
    from pit_hip.modules import unet as U

    torch.manual_seed(1)
    for ch, H, W in ((64, 9, 6), (256, 16, 16)):
        up = U.Upsample(ch).eval().to(DEV).to(memory_format=torch.channels_last)
        x = torch.randn(2, ch, H, W).to(DEV).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            y, b = up(x)
            xu = F.interpolate(x, scale_factor=2.0, mode="nearest")
            ref = F.conv2d(xu, up.conv.weight, up.conv.bias, 1, 1)
            ref64 = F.conv2d(xu.double(), up.conv.weight.double(), up.conv.bias.double(), 1, 1)
            mag = F.conv2d(xu.double().abs(), up.conv.weight.double().abs(), None, 1, 1)      # sum |x||w| per output
        assert y.shape == (2, ch, 2 * H, 2 * W) and y.is_contiguous(memory_format=torch.channels_last)
        got = (y if b is None else y + b[None, :, None, None]).detach()
        # error model instead of a measured atol: every route here accumulates in fp32 over K = 9 ch terms -- at most
        # ~2^-24 sqrt(K)-ish of sum|x||w| in practice, charged generously at 4e-7 (the fp32 library GEMM's own figure is 2.6-3.5e-7,
        # Winograd F(4x4)'s transforms amplify by a few); the library reference itself is held to the same yardstick
        e_got = float(((got.double() - ref64).abs() / mag).max())
        e_ref = float(((ref.double() - ref64).abs() / mag).max())
        print(f"upsample fallback {ch} ch {H}x{W}: error / sum|x||w|: this route {e_got:.2e}, F.conv2d {e_ref:.2e}")
        assert e_got <= max(4e-7 * (4 if ch >= 128 else 1), 2.0 * e_ref), (e_got, e_ref)

