"""-m gpu, round 3: what VERDICT r2 asked to harden -- the tokenizer is bit-reproducible run to run (order-independent
GroupNorm statistics, the encoder's conv_out / decoder's conv_in on libgqhip's fixed-order fp32 matrix-core convolution
instead of MIOpen's split-K pick), and end-to-end index parity over eight images (8 192 rows) against a golden captured
from the reference on CPU (tests/golden/make_golden_r3.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"
FULL = dict(attn_type="vanilla", double_z=True, z_channels=16, resolution=256, in_channels=3, out_ch=3, ch=128,
            ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[32], dropout=0.0)


def _engine(reg_target="pit.quantization.gaussian.GaussianQuantRegularizer", reg_params=None, unet=FULL, seed=1234):
    from pit_hip.models.autoencoder import AutoencodingEngine

    torch.manual_seed(seed)
    reg_params = reg_params or {"format": "bchw", "group": 16, "n_samples": 65536, "backend": "hip"}
    return AutoencodingEngine(encoder_config={"target": "pit.modules.unet.Encoder", "params": unet},
                              decoder_config={"target": "pit.modules.unet.Decoder", "params": unet},
                              regularizer_config={"target": reg_target, "params": reg_params}).eval()


def _rows(ind):   # [B, K, h, w] -> rows (b, l, k)
    return np.asarray(ind).transpose(0, 2, 3, 1).reshape(-1)


# ------------------------------------------------------------------------------------------ fixed-order fp32 convolution
@pytest.mark.convstack
@pytest.mark.parametrize("cin,cout,H,W,gn", [(512, 32, 32, 32, True), (512, 16, 32, 32, True), (128, 32, 8, 64, False),
                                              (16, 512, 32, 32, False), (8, 40, 5, 32, False), (32, 64, 4, 96, False),
                                              (512, 32, 8, 8, True), (16, 512, 8, 8, False), (64, 8, 3, 45, False)])
def test_conv3x3_f32_matches_fp64_and_is_bit_reproducible(cin, cout, H, W, gn):
    """conv3x3_f32 (gq_conv_f32.h) against an fp64 convolution: both tilings (K split over the waves for >= 64 input channels,
    output channels split otherwise), fused GroupNorm + SiLU, a partial last channel tile (Cout 16 / 40), non-square images.
    Error gate: fp32 accumulation over K = 9 Cin terms, charged against sum |x||w|.  Five runs: identical bits."""
    from pit_hip import _lib

    B = 3
    g = torch.Generator().manual_seed(100 + cin + cout)
    x = (torch.randn(B, cin, H, W, generator=g) * 1.5 + 0.2).to(DEV).contiguous(memory_format=torch.channels_last)
    conv = torch.nn.Conv2d(cin, cout, 3, 1, 1).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.05)
        conv.bias.copy_(torch.randn(cout, generator=g))
    wk = _lib.conv_f32_weights(conv.weight)
    gn_t, xin = None, x.double()
    if gn:
        norm = torch.nn.GroupNorm(32, cin, eps=1e-6).to(DEV)
        with torch.no_grad():
            norm.weight.copy_(torch.rand(cin, generator=g) + 0.5)
            norm.bias.copy_(torch.randn(cin, generator=g) * 0.3)
        pre = (torch.randn(cin, generator=g) * 0.1).to(DEV)
        gn_t = (norm.weight, norm.bias, 32, 1e-6, True, _lib.gn_stats(x, 32, pre), pre)
        xin = torch.nn.functional.silu(torch.nn.functional.group_norm(x.double() + pre.double()[None, :, None, None], 32,
                                                                      norm.weight.double(), norm.bias.double(), 1e-6))
    with torch.no_grad():
        ref = torch.nn.functional.conv2d(xin, conv.weight.double(), conv.bias.double(), 1, 1)
        sc = torch.nn.functional.conv2d(xin.abs(), conv.weight.double().abs(), None, 1, 1) + conv.bias.double().abs()[None, :, None, None]
        y = _lib.conv3x3_f32(x, wk, cout, bias=conv.bias, gn=gn_t)
        assert y.shape == (B, cout, H, W) and y.is_contiguous(memory_format=torch.channels_last)
        err = float(((y.double() - ref).abs() / sc).max())
        print(f"conv3x3_f32 {cin}->{cout} {H}x{W} gn={gn}: max err {err:.2e} of sum|x||w|")
        assert err <= (3e-6 if gn else 1e-6), err       # GN: the fp32 normalisation itself is ~1e-6 of |x|
        for _ in range(5):
            assert torch.equal(_lib.conv3x3_f32(x, wk, cout, bias=conv.bias, gn=gn_t), y)


@pytest.mark.convstack
def test_groupnorm_statistics_are_order_independent_and_poison_loudly():
    """gq_stats.h: integer-limb accumulation.  The same tensor through kernels with different thread -> element maps
    (NHWC statistics kernel on x, the residual add's fused statistics on a + b = x): both must agree with the fp64 sums to
    fp32-partial accuracy and, each on its own, give identical bits on every run; a non-finite input poisons the record
    (NaN out) instead of producing a garbage integer."""
    from pit_hip import _lib

    g = torch.Generator().manual_seed(5)
    x = (torch.randn(4, 128, 64, 64, generator=g) * 3 + 0.5).to(DEV).contiguous(memory_format=torch.channels_last)
    a = (torch.randn(4, 128, 64, 64, generator=g)).to(DEV).contiguous(memory_format=torch.channels_last)
    b = x - a
    xs = a + b                                   # what the fused add sees (fp32 a + b, not bitwise x)
    st = _lib.gn_stats(xs, 32)
    for _ in range(10):
        assert torch.equal(_lib.gn_stats(xs, 32), st)
    _, st2 = _lib.add_bias_stats(a, b, torch.zeros(128, device=DEV), 32)
    for _ in range(10):
        assert torch.equal(_lib.add_bias_stats(a, b, torch.zeros(128, device=DEV), 32)[1], st2)
    xd = xs.double().permute(0, 2, 3, 1).reshape(4, 64 * 64, 32, 4)
    want = torch.stack([xd.sum((1, 3)), (xd ** 2).sum((1, 3))], -1).flatten()
    for s in (st, st2):
        v = _lib.gn_stats_values(s)
        assert torch.allclose(v, want, rtol=2e-6, atol=1e-3), float((v - want).abs().max())
    # tiny activations keep their statistics (limb 0 reaches 2^-56)
    tiny = (xs * 1e-6).contiguous(memory_format=torch.channels_last)
    vt = _lib.gn_stats_values(_lib.gn_stats(tiny, 32))
    td = tiny.double().permute(0, 2, 3, 1).reshape(4, 64 * 64, 32, 4)
    wt = torch.stack([td.sum((1, 3)), (td ** 2).sum((1, 3))], -1).flatten()
    assert torch.allclose(vt, wt, rtol=1e-5, atol=1e-14), float((vt - wt).abs().max())
    bad = xs.clone()
    bad[1, 5, 3, 3] = float("inf")
    vb = _lib.gn_stats_values(_lib.gn_stats(bad.contiguous(memory_format=torch.channels_last), 32)).reshape(4, 32, 2)
    assert torch.isnan(vb[1, 5 // 4]).all() and not torch.isnan(vb[0]).any() and not torch.isnan(vb[1, 3]).any()


# ------------------------------------------------------------------------------------------ run-to-run reproducibility
@pytest.mark.e2e
@pytest.mark.parametrize("size,batches", [(256, (1, 4, 16)), (512, (1, 4, 16))])
def test_encoder_and_decoder_are_bit_reproducible(size, batches):
    """VERDICT r2 next #1a: encoder(x) bit-identical across 20 calls at B = 1, 4, 16, at 256^2 and 512^2 (the reference's
    CPU path is deterministic, pit/quantization/gaussian.py:136-150 sees ONE z per image); so are the tokens and the
    decoder.  channels_last = the bench configuration."""
    vae = _engine().to(DEV).to(memory_format=torch.channels_last)
    for B in batches:
        g = torch.Generator().manual_seed(9 + B + size)
        x = (torch.rand(B, 3, size, size, generator=g) * 2 - 1).to(DEV).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            z0 = vae.encoder(x)
            zq0, info0 = vae.regularization(z0)
            r0 = vae.decode(zq0)
            runs = 20 if B * size * size <= 16 * 256 * 256 else 8
            for i in range(runs):
                z = vae.encoder(x)
                assert torch.equal(z, z0), f"encoder run {i} at B={B}, {size}^2 differs: max {float((z - z0).abs().max()):.2e}"
            for i in range(4):
                zq, info = vae.regularization(vae.encoder(x))
                assert torch.equal(info["indices"], info0["indices"]) and torch.equal(zq, zq0)
                assert torch.equal(vae.decode(zq0), r0), f"decoder run {i} at B={B}, {size}^2 differs"


@pytest.mark.e2e
def test_nchw_encoder_tokens_follow_z_between_passes():
    """The NCHW module (no channels_last conversion: ATen / MIOpen convolutions, libgqhip's NCHW GroupNorm) is NOT claimed to be
    bit-reproducible: MIOpen's pick for conv_out (512 -> 32) is a split-K kernel with floating-point atomics, and conv3x3_f32
    needs channels_last (README / INTEGRATION: only the channels_last path is bit-reproducible).  What we own is asserted: two passes
    whose z agree give identical tokens; when z differs by the library's rounding, at most a near-tie token moves."""
    vae = _engine().to(DEV)
    g = torch.Generator().manual_seed(77)
    x = (torch.rand(2, 3, 256, 256, generator=g) * 2 - 1).to(DEV)
    with torch.no_grad():
        z0 = vae.encoder(x)
        z1 = vae.encoder(x)
        i0 = vae.regularization(z0)[1]["indices"]
        i1 = vae.regularization(z1)[1]["indices"]
    if torch.equal(z0, z1):
        assert torch.equal(i0, i1)
    else:   # a library kernel of the NCHW route is not reproducible: report it, the channels_last route is the product path
        print(f"NCHW route: z differs by {float((z0 - z1).abs().max()):.2e} between two passes (conv library)")
        assert int((i0 != i1).sum()) <= 2


# ------------------------------------------------------------------------------------------ 8-image end-to-end golden
@pytest.mark.e2e
@pytest.mark.parametrize("channels_last", [True, False])
def test_g14_eight_images_end_to_end_vs_reference_golden(channels_last):
    """8 images at 256^2 (eval.py:144-151 feeds batches) through GPU encoder -> GPU quantiser -> GPU
    decoder against the reference's CPU path (golden g14: z, indices, top-2 gaps, reconstruction).  Gate per 1024 rows as
    for g7: |dz| <= 5e-5, at most 2 indices differ and only where the reference's own top-2 gap is < 1e-3; the golden z
    through the GPU quantiser: identical indices except where the gap is below the libm difference (< 1e-4)."""
    d = np.load(os.path.join(G, "g14_e2e_8x256.npz"))
    vae = _engine().to(DEV)
    gx = torch.Generator().manual_seed(3256)
    x = (torch.rand(8, 3, 256, 256, generator=gx) * 2 - 1).to(DEV)
    if channels_last:
        vae = vae.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        z_enc = vae.encode(x, unregularized=True)[0]
        z, ind = vae.quant(x)
        rec = vae.dequant(ind)
    from bench import GATES     # the ONE definition of the end-to-end gates

    dz = float((z_enc.cpu() - torch.from_numpy(d["z_enc"])).abs().max())
    assert dz <= GATES["z_enc_max_abs"], dz
    got, want, gap = _rows(ind.cpu().numpy()), _rows(d["indices"]), d["gap"]
    diff = got != want
    per_image = diff.reshape(8, 1024).sum(1)
    print(f"g14 e2e 8 x 256^2 (channels_last={channels_last}): |dz| {dz:.2e}, {int(diff.sum())} of 8192 indices differ"
          f"{' at gaps ' + str(gap[diff]) if diff.any() else ''}; rows with gap < 1e-3 in the golden: {int((gap < 1e-3).sum())}")
    assert per_image.max() <= GATES["indices_differing_per_1024"] and np.all(gap[diff] < GATES["near_tie_gap"]), (per_image, gap[diff])
    ref = torch.from_numpy(d["x_rec"].astype(np.float32))
    same = ~diff.reshape(8, 1024).any(1)
    if same.any():    # images whose tokens all agree: the reconstruction is the reference's up to conv rounding (golden is fp16)
        assert float((rec.cpu()[same] - ref[same]).abs().max()) <= GATES["recon_max_abs_if_indices_equal"]
    mse = float(((rec.cpu() - ref) ** 2).mean())
    assert 10 * np.log10(4.0 / max(mse, 1e-20)) >= (GATES["recon_psnr_db"] if diff.any() else GATES["recon_psnr_db_if_indices_equal"])
    zhat, info = vae.regularization(torch.from_numpy(d["z_enc"]).to(DEV))
    diff2 = _rows(info["indices"].cpu().numpy()) != want
    assert diff2.sum() == 0 or np.all(gap[diff2] < GATES["same_z_gap"]), (diff2.sum(), gap[diff2])


# ------------------------------------------------------------------------------------------ self-invalidating weight caches
@pytest.mark.e2e
@pytest.mark.parametrize("which", ["decoder", "encoder"])
def test_weight_caches_notice_data_writes_without_any_call(which):
    """VERDICT r2 next #6 / ADVICE: `conv.weight.data.mul_()` bumps no version counter, and nobody calls
    `invalidate_caches()` here.  The forward's content-hash guard (unet._WeightGuard, gqhip_checksum_tensors) must notice
    the new bytes -- conv weights, GroupNorm gamma / beta (the fp16 operand bounds depend on them), biases -- rebuild every
    weight-derived cache and return the answer of the NEW weights: compared with the direct NCHW path (no caches).  A
    third forward with unchanged weights is bit-identical to the second (no spurious rebuild changes anything)."""
    from pit_hip.modules import unet as U

    torch.manual_seed(3)
    if which == "decoder":
        mod = U.Decoder(**FULL).eval().to(DEV).to(memory_format=torch.channels_last)
        x = torch.randn(2, 16, 32, 32, device=DEV)
    else:
        mod = U.Encoder(**FULL).eval().to(DEV).to(memory_format=torch.channels_last)
        x = (torch.rand(2, 3, 256, 256, device=DEV) * 2 - 1)
    assert U.WEIGHT_GUARD
    with torch.no_grad():
        y0 = mod(x).float().contiguous()
        for p in mod.parameters():                      # an "EMA swap": every parameter rewritten through .data
            p.data.mul_(1.0 + 0.2 * torch.rand_like(p))
        y1 = mod(x).float().contiguous()                # NO invalidate_caches()
        y2 = mod(x).float().contiguous()
        mod.conv_in.weight.data[0, 0, 0, 0] += 0.5     # ... and a single element of a single tensor
        y3 = mod(x).float().contiguous()
        ref3 = mod.to(memory_format=torch.contiguous_format)(x).float().contiguous()   # direct convolutions, nothing cached
    assert float((y1 - y0).abs().max()) > 1e-3          # the weights did change the output
    assert torch.equal(y1, y2)
    assert float((y3 - y2).abs().max()) > 1e-4
    scale = max(float(ref3.abs().max()), 1.0)
    assert float((y3 - ref3).abs().max()) <= 2e-4 * scale, (float((y3 - ref3).abs().max()), scale)


@pytest.mark.convstack
def test_checksum_tensors_sees_every_word():
    from pit_hip import _lib

    g = torch.Generator().manual_seed(1)
    ts = [torch.randn(n, generator=g).to(DEV) for n in (1, 3, 4, 5, 1024, 100003, 2359296)]
    table = _lib.checksum_table(ts)
    out = torch.empty(len(ts), dtype=torch.int64, device=DEV)
    base = _lib.checksum_tensors(table, out).clone()
    assert torch.equal(_lib.checksum_tensors(table, out), base)          # deterministic
    for k, t in enumerate(ts):
        for pos in {0, t.numel() // 2, t.numel() - 1}:
            old = t[pos].clone()
            t[pos] = old + 1.0
            cur = _lib.checksum_tensors(table, out).clone()
            assert cur[k] != base[k] and all(cur[j] == base[j] for j in range(len(ts)) if j != k), (k, pos)
            t[pos] = old
    a, b = ts[4][10].clone(), ts[4][11].clone()                            # a swap of two elements is seen too (position salt)
    ts[4][10], ts[4][11] = b, a
    assert _lib.checksum_tensors(table, out)[4] != base[4]


# ------------------------------------------------------------------------------------------ checkpoint-shaped weights
from ckpt_like import checkpoint_like_ as _checkpoint_like_  # noqa: E402  (shared with tests/golden/make_golden_r4.py)


# ------------------------------------------------------------------------------------------ compat op: every tiling path
@pytest.mark.parametrize("dim,rows,n", [(16, 600, 65536 + 40), (4, 300, 4096 + 33), (32, 520, 8192), (8, 37, 65536), (16, 257, 96),
                                        (16, 1, 32), (32, 1, 33), (32, 129, 65)])
def test_compat_scores_tile_pairs_chunk_rotation_and_ragged_edges(dim, rows, n):
    """gq_scores_f32's matrix-core kernel beyond the small cases of test_compat_scores_op_matches_cuda_formula: several row
    blocks (the chunk order is rotated by the row block), several chunks per code split with a ragged last one, tile pairs
    (two tiles leave as one 256-byte run per row) with the pair order rotated by the wave, a codebook that ends inside a
    pair, a row count that ends inside a wave's tile.  Against the per-pair restatement of gq_cuda.cu:31-38 (oracle)."""
    import oracle.gq_oracle as O
    from pit_hip import _lib

    g = torch.Generator().manual_seed(dim * 1000 + rows)
    mu = 0.9 * torch.randn(rows, dim, generator=g)
    sd = torch.exp(0.5 * (-1.5 + 0.3 * torch.randn(rows, dim, generator=g)))
    cb = torch.randn(n, dim, generator=g).clamp(-4.6, 4.6)
    out = torch.full((rows, n), float("nan"), device=DEV)
    _lib.gq_scores(mu.to(DEV), sd.to(DEV), cb.to(DEV), out, 1.0)
    got = out.cpu().numpy()
    ref = O.cuda_formula_scores(mu.numpy(), sd.numpy(), cb.numpy(), 1.0)
    assert np.isfinite(got).all()                    # every element written (the buffer started as NaN)
    scale = np.abs(ref).max(axis=1, keepdims=True)
    assert np.abs(got - ref).max() <= 2e-5 * scale.max()
    np.testing.assert_allclose(got, ref, rtol=5e-5, atol=2e-5 * float(scale.max()))
    # the same kernel launched again writes the same bits (fixed tiling, no atomics)
    out2 = torch.empty_like(out)
    _lib.gq_scores(mu.to(DEV), sd.to(DEV), cb.to(DEV), out2, 1.0)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("dim,beta", [(16, 1.0), (32, 1.0), (16, 0.25)])
def test_compat_scores_fp16_products_ranges_and_out_of_range_chunks(dim, beta):
    """Dims 16 / 32 of gq_scores_f32 run as three fp16 products of two-term splits (csrc/gq_scores_f16.h).  What that form has
    to get right beyond random data: rows whose coefficients 1/sd^2 span 1e-6 ... 1e8 (per-row power-of-two scaling), one
    dominant dimension, large means, a chunk of codes with a value outside fp16's range (|n| > 255: recomputed by the per-pair
    formula in the kernel's second pass) or with an infinite one, tiny code values (fp16 subnormals: absolute error), a row with
    sd = 0 (non-finite row, neighbours untouched).  Gate: |out - fp64| <= 8e-7 of sum_i |terms| per element (+ the absolute floor of
    sub-normal code values) -- the level of the fp32 kernels (measured on random data: 2.9-3.8e-7 here, 4.5-6.5e-7 for the fp32 MFMA kernel, 4.9-5.0e-7 per pair)."""
    from pit_hip import _lib

    rows, n = 200, 4096 + 17
    g = torch.Generator().manual_seed(dim + int(beta * 100))
    mu = 0.9 * torch.randn(rows, dim, generator=g)
    sd = torch.exp(0.5 * (-1.5 + 0.3 * torch.randn(rows, dim, generator=g)))
    cb = torch.randn(n, dim, generator=g).clamp(-4.6, 4.6)
    sd[3, :] = 1e-4
    sd[4, :] = 1e3
    sd[5, 2] = 1e-5
    mu[6, :] = 50.0
    sd[8, :] = torch.logspace(-3, 2, dim)
    cb[700, 3] = 300.0             # out of fp16's range for n^2: its chunk takes the second pass
    cb[2000, :] = 1e-5             # squares far below fp16's subnormals
    cb[2001, :] = 0.0
    cb[3000, 1] = float("inf")
    sd[7, 0] = 0.0
    out = torch.full((rows, n), float("nan"), device=DEV)
    _lib.gq_scores(mu.to(DEV), sd.to(DEV), cb.to(DEV), out, beta)
    got = out.cpu().double()
    m, s_, c = mu.double()[:, None, :], sd.double()[:, None, :], cb.double()[None, :, :]
    with np.errstate(all="ignore"):
        ref = (-((c - m) / s_) ** 2 + beta * c * c).sum(-1)
        terms = (((beta - 1.0 / s_ ** 2).abs() * c * c) + (2 * m / s_ ** 2 * c).abs() + (m / s_) ** 2).sum(-1)
    ok_rows = torch.ones(rows, dtype=torch.bool)
    ok_rows[7] = False
    ok_cols = torch.ones(n, dtype=torch.bool)
    ok_cols[3000] = False
    sub = lambda t: t[ok_rows][:, ok_cols]
    assert torch.isfinite(sub(got)).all()
    # code values below fp16's normal range (|n| or n^2 < 2^-14 after the split) keep an absolute quantisation <= 2^-25 each:
    # the floor 2^-24 sum_i (|A'_i| + |B'_i|) next to the relative term (row 4 x code 2000 is that case: n^2 = 1e-10)
    floor = 2.0 ** -24 * ((beta - 1.0 / s_ ** 2).abs() + (2 * m / s_ ** 2).abs()).sum(-1).expand(rows, n)
    excess = (sub(got) - sub(ref)).abs() - floor[ok_rows][:, ok_cols]
    rel = excess / sub(terms)
    assert float(rel.max()) <= 8e-7, float(rel.max())
    assert not torch.isfinite(got[7]).any()                       # sd = 0: the whole row is inf / NaN, as in the per-pair formula
    assert not torch.isfinite(got[ok_rows][:, 3000]).any()        # an infinite code value: that column only
    # the chunk that went through the second pass (codes 512 ... 767 at 8 tiles of 32) agrees with the oracle's per-pair formula
    import oracle.gq_oracle as O

    ref32 = O.cuda_formula_scores(mu.numpy(), sd.numpy(), cb.numpy(), beta)
    blk = got[ok_rows][:, 512:768].float().numpy()
    np.testing.assert_allclose(blk, ref32[ok_rows.numpy()][:, 512:768], rtol=2e-5, atol=2e-5 * float(np.abs(ref32[ok_rows.numpy()][:, 512:768]).max()))


# ------------------------------------------------------------------------------------------ shapes the tilings do not divide
