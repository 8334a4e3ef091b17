"""-m gpu: every route of the conv stack (pit_hip/modules/unet.py's switches and shape fallbacks) against an fp64 evaluation of the
same function on the same input, gated by an error model (tests/convstack_ref.py) -- never route against route, never through
upstream library layers under a measured tolerance (such a test failed on a fresh box by 3 % of its tolerance and
hid 188 tests behind `pytest -x`).  Collected LAST (tests/conftest.py): a failure here cannot hide quantiser parity.

Reference functions: pit/modules/unet.py:137-153 (ResnetBlock), :185-206 (AttnBlock), :90-97 (Downsample), :69-73 (Upsample),
:411-436 / :554-587 (Encoder / Decoder forward)."""
import contextlib

import pytest
import torch
import torch.nn.functional as F

import convstack_ref as R

pytestmark = [pytest.mark.gpu, pytest.mark.convstack]
DEV = "cuda:0"
TOY = dict(attn_type="vanilla", ch=128, out_ch=3, in_channels=3, resolution=64, z_channels=16, double_z=True, ch_mult=[1, 2, 4, 4],
           num_res_blocks=2, attn_resolutions=[8], dropout=0.0)
FULL = dict(attn_type="vanilla", double_z=True, z_channels=16, resolution=256, in_channels=3, out_ch=3, ch=128,
            ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[32], dropout=0.0)


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


@contextlib.contextmanager
def switch(**flags):
    from pit_hip.modules import unet as U

    old = {k: getattr(U, k) for k in flags}
    try:
        for k, v in flags.items():
            setattr(U, k, v)
        yield
    finally:
        for k, v in old.items():
            setattr(U, k, v)


def _stv(stats):
    from pit_hip import _lib

    return _lib.gn_stats_values(stats)


def _stats_follow_output(y, what):
    """The GroupNorm statistics a kernel leaves with its output are sums over THAT output (fp32 partial sums per thread, exact
    integer accumulation across threads): relative 2e-6 of the sum of magnitudes."""
    st = getattr(y, "_gn_stats", None)
    assert st is not None, what
    y64 = R.d64(y)
    got = _stv(st[0]).double()
    want = R.stats_of(y64, st[1])
    mag = R.stats_of(y64.abs(), st[1])
    assert float(((got - want).abs() / (mag + 1e-30)).max()) <= 2e-6, what


# ------------------------------------------------------------------------------------------ whole encoder / decoder, every switch
@pytest.fixture(scope="module")
def toy():
    """The 64 x 64 toy UNet (every level narrower than some kernel's tile -> a mix of own kernels and library fallbacks), both
    layouts, and its fp64 twin's outputs -- computed once."""
    from pit_hip.modules import unet as U

    torch.manual_seed(1234)
    enc, dec = U.Encoder(**TOY).eval().to(DEV), U.Decoder(**TOY).eval().to(DEV)
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(2, 3, 64, 64, generator=g) * 2 - 1).to(DEV)
    zin = torch.randn(2, 16, 8, 8, generator=g).to(DEV)
    with torch.no_grad():
        z64 = R.twin64(enc)(R.d64(x))
        x64 = R.twin64(dec)(R.d64(zin))
    return dict(enc=enc, dec=dec, x=x, zin=zin, z64=z64, x64=x64)


SWITCHES = ["default", "FUSED_GN", "DEFER_BIAS", "WINOGRAD", "FUSED_WINO_TAIL", "WINOGRAD_F4", "WINOGRAD_F16X3", "WINOGRAD_OWN_GEMM",
            "DIRECT_CONV", "DIRECT_CONV_S2", "DIRECT_CONV_1X1", "FUSED_CONV_OUT", "CONV_F32", "FUSED_WINO_GN", "FUSED_WINO_GN_F4",
            "DIRECT_UPCONV", "FUSED_QKV", "FUSED_ADD_STATS", "STATS_ARENA", "CONV_IN_SMALL", "ATTN_F16X3"]


@pytest.mark.parametrize("channels_last", [True, False])
@pytest.mark.parametrize("off", SWITCHES)
def test_every_route_switch_meets_the_contract_against_fp64(toy, off, channels_last):
    """Encoder and decoder with one switch of pit_hip.modules.unet turned off at a time (the A/B switches double as the fallback
    routes of shapes a kernel does not tile), channels_last (own kernels) and NCHW (ATen convolutions + fused GroupNorm): each
    setting by itself meets the product's stated tolerance against the fp64 twin on the same input."""
    enc, dec = toy["enc"], toy["dec"]
    fmt = torch.channels_last if channels_last else torch.contiguous_format
    enc, dec = enc.to(memory_format=fmt), dec.to(memory_format=fmt)
    x, zin = toy["x"].contiguous(memory_format=fmt), toy["zin"].contiguous(memory_format=fmt)
    flags = {} if off == "default" else {off: False}
    with switch(**flags), torch.no_grad():
        z = enc(x)
        xr = dec(zin)
    tag = f"{off} off, {'channels_last' if channels_last else 'NCHW'}"
    R.contract_z(z, toy["z64"], "encoder, " + tag)
    R.contract_x(xr, toy["x64"], "decoder, " + tag)


@pytest.mark.parametrize("which", ["encoder", "decoder"])
def test_bench_shape_product_path_meets_the_contract_with_checkpoint_like_weights(which):
    """The fp16 x 3 routes scale their operands by powers of two derived from RIGOROUS bounds (sqrt(n - 1) max|gamma| + max|beta| for
    everything a GroupNorm feeds, row sums for the attention operands).  A checkpoint has gamma over two decades and beta of a few
    units, which makes those bounds ~10x looser and pushes small operands toward fp16's subnormal range.  At the bench shape
    (256 x 256), with checkpoint-like weights: the channels_last product path and the NCHW library path each against fp64."""
    from ckpt_like import checkpoint_like_
    from pit_hip.modules import unet as U

    torch.manual_seed(11)
    if which == "encoder":
        mod, x = U.Encoder(**FULL).eval(), torch.rand(1, 3, 256, 256) * 2 - 1
    else:
        mod, x = U.Decoder(**FULL).eval(), torch.randn(1, 16, 32, 32)
    checkpoint_like_(mod, 5)
    mod, x = mod.to(DEV), x.to(DEV)
    gate = R.contract_z if which == "encoder" else R.contract_x
    with torch.no_grad():
        ref = R.twin64(mod)(R.d64(x))
        gate(mod(x), ref, f"{which}, checkpoint-like weights, NCHW library path")
        mod = mod.to(memory_format=torch.channels_last)
        gate(mod(_cl(x)), ref, f"{which}, checkpoint-like weights, channels_last product path")


@pytest.mark.parametrize("B,H,W", [(3, 256, 320), (2, 264, 200), (1, 40, 24)])
def test_odd_batches_and_non_square_sizes_meet_the_contract(B, H, W):
    """Batch sizes and image sizes that the conv-stack kernels' tiles do not divide (every size is a multiple of 8, as the reference
    requires): the channels_last product path -- whatever mix of own kernels and library fallbacks the route logic picks per
    layer -- against the fp64 twin: z within the contract, tokens equal to the tokens of the twin's z except at near-ties (the
    end-to-end gate of bench.GATES), reconstruction of the SAME zhat within the contract.  (Bit-reproducibility is claimed -- and tested, test_gpu_e2e_goldens.py -- for the
    shapes the own kernels tile; here some layers fall back to library convolutions.)"""
    from bench import GATES
    from pit_hip.models.autoencoder import AutoencodingEngine

    torch.manual_seed(1234)
    eng = AutoencodingEngine(encoder_config={"target": "pit.modules.unet.Encoder", "params": FULL},
                             decoder_config={"target": "pit.modules.unet.Decoder", "params": FULL},
                             regularizer_config={"target": "pit.quantization.gaussian.GaussianQuantRegularizer",
                                                 "params": {"format": "bchw", "group": 16, "n_samples": 65536, "backend": "hip"}}).eval()
    eng = eng.to(DEV).to(memory_format=torch.channels_last)
    g = torch.Generator().manual_seed(B * 1000 + H + W)
    x = _cl((torch.rand(B, 3, H, W, generator=g) * 2 - 1).to(DEV))
    with torch.no_grad():
        z1 = eng.encoder(x)
        zq1, i1 = eng.regularization(z1)
        r1 = eng.decode(zq1)
        z64 = R.twin64(eng.encoder)(R.d64(x))
        _, i64 = eng.regularization(z64.float())
        r64 = R.twin64(eng.decoder)(R.d64(zq1))
    R.contract_z(z1, z64, f"encoder {B} x {H} x {W}")
    nd = int((i1["indices"] != i64["indices"]).sum())
    assert nd <= GATES["indices_differing_per_1024"] * max(1, i1["indices"].numel() // 1024), nd
    R.contract_x(r1, r64, f"decoder {B} x {H} x {W}")


def test_weight_caches_follow_data_writes_after_invalidate():
    """The Winograd / sub-pixel / fused-QKV matrices are cached per weight (data_ptr, _version); a write through `.data` bumps no
    version, so the documented route is `invalidate_caches()` (init_from_ckpt and load_state_dict call it themselves).  After it the
    channels_last path computes the function of the NEW weights (fp64 twin of the rewritten module)."""
    from pit_hip.modules.unet import Decoder

    cfg = dict(TOY, num_res_blocks=1)
    torch.manual_seed(3)
    dec = Decoder(**cfg).eval().to(DEV).to(memory_format=torch.channels_last)
    z = torch.randn(2, 16, 8, 8, device=DEV)
    with torch.no_grad():
        y0 = dec(z).float().contiguous()
        for p in dec.parameters():                      # an "EMA swap": every weight rewritten through .data
            p.data.mul_(1.0 + 0.05 * torch.rand_like(p))
        dec.invalidate_caches()
        y1 = dec(z).float().contiguous()
        ref = R.twin64(dec)(R.d64(z))
    assert float((y1 - y0).abs().max()) > 1e-3          # the weights did change the output
    R.contract_x(y1, ref, "decoder after .data writes + invalidate_caches")


def test_conv_in_small_keeps_the_encoder_free_of_library_convolutions_at_the_bench_shape():
    """With conv_in on libgqhip the channels_last encoder calls no MIOpen convolution at 256 x 256 (MIOpen's immediate mode ran a
    process's first eight conv_in calls on a 4 ms naive kernel).  Checked by counting torch's convolution dispatches; with the
    switch off exactly that one convolution goes to the library; both settings meet the contract against fp64."""
    from pit_hip.modules import unet as U

    torch.manual_seed(1234)
    enc = U.Encoder(**FULL).eval().to(DEV).to(memory_format=torch.channels_last)
    x = _cl(torch.rand(2, 3, 256, 256, device=DEV) * 2 - 1)
    calls = []
    real = torch.nn.functional.conv2d

    def counting(*a, **k):
        calls.append(tuple(a[0].shape))
        return real(*a, **k)

    torch.nn.functional.conv2d = counting
    U.F.conv2d = counting
    try:
        with torch.no_grad():
            z1 = enc(x)
            n_on = len(calls)
            with switch(CONV_IN_SMALL=False):
                z0 = enc(x)
    finally:
        torch.nn.functional.conv2d = real
        U.F.conv2d = real
    assert n_on == 0, calls[:n_on]
    assert len(calls) == 1 and calls[0] == (2, 3, 256, 256), calls
    with torch.no_grad():
        z64 = R.twin64(enc)(R.d64(x))
    R.contract_z(z1, z64, "encoder, conv_in on libgqhip")
    R.contract_z(z0, z64, "encoder, conv_in on the library")


# ------------------------------------------------------------------------------------------ blocks: propagated error models
def test_resnet_block_direct_and_winograd_routes_each_match_fp64():
    """The 128-channel ResnetBlock with its 3x3 convolutions on the direct fp16 x 3 kernel and -- switch off -- on Winograd
    F(2x2,3x3): each against the propagated fp64 model of the block."""
    from pit_hip.modules import unet as U

    torch.manual_seed(5)
    blk = U.ResnetBlock(128, 128, 0.0).eval().to(DEV).to(memory_format=torch.channels_last)
    U.mark_winograd(blk)
    x = _cl(torch.randn(2, 128, 16, 32, device=DEV))
    t = R.twin64(blk)
    with torch.no_grad():
        for direct, c in ((True, R.C_F16X3), (False, R.C_WINO_F2)):
            ref, bound = R.resnet_ref_and_bound(t, R.d64(x), None, c, c)
            with switch(DIRECT_CONV=direct):
                y = blk(x)
            R.bound_gate(y, ref, bound, f"ResnetBlock 128, {'direct' if direct else 'Winograd F(2x2,3x3)'}")
            _stats_follow_output(y, "ResnetBlock statistics")


def test_shortcut_and_attention_pointwise_routes_each_match_fp64():
    """ResnetBlock with a channel change (nin_shortcut, with and without a pending bias) and AttnBlock (q | k | v, proj_out +
    residual add) with the 1x1 convolutions as libgqhip GEMMs and -- switch off -- as library convolutions: each against the
    propagated fp64 model; the statistics left for the next GroupNorm are those of the route's own output."""
    from pit_hip.modules import unet as U

    torch.manual_seed(7)
    blk = U.ResnetBlock(256, 128, 0.0).eval().to(DEV).to(memory_format=torch.channels_last)
    att = U.AttnBlock(512).eval().to(DEV).to(memory_format=torch.channels_last)
    U.mark_winograd(blk)
    x = _cl(torch.randn(2, 256, 16, 32, device=DEV))
    pb = torch.randn(256, device=DEV)
    xa = _cl(torch.randn(2, 512, 16, 16, device=DEV))
    tb, ta = R.twin64(blk), R.twin64(att)
    with torch.no_grad():
        for own in (True, False):
            cs = R.C_F16X3 if own else R.C_LIB_FP32
            with switch(DIRECT_CONV_1X1=own):
                for pre in (None, pb):
                    ref, bound = R.resnet_ref_and_bound(tb, R.d64(x), None if pre is None else R.d64(pre), cs=cs)
                    y = blk(x) if pre is None else blk(x, pre)
                    R.bound_gate(y, ref, bound, f"ResnetBlock 256 -> 128, 1x1 {'own' if own else 'library'}, pending bias {pre is not None}")
                    _stats_follow_output(y, "ResnetBlock 256 -> 128 statistics")
                ref, bound = R.attn_ref_and_bound(ta, R.d64(xa), c_proj=cs)
                ya = att(xa)
                R.bound_gate(ya, ref, bound, f"AttnBlock 512 at 16 x 16, 1x1 {'own' if own else 'library'}")
                if getattr(ya, "_gn_stats", None) is not None:
                    _stats_follow_output(ya, "AttnBlock statistics")


def test_attn_block_f16x3_and_fp32_attention_gemms_each_match_fp64():
    """AttnBlock at the bench's token count (32 x 32) with the attention GEMMs on the fp16 x 3 route and -- switch off -- as fp32
    library GEMMs: each against the propagated fp64 model."""
    from pit_hip.modules import unet as U

    torch.manual_seed(22)
    blk = U.AttnBlock(512).to(DEV).eval().to(memory_format=torch.channels_last)
    x = _cl(torch.randn(4, 512, 32, 32, device=DEV))
    t = R.twin64(blk)
    with torch.no_grad():
        for f16 in (True, False):
            ref, bound = R.attn_ref_and_bound(t, R.d64(x), c_gemm=R.C_F16X3 if f16 else R.C_LIB_FP32)
            with switch(ATTN_F16X3=f16):
                y = blk(x)
            R.bound_gate(y, ref, bound, f"AttnBlock 512 at 32 x 32, attention GEMMs {'fp16 x 3' if f16 else 'fp32 library'}")
            _stats_follow_output(y, "AttnBlock statistics")


def test_attention_fused_qkv_and_three_convolutions_each_match_fp64():
    from pit_hip.modules import unet as U

    torch.manual_seed(3)
    blk = U.AttnBlock(128).eval().to(DEV).to(memory_format=torch.channels_last)
    x = _cl(torch.randn(2, 128, 8, 8).to(DEV))
    t = R.twin64(blk)
    with torch.no_grad():
        ref, bound = R.attn_ref_and_bound(t, R.d64(x), c_proj=R.C_LIB_FP32, c_gemm=R.C_LIB_FP32)
        for fused in (True, False):
            with switch(FUSED_QKV=fused):
                R.bound_gate(blk(x), ref, bound, f"AttnBlock 128 at 8 x 8, {'one q|k|v GEMM' if fused else 'three convolutions'}")


def test_downsample_direct_and_library_routes_each_match_fp64():
    """Downsample (reference unet.py:90-97: zero row / column at the bottom / right, 3x3 stride 2) on libgqhip's phase-image kernel and
    -- switch off -- on the library convolution: each against fp64 with the linear-op gate."""
    from pit_hip.modules import unet as U

    torch.manual_seed(9)
    ds = U.Downsample(128).eval().to(DEV).to(memory_format=torch.channels_last)
    x = _cl(torch.randn(2, 128, 32, 64, device=DEV))
    with torch.no_grad():
        ref, mag = R.conv_ref_and_mag(F.pad(R.d64(x), (0, 1, 0, 1)), ds.conv.weight.double(), ds.conv.bias.double(), 2, 0)
        for own, c in ((True, R.C_F16X3), (False, R.C_LIB_FP32)):
            with switch(DIRECT_CONV_S2=own):
                y, pb = ds(x)
            if own:
                assert pb is None
                _stats_follow_output(y, "Downsample statistics")
            y = y if pb is None else y + pb[None, :, None, None]
            R.lin_gate(y, ref, mag, c, f"Downsample 128, {'own kernel' if own else 'library'}")


def test_conv_out_inside_a_decoder_forward_fused_and_unfused_each_match_fp64():
    """The decoder's conv_out (reference unet.py:585-587: norm_out -> swish -> conv 128 -> 3) as it is called inside a real forward
    (statistics left by the last ResnetBlock, a pending bias or none): the call's own input is CAPTURED and the fp64 reference is
    formed from it, for the fused kernel and for the unfused route -- whatever the upstream layers did is not part of the comparison."""
    from pit_hip.modules import unet as U

    torch.manual_seed(2)
    dec = U.Decoder(**TOY).eval().to(DEV).to(memory_format=torch.channels_last)
    z = _cl(torch.randn(2, 16, 8, 8, device=DEV))
    real = U._norm_act_conv_small
    seen = []

    def capturing(norm, conv, x, pre_bias=None):
        y = real(norm, conv, x, pre_bias)
        seen.append((x.detach().clone(), None if pre_bias is None else pre_bias.detach().clone(), y.detach().clone()))
        return y

    U._norm_act_conv_small = capturing
    try:
        with torch.no_grad():
            for fused in (True, False):
                with switch(FUSED_CONV_OUT=fused):
                    out = dec(z)
                x, pb, y = seen.pop()
                assert torch.equal(out, y) and not seen
                xin = R.d64(x) if pb is None else R.d64(x) + R.d64(pb)[None, :, None, None]
                norm, conv = R.twin64(dec.norm_out), R.twin64(dec.conv_out)
                a = F.silu(norm(xin))
                ref, mag = R.conv_ref_and_mag(a, conv.weight, conv.bias)
                bound = (R.C_FP32 if fused else R.C_LIB_FP32) * mag + F.conv2d(R.gn_own_error(norm, xin, True), conv.weight.abs(), None, 1, 1)
                R.bound_gate(y, ref, R.SECOND_ORDER * bound, f"decoder conv_out, {'fused kernel' if fused else 'norm pass + library convolution'}")
    finally:
        U._norm_act_conv_small = real


# ------------------------------------------------------------------------------------------ Winograd pieces
def test_winograd_conv3x3_matches_fp64():
    """Winograd F(2x2,3x3) / F(4x4,3x3) (transform kernels + the fp32 library GEMMs) against an fp64 convolution, linear-op gate."""
    from pit_hip import _lib
    from pit_hip.modules import unet as U

    torch.manual_seed(2)
    for cin, cout, H, W in ((256, 256, 16, 24), (512, 256, 8, 8), (64, 32, 6, 10), (128, 128, 32, 32)):
        conv = torch.nn.Conv2d(cin, cout, 3, 1, 1).to(DEV).to(memory_format=torch.channels_last)
        x = _cl(torch.randn(2, cin, H, W).to(DEV))
        with torch.no_grad():
            ref, mag = R.conv_ref_and_mag(R.d64(x), conv.weight.double())
            for f4 in (False, True):
                if f4 and (H % 4 or W % 4):
                    continue
                y = _lib.wino_conv3x3(x, U._wino_weights(conv, f4))
                assert y.is_contiguous(memory_format=torch.channels_last)
                R.lin_gate(y, ref, mag, R.C_WINO_F4 if f4 else R.C_WINO_F2, f"Winograd F({4 if f4 else 2},3) {cin}->{cout} {H}x{W}, fp32 GEMMs")


def test_winograd_with_fused_groupnorm_and_unfused_each_match_fp64():
    """GroupNorm + swish applied inside the Winograd input transform, and gn_silu followed by the plain transform: each against fp64
    (GroupNorm's own fp32 error propagated through |w|)."""
    from pit_hip import _lib
    from pit_hip.modules import unet as U

    torch.manual_seed(4)
    conv = torch.nn.Conv2d(256, 128, 3, 1, 1).to(DEV).to(memory_format=torch.channels_last)
    norm = torch.nn.GroupNorm(32, 256, eps=1e-6).to(DEV)
    with torch.no_grad():
        norm.weight.normal_(); norm.bias.normal_()
    x = _cl(torch.randn(3, 256, 12, 20).to(DEV))
    pb = torch.randn(256).to(DEV)
    n64 = R.twin64(norm)
    with torch.no_grad():
        for pre in (None, pb):
            xin = R.d64(x) if pre is None else R.d64(x) + R.d64(pre)[None, :, None, None]
            ref, mag = R.conv_ref_and_mag(F.silu(n64(xin)), conv.weight.double())
            gnerr = F.conv2d(R.gn_own_error(n64, xin, True), conv.weight.double().abs(), None, 1, 1)
            for f4 in (False, True):
                Uw = U._wino_weights(conv, f4)
                bound = R.SECOND_ORDER * ((R.C_WINO_F4 if f4 else R.C_WINO_F2) * mag + gnerr)
                stats = _lib.gn_stats(x, 32, pre)
                fused = _lib.wino_conv3x3(x, Uw, gn=(norm.weight, norm.bias, 32, 1e-6, True, stats, pre))
                plain = _lib.wino_conv3x3(_lib.gn_silu(x, norm.weight, norm.bias, 32, 1e-6, silu=True, pre_bias=pre), Uw)
                tag = f"F({4 if f4 else 2},3), pending bias {pre is not None}"
                R.bound_gate(fused, ref, bound, "GroupNorm inside the input transform, " + tag)
                R.bound_gate(plain, ref, bound, "gn_silu + plain transform, " + tag)


def test_winograd_fused_tail_and_unfused_each_match_fp64():
    """Output transform + bias + residual + GroupNorm statistics in one pass, and plain transform followed by add_bias_stats: each
    against fp64; the statistics are those of the route's own output."""
    from pit_hip import _lib
    from pit_hip.modules import unet as U

    torch.manual_seed(6)
    for cin, cout, H, W in ((256, 128, 16, 24), (128, 512, 8, 12)):
        conv = torch.nn.Conv2d(cin, cout, 3, 1, 1).to(DEV).to(memory_format=torch.channels_last)
        x = _cl(torch.randn(3, cin, H, W).to(DEV))
        res = _cl(torch.randn(3, cout, H, W).to(DEV))
        bias = torch.randn(cout).to(DEV)
        with torch.no_grad():
            ref, mag = R.conv_ref_and_mag(R.d64(x), conv.weight.double(), R.d64(bias), 1, 1, R.d64(res))
            for f4 in (False, True):
                Uw = U._wino_weights(conv, f4)
                c = R.C_WINO_F4 if f4 else R.C_WINO_F2
                y, stats = _lib.wino_conv3x3(x, Uw, residual=res, bias=bias, stats_groups=32)
                y0, stats0 = _lib.add_bias_stats(res, _lib.wino_conv3x3(x, Uw), bias, 32)
                for name, yy, st in (("fused tail", y, stats), ("plain + add_bias_stats", y0, stats0)):
                    R.lin_gate(yy, ref, mag, c, f"Winograd F({4 if f4 else 2},3) {cin}->{cout}, {name}")
                    yy._gn_stats = (st, 32)
                    _stats_follow_output(yy, name)
