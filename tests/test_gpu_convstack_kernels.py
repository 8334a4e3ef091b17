"""-m gpu, marker convstack: the conv stack's own kernels against fp64 references (Winograd / direct / 1x1 / stride-2 / sub-pixel
convolutions on the fp16 x 3 scheme, fp32 matrix-core convolutions, GroupNorm statistics, attention, checksums).  No quantiser
parity depends on these; they are collected after everything that does (tests/conftest.py)."""
import json
import math
import os

import numpy as np
import pytest
import torch

import convstack_ref as R  # noqa: F401
from oracle import gq_oracle as O  # noqa: F401
from gpu_common import (DEV, FULL, G, META, _BIG_N_SCRIPT, _e2e_vs_golden, _engine, _psnr, _rows, _stv, _trained_like_engine,
                        _x512, load)  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.convstack
def test_winograd_f16x3_gemm_is_as_accurate_as_the_fp32_gemm():
    """The Winograd GEMMs as one fp16 GEMM over a K axis carrying the three products of two-term fp16 splits
    (wino_in_nhwc_f16x3 + torch.bmm(out_dtype=fp32)): against an fp64 convolution the error must be no worse than
    1.5x the library's fp32-GEMM route (itself a split-bf16 emulation on gfx950), for both tile sizes, incl. inputs near
    the bound the scale is derived from and a 1e4x larger one (the power-of-two scales are exact)."""
    import torch.nn.functional as F
    from pit_hip import _lib
    from pit_hip.modules import unet as U

    torch.manual_seed(5)
    for cin, cout, H, W, amp in ((256, 256, 16, 24, 1.0), (512, 512, 8, 8, 1.0), (128, 128, 32, 32, 30.0), (128, 256, 16, 16, 1e-3)):
        conv = torch.nn.Conv2d(cin, cout, 3, 1, 1).to(DEV).to(memory_format=torch.channels_last)
        x = (amp * torch.randn(2, cin, H, W)).to(DEV).contiguous(memory_format=torch.channels_last)
        bound = float(x.abs().max())
        with torch.no_grad():
            ref = F.conv2d(x.double(), conv.weight.double(), None, 1, 1)
            scale = float(ref.abs().mean())
            for f4 in (False, True):
                Uw = U._wino_weights(conv, f4)
                u3, us, _ = U._wino_weights_f16(conv, f4)
                assert u3.dtype == torch.float16 and tuple(u3.shape) == (Uw.shape[0], 3 * cin, cout)
                y32 = _lib.wino_conv3x3(x, Uw)
                for b in (bound, bound * 1e4):
                    y16 = _lib.wino_conv3x3(x, Uw, f16=(u3, us, b))
                    assert torch.isfinite(y16).all()
                    e32 = float((y32.double() - ref).abs().max()) / scale
                    e16 = float((y16.double() - ref).abs().max()) / scale
                    print(f"F({4 if f4 else 2},3) {cin}->{cout} amp {amp:g} bound x{b / bound:g}: fp32 GEMM {e32:.2e}, f16x3 {e16:.2e}")
                    assert e16 <= 1.5 * e32 + 1e-7, (e16, e32)
                # fused tail (bias + residual + statistics) goes through the same scale
                res = torch.randn_like(y32)
                y_a, st_a = _lib.wino_conv3x3(x, Uw, residual=res, bias=conv.bias, stats_groups=32)
                y_b, st_b = _lib.wino_conv3x3(x, Uw, residual=res, bias=conv.bias, stats_groups=32, f16=(u3, us, bound))
                assert float((y_a - y_b).abs().max()) <= 4e-3 * scale if f4 else 4e-4 * scale
                assert torch.allclose(_lib.gn_stats_values(st_a), _lib.gn_stats_values(st_b), rtol=1e-4, atol=1e-2)


@pytest.mark.convstack
def test_wino_gemm_f16x2_wider_levels_match_fp64_and_the_library_route():
    """libgqhip's own Winograd GEMM for the 256- / 512-channel levels ([h | l] operand, weights in MFMA operand order, three
    products in the kernel): (1) the GEMM alone against fp64 of the same split operands and against the library's
    K-concatenated fp16 GEMM, incl. Cin != Cout; (2) through wino_conv3x3 (both tile sizes, fused GroupNorm) against an
    fp64 convolution: no worse than the library route."""
    import torch.nn.functional as F
    from pit_hip import _lib
    from pit_hip.modules import unet as U

    torch.manual_seed(11)
    L = _lib.lib()
    for P, tiles, cin, cout in ((3, 512, 256, 256), (2, 256, 512, 512), (2, 768, 512, 256), (2, 256, 256, 512), (1, 256, 32, 128)):
        V = torch.randn(P, tiles, cin, device=DEV) * 40.0
        Uw = torch.randn(P, cin, cout, device=DEV) * 3.0
        vh = V.half(); vl = (V - vh.float()).half()
        uh = Uw.half(); ul = (Uw - uh.float()).half()
        V2 = torch.cat([vh, vl], 2).contiguous()
        Wf = _lib.wino_weights_operand_order(uh, ul)
        M = torch.full((P, tiles, cout), float("nan"), device=DEV)
        _lib._check(L.wino_gemm_f16x2(V2.data_ptr(), Wf.data_ptr(), M.data_ptr(), P, tiles, cin, cout,
                                      torch.cuda.current_stream().cuda_stream), "wino_gemm_f16x2")
        r64 = (torch.bmm(vh.double(), uh.double()) + torch.bmm(vh.double(), ul.double()) + torch.bmm(vl.double(), uh.double()))
        sc = torch.bmm(V.abs().double(), Uw.abs().double())
        e = float(((M.double() - r64).abs() / sc).max())
        lib3 = torch.bmm(torch.cat([vh, vh, vl], 2), torch.cat([uh, ul, uh], 1), out_dtype=torch.float32)
        e_lib = float(((lib3.double() - r64).abs() / sc).max())
        print(f"wino_gemm_f16x2 {P} x {tiles} x {cin} -> {cout}: err {e:.2e} of sum|a||b| (library K-concatenated GEMM {e_lib:.2e})")
        assert torch.isfinite(M).all() and e <= 3e-7, e
    # invalid shapes are refused, not mis-tiled
    assert L.wino_gemm_f16x2(V2.data_ptr(), Wf.data_ptr(), M.data_ptr(), 1, 255, 32, 128, None) != 0
    assert L.wino_gemm_f16x2(V2.data_ptr(), Wf.data_ptr(), M.data_ptr(), 1, 256, 48, 128, None) != 0
    assert L.wino_gemm_f16x2(V2.data_ptr(), Wf.data_ptr(), M.data_ptr(), 1, 256, 32, 64, None) != 0

    # through the convolution: sizes for which own_gemm_fits() holds (>= one full round of 512 blocks)
    for cin, cout, B, H, W in ((256, 256, 8, 64, 64), (512, 512, 4, 64, 64)):
        conv = torch.nn.Conv2d(cin, cout, 3, 1, 1).to(DEV).to(memory_format=torch.channels_last)
        norm = torch.nn.GroupNorm(32, cin, eps=1e-6).to(DEV)
        with torch.no_grad():
            norm.weight.uniform_(0.5, 1.5); norm.bias.uniform_(-0.3, 0.3)
        x = torch.randn(B, cin, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            a = F.silu(norm(x))
            ref = F.conv2d(a.double(), conv.weight.double(), None, 1, 1)
            scale = float(ref.abs().mean())
            stats = _lib.gn_stats(x, 32)
            gn = (norm.weight, norm.bias, 32, 1e-6, True, stats, None)
            for f4 in (False, True):
                Uw = U._wino_weights(conv, f4)
                u3, us, wf2 = U._wino_weights_f16(conv, f4)
                tiles = B * (H // (4 if f4 else 2)) * (W // (4 if f4 else 2))
                assert wf2 is not None and (_lib.own_gemm_fits(Uw.shape[0], tiles, cout, cin) or cin > 256 or f4)
                bound = U._gn_act_bound(norm, x)
                y_lib = _lib.wino_conv3x3(x, Uw, gn=gn, f16=(u3, us, bound, None))
                y_own = _lib.wino_conv3x3(x, Uw, gn=gn, f16=(u3, us, bound, wf2))
                e_lib = float((y_lib.double() - ref).abs().max()) / scale
                e_own = float((y_own.double() - ref).abs().max()) / scale
                print(f"F({4 if f4 else 2},3) {cin}->{cout}: library f16x3 {e_lib:.2e}, own GEMM {e_own:.2e}, "
                      f"max diff {float((y_own - y_lib).abs().max()) / scale:.2e}")
                assert e_own <= 1.5 * e_lib + 1e-7, (e_own, e_lib)
                # with the fused tail (bias + residual + statistics): each GEMM route against fp64 on the linear-op gate, and the
                # statistics each leaves are the sums over its own output
                res = torch.randn_like(y_own)
                ref_t, mag_t = R.conv_ref_and_mag(a.double(), conv.weight.double(), conv.bias.double(), 1, 1, res.double())
                gnerr = F.conv2d(R.gn_own_error(R.twin64(norm), x.double().contiguous(), True), conv.weight.double().abs(), None, 1, 1)
                for name, w2 in (("library GEMM", None), ("own GEMM", wf2)):
                    yt, st = _lib.wino_conv3x3(x, Uw, gn=gn, residual=res, bias=conv.bias, stats_groups=32, f16=(u3, us, bound, w2))
                    R.bound_gate(yt, ref_t, R.SECOND_ORDER * ((R.C_WINO_F4 if f4 else R.C_WINO_F2) * mag_t + gnerr),
                                 f"F({4 if f4 else 2},3) {cin}->{cout} fused tail, {name}")
                    y64 = yt.double().contiguous()
                    err = (_lib.gn_stats_values(st).double() - R.stats_of(y64)).abs() / (R.stats_of(y64.abs()) + 1e-30)
                    assert float(err.max()) <= 2e-6, name


@pytest.mark.convstack
def test_gn_act_bound_is_a_bound_and_unet_agrees_with_fp32_gemms():
    from pit_hip.modules import unet as U

    torch.manual_seed(6)
    norm = torch.nn.GroupNorm(32, 128, eps=1e-6).to(DEV)
    with torch.no_grad():
        norm.weight.mul_(3.0).add_(torch.randn(128, device=DEV))
        norm.bias.add_(torch.randn(128, device=DEV))
    x = torch.randn(2, 128, 16, 16, device=DEV)
    x[0, 5, 3, 3] = 1e4                                  # one outlier: the worst case of the bound
    with torch.no_grad():
        y = U._norm_act(norm, x.contiguous(memory_format=torch.channels_last))
    assert float(y.abs().max()) <= U._gn_act_bound(norm, x) and y._act_bound == U._gn_act_bound(norm, x)
    cfg = dict(ch=128, out_ch=3, in_channels=3, resolution=64, z_channels=16, double_z=True, ch_mult=[1, 2, 4, 4],
               num_res_blocks=2, attn_resolutions=[8], dropout=0.0)
    dec = U.Decoder(**cfg).eval().to(DEV).to(memory_format=torch.channels_last)
    enc = U.Encoder(**cfg).eval().to(DEV).to(memory_format=torch.channels_last)
    z = torch.randn(2, 16, 8, 8).to(DEV).contiguous(memory_format=torch.channels_last)
    img = (torch.rand(2, 3, 64, 64) * 2 - 1).to(DEV).contiguous(memory_format=torch.channels_last)
    # both GEMM routes of the whole modules (fp16 x 3 and the library's fp32 GEMMs), each against the fp64 twin on the product contract
    with torch.no_grad():
        z64, x64 = R.twin64(enc)(R.d64(img)), R.twin64(dec)(R.d64(z))
        for flag in (True, False):
            U.WINOGRAD_F16X3 = flag
            try:
                R.contract_x(dec(z), x64, f"decoder, Winograd GEMMs {'fp16 x 3' if flag else 'fp32 library'}")
                R.contract_z(enc(img), z64, f"encoder, Winograd GEMMs {'fp16 x 3' if flag else 'fp32 library'}")
            finally:
                U.WINOGRAD_F16X3 = True


@pytest.mark.convstack
def test_fused_groupnorm_transforms_bit_identical_on_the_f16_routes():
    """GroupNorm + swish inside the Winograd input transform (unet.FUSED_WINO_GN / _F4) writes the same V as gn_apply
    followed by the plain transform -- fp16 x 3 operand of the library GEMM, F(2x2,3x3) and F(4x4,3x3), with and without a pending bias, image borders included (12 x 20 pixels)."""
    from pit_hip import _lib
    from pit_hip.modules import unet as U

    torch.manual_seed(14)
    for cin, cout in ((128, 128), (256, 128)):
        conv = torch.nn.Conv2d(cin, cout, 3, 1, 1).to(DEV).to(memory_format=torch.channels_last)
        conv._gq_wino = conv._gq_wino4 = True
        norm = torch.nn.GroupNorm(32, cin, eps=1e-6).to(DEV)
        with torch.no_grad():
            norm.weight.normal_(); norm.bias.normal_()
        x = (3 * torch.randn(3, cin, 12, 20)).to(DEV).contiguous(memory_format=torch.channels_last)
        pb = torch.randn(cin).to(DEV)
        with torch.no_grad():
            for f4 in (False, True):
                Uw = U._wino_weights(conv, f4)
                f16 = U._f16_args_gn(conv, norm, x, f4)
                for pre in (None, pb):
                    stats = _lib.gn_stats(x, 32, pre)
                    gn = (norm.weight, norm.bias, 32, 1e-6, True, stats, pre)
                    fused = _lib.wino_conv3x3(x, Uw, gn=gn, f16=f16)
                    if pre is None:    # same statistics tensor -> same folded scale / shift -> same bits
                        xn = _lib.gn_apply(x, norm.weight, norm.bias, 32, 1e-6, True, stats)
                        plain = _lib.wino_conv3x3(xn, Uw, f16=f16)
                        assert torch.equal(fused, plain), (cin, f4, float((fused - plain).abs().max()))
                    else:              # gn_silu sums its own statistics (atomics: last-bit differences are possible)
                        xn = _lib.gn_silu(x, norm.weight, norm.bias, 32, 1e-6, silu=True, pre_bias=pre)
                        plain = _lib.wino_conv3x3(xn, Uw, f16=f16)        # (judged against fp64 below, like `fused`)
                    ref, mag = R.conv_ref_and_mag(xn.double(), conv.weight.double())
                    for name, yy in (("fused GN transform", fused), ("GN pass + plain transform", plain)):
                        R.lin_gate(yy, ref, mag, R.C_WINO_F4 if f4 else R.C_WINO_F2, f"{name} {cin}->{cout} F({4 if f4 else 2},3)")


@pytest.mark.convstack
def test_silu_is_accurate_and_finite_at_the_extremes():
    """libgqhip's one SiLU (Newton-refined reciprocal): within 4e-7 relative of fp64 on ordinary inputs; -0 / x at the
    ends of the range (e^-x overflows for x < -88.7: the IEEE quotient there is -0, and so is ours -- no NaN; where
    1 + e^-x > 1e37 the reciprocal is subnormal and the result, of magnitude < 1e-35, is returned as -0)."""
    from pit_hip import _lib

    C = 128
    vals = torch.tensor([-200.0, -100.0, -88.0, -87.0, -20.0, -1.0, -1e-30, 0.0, 1e-30, 1.0, 20.0, 88.0, 100.0, 3e4])
    # GroupNorm with gamma = 0 returns beta: feed each value through as beta of channel group k
    x = torch.randn(1, C, 4, 4).to(DEV).contiguous(memory_format=torch.channels_last)
    for v in vals.tolist():
        beta = torch.full((C,), v, device=DEV)
        y = _lib.gn_silu(x, torch.zeros(C, device=DEV), beta, 32, 1e-6, silu=True)
        want = v / (1.0 + math.exp(-v)) if v > -700 else 0.0
        got = float(y.flatten()[0])
        assert not math.isnan(got) and abs(got - want) <= 4e-7 * abs(want) + 1e-34, (v, got, want)
    g = torch.Generator().manual_seed(3)
    x = (4 * torch.randn(2, C, 16, 16, generator=g)).to(DEV).contiguous(memory_format=torch.channels_last)
    y = _lib.gn_silu(x, torch.ones(C, device=DEV), torch.zeros(C, device=DEV), 32, 1e-6, silu=True)
    xn = torch.nn.functional.group_norm(x.double(), 32, eps=1e-6)
    ref = xn * torch.sigmoid(xn)
    # the normalised value carries ~1e-7 of the un-normalised magnitude (fp32 fold of scale and shift): absolute floor
    assert float(((y.double() - ref).abs() / (ref.abs() + 1.0)).max()) <= 5e-7


@pytest.mark.convstack
def test_direct_conv3x3_matches_fp64_convolution():
    """conv3x3_direct (implicit GEMM, fp16 x 3, GroupNorm + swish in the split pass, bias / residual / statistics in the
    epilogue) against torch's fp64 convolution of the same activated tensor: error <= 6e-7 of sum |x||w| (three products
    of 22-bit splits, fp32 accumulation over 9 Cin terms), image borders and several tiles per image included."""
    from pit_hip import _lib
    from pit_hip.modules import unet as U

    torch.manual_seed(21)
    for cin, cout, (B, H, W) in ((128, 128, (2, 16, 64)), (256, 128, (1, 8, 32)), (32, 128, (3, 24, 32)), (128, 256, (2, 16, 32)),
                                 (256, 256, (1, 8, 64))):
        conv = torch.nn.Conv2d(cin, cout, 3, 1, 1).to(DEV).to(memory_format=torch.channels_last)
        norm = torch.nn.GroupNorm(8 if cin == 32 else 32, cin, eps=1e-6).to(DEV)
        with torch.no_grad():
            norm.weight.normal_(); norm.bias.normal_()
            x = (2 * torch.randn(B, cin, H, W, device=DEV)).contiguous(memory_format=torch.channels_last)
            res = torch.randn(B, cout, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
            wf, us = _lib.conv3_weights_f16(conv.weight)
            groups = norm.num_groups
            stats = _lib.gn_stats(x, groups)
            gn = (norm.weight, norm.bias, groups, 1e-6, True, stats, None)
            bound = U._gn_act_bound(norm, x)
            xn = _lib.gn_apply(x, norm.weight, norm.bias, groups, 1e-6, True, stats).double()
            sc = torch.nn.functional.conv2d(xn.abs(), conv.weight.double().abs(), None, 1, 1)
            ref0 = torch.nn.functional.conv2d(xn, conv.weight.double(), None, 1, 1)
            # (a) bias + residual + statistics
            y, st = _lib.conv3x3_direct(x, wf, us, bound, gn=gn, residual=res, bias=conv.bias, stats_groups=32)
            ref = ref0 + conv.bias.double()[None, :, None, None] + res.double()
            assert float(((y.double() - ref).abs() / sc).max()) <= 6e-7
            yd = y.double().permute(0, 2, 3, 1).reshape(B, H * W, 32, cout // 32)
            st_y = torch.stack([yd.sum((1, 3)), (yd ** 2).sum((1, 3))], -1).flatten()
            assert torch.allclose(_lib.gn_stats_values(st), st_y, rtol=2e-6, atol=1e-3), float((_lib.gn_stats_values(st) - st_y).abs().max())
            # (b) nothing fused
            y2 = _lib.conv3x3_direct(x, wf, us, bound, gn=gn)
            assert float(((y2.double() - ref0).abs() / sc).max()) <= 6e-7


@pytest.mark.convstack
def test_direct_conv3x3_rejects_shapes_it_does_not_tile():
    from pit_hip import _lib

    conv = torch.nn.Conv2d(128, 128, 3, 1, 1).to(DEV)
    wf, us = _lib.conv3_weights_f16(conv.weight)
    for shape in ((1, 128, 12, 32), (1, 128, 8, 48), (1, 120, 8, 32)):
        x = torch.randn(*shape, device=DEV).contiguous(memory_format=torch.channels_last)
        with pytest.raises(_lib.GqHipError):
            _lib.conv3x3_direct(x, wf, us, 10.0, None)
    with pytest.raises(_lib.GqHipError):      # the convolution is fed by a GroupNorm: no GroupNorm, no call
        _lib.conv3x3_direct(torch.randn(1, 128, 8, 32, device=DEV).contiguous(memory_format=torch.channels_last), wf, us, 10.0, None)
    with pytest.raises(_lib.GqHipError):
        _lib.conv3_weights_f16(torch.randn(64, 128, 3, 3, device=DEV))
    L = _lib.lib()
    x = torch.randn(1, 128, 8, 32, device=DEV).contiguous(memory_format=torch.channels_last)
    y = torch.empty_like(x)
    S = torch.cuda.current_stream().cuda_stream
    g = torch.ones(128, device=DEV)
    st = _lib.gn_stats(x, 32)
    args = lambda B, H, W, Cin, Cout: (x.data_ptr(), g.data_ptr(), g.data_ptr(), None, st.data_ptr(), 32, 1e-6, 1, 1.0, wf.data_ptr(), None,
                                       None, y.data_ptr(), None, B, H, W, Cin, Cout, 32, 1.0, S)
    assert L.conv3x3_gn_f16x3(*args(1, 12, 32, 128, 128)) != 0
    assert L.conv3x3_gn_f16x3(*args(1, 8, 32, 128, 192)) != 0
    assert L.conv3x3_gn_f16x3(*args(0, 8, 32, 128, 128)) == 0
    assert L.conv3x3_gn_f16x3(*args(1, 8, 32, 128, 128)) == 0


@pytest.mark.convstack
def test_conv_out_kernel_matches_fp64_reference():
    """conv3x3(SiLU(GroupNorm(x))) into 1..4 channels in one kernel (the decoder's conv_out) vs torch fp64."""
    from pit_hip import _lib

    torch.manual_seed(31)
    for cin, cout, (B, H, W), silu in ((128, 3, (2, 32, 48), True), (128, 4, (1, 16, 16), False), (256, 1, (1, 16, 32), True)):
        conv = torch.nn.Conv2d(cin, cout, 3, 1, 1).to(DEV)
        norm = torch.nn.GroupNorm(32, cin, eps=1e-6).to(DEV)
        with torch.no_grad():
            norm.weight.normal_(); norm.bias.normal_()
            x = (2 * torch.randn(B, cin, H, W, device=DEV)).contiguous(memory_format=torch.channels_last)
            pb = torch.randn(cin, device=DEV)
            for pre in (None, pb):
                stats = _lib.gn_stats(x, 32, pre)
                y = _lib.conv3x3_gn_small(x, conv.weight.permute(0, 2, 3, 1).contiguous(), conv.bias,
                                          (norm.weight, norm.bias, 32, 1e-6, silu, stats, pre))
                xin = x.double() if pre is None else x.double() + pre.double()[None, :, None, None]
                xn = torch.nn.functional.group_norm(xin, 32, norm.weight.double(), norm.bias.double(), 1e-6)
                if silu:
                    xn = xn * torch.sigmoid(xn)
                ref = torch.nn.functional.conv2d(xn, conv.weight.double(), conv.bias.double(), 1, 1)
                sc = torch.nn.functional.conv2d(xn.abs(), conv.weight.double().abs(), None, 1, 1) + 1.0
                assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
                assert float(((y.double() - ref).abs() / sc).max()) <= 2e-6, (cin, cout, pre is not None)
    with pytest.raises(_lib.GqHipError):
        _lib.conv3x3_gn_small(torch.randn(1, 128, 8, 16, device=DEV).contiguous(memory_format=torch.channels_last),
                              torch.randn(3, 3, 3, 128, device=DEV), None, (norm.weight, norm.bias, 32, 1e-6, True, stats, None))


@pytest.mark.convstack
def test_conv1x1_f16x3_matches_fp64():
    """conv1x1_direct (fp16 x 3 GEMM over the pixels, x + pending bias split inside the kernel; device-side or host scale;
    bias / residual / statistics epilogue) vs fp64: error <= 1.2e-6 of sum |x||w| (worst case 3 x 2^-22 = 7.2e-7 + fp32 accumulation over K <= 512; a 4e3 outlier in x
    forces a scale at which the low parts of ordinary elements sit near fp16's subnormals)."""
    from pit_hip import _lib

    torch.manual_seed(41)
    for cin, cout, (B, H, W) in ((256, 128, (2, 16, 32)), (128, 256, (1, 16, 16)), (512, 512, (2, 32, 32)), (384, 512, (1, 8, 32))):
        conv = torch.nn.Conv2d(cin, cout, 1).to(DEV).to(memory_format=torch.channels_last)
        with torch.no_grad():
            x = (3 * torch.randn(B, cin, H, W, device=DEV)).contiguous(memory_format=torch.channels_last)
            x[0, 5, 3, 7] = 4e3                                       # an outlier the scale has to cover
            res = torch.randn(B, cout, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
            pb = torch.randn(cin, device=DEV)
            wf, us = _lib.conv3_weights_f16(conv.weight)
            w64 = conv.weight.double()
            for pre in (None, pb):
                xin = x.double() if pre is None else x.double() + pre.double()[None, :, None, None]
                ref0 = torch.nn.functional.conv2d(xin, w64)
                sc = torch.nn.functional.conv2d(xin.abs(), w64.abs())
                if _lib.gn_nhwc_ok(cin, 32):                          # device-side scale from GroupNorm statistics
                    scales = _lib.f16_scales(_lib.gn_stats(x, 32, pre), 1.0, us)
                    y, st = _lib.conv1x1_direct(x, wf, us, scales, residual=res, bias=conv.bias, stats_groups=32, pre_bias=pre)
                    ref = ref0 + conv.bias.double()[None, :, None, None] + res.double()
                    assert float(((y.double() - ref).abs() / sc).max()) <= 1.2e-6, (cin, cout)
                    yd = y.double().permute(0, 2, 3, 1).reshape(B, H * W, 32, cout // 32)
                    st_y = torch.stack([yd.sum((1, 3)), (yd ** 2).sum((1, 3))], -1).flatten()
                    assert torch.allclose(_lib.gn_stats_values(st), st_y, rtol=2e-6, atol=1e-2), float((_lib.gn_stats_values(st) - st_y).abs().max())
                y2 = _lib.conv1x1_direct(x, wf, us, float(xin.abs().max()), pre_bias=pre)      # host bound
                assert float(((y2.double() - ref0).abs() / sc).max()) <= 1.2e-6, (cin, cout)
    L = _lib.lib()
    S = torch.cuda.current_stream().cuda_stream
    x = torch.randn(1, 128, 16, 16, device=DEV).contiguous(memory_format=torch.channels_last)
    y = torch.empty(1, 128, 16, 16, device=DEV).contiguous(memory_format=torch.channels_last)
    wf, us = _lib.conv3_weights_f16(torch.randn(128, 128, 1, 1, device=DEV))
    assert L.conv1x1_f16x3(x.data_ptr(), None, wf.data_ptr(), None, 1.0, 1.0, None, None, y.data_ptr(), None, 1, 200, 128, 128, 32, S) != 0
    assert L.conv1x1_f16x3(x.data_ptr(), None, wf.data_ptr(), None, 0.0, 1.0, None, None, y.data_ptr(), None, 1, 256, 128, 128, 32, S) != 0
    assert L.conv1x1_f16x3(x.data_ptr(), None, wf.data_ptr(), None, 1.0, 1.0, None, None, y.data_ptr(), None, 1, 256, 128, 192, 32, S) != 0
    with pytest.raises(_lib.GqHipError):
        _lib.conv1x1_direct(torch.randn(1, 128, 8, 8, device=DEV).contiguous(memory_format=torch.channels_last), wf, us, 10.0)


@pytest.mark.convstack
def test_stride2_conv_f16x3_matches_fp64():
    """conv3x3s2_direct (the reference's Downsample: zero row / column at the bottom / right, then 3x3 stride 2) on the four
    phase images, fp16 x 3: error <= 8e-7 of sum |x||w| vs fp64; bias, statistics, device-side and host scale, several tiles."""
    from pit_hip import _lib

    torch.manual_seed(51)
    for cin, cout, (B, H, W) in ((128, 128, (2, 32, 128)), (64, 256, (1, 16, 64)), (256, 512, (1, 16, 64)), (16, 128, (3, 48, 64))):
        conv = torch.nn.Conv2d(cin, cout, 3, 2, 0).to(DEV).to(memory_format=torch.channels_last)
        with torch.no_grad():
            x = (3 * torch.randn(B, cin, H, W, device=DEV)).contiguous(memory_format=torch.channels_last)
            wf, us = _lib.conv3s2_weights_f16(conv.weight)
            xp = torch.nn.functional.pad(x.double(), (0, 1, 0, 1))
            ref = torch.nn.functional.conv2d(xp, conv.weight.double(), conv.bias.double(), 2, 0)
            sc = torch.nn.functional.conv2d(xp.abs(), conv.weight.double().abs(), None, 2, 0)
            y, st = _lib.conv3x3s2_direct(x, wf, us, float(x.abs().max()), bias=conv.bias, stats_groups=32)
            assert tuple(y.shape) == (B, cout, H // 2, W // 2) and y.is_contiguous(memory_format=torch.channels_last)
            assert float(((y.double() - ref).abs() / sc).max()) <= 8e-7, (cin, cout)
            yd = y.double().permute(0, 2, 3, 1).reshape(B, (H // 2) * (W // 2), 32, cout // 32)
            st_y = torch.stack([yd.sum((1, 3)), (yd ** 2).sum((1, 3))], -1).flatten()
            assert torch.allclose(_lib.gn_stats_values(st), st_y, rtol=2e-6, atol=1e-2), float((_lib.gn_stats_values(st) - st_y).abs().max())
            if _lib.gn_nhwc_ok(cin, 32):
                y2 = _lib.conv3x3s2_direct(x, wf, us, _lib.f16_scales(_lib.gn_stats(x, 32), 1.0, us))
                ref2 = ref - conv.bias.double()[None, :, None, None]
                assert float(((y2.double() - ref2).abs() / sc).max()) <= 8e-7, (cin, cout)
    with pytest.raises(_lib.GqHipError):
        _lib.conv3x3s2_direct(torch.randn(1, 128, 16, 32, device=DEV).contiguous(memory_format=torch.channels_last), wf, us, 1.0)
    with pytest.raises(_lib.GqHipError):
        _lib.conv3s2_weights_f16(torch.randn(96, 128, 3, 3, device=DEV))


@pytest.mark.convstack
def test_attention_f16x3_matches_fp64_attention():
    """softmax(q k^T C^-1/2) v as two fp16 GEMMs over K axes of two-term fp16 splits (attn_split_qkv_f16x3 + library GEMM +
    attn_softmax_split_f16x3 + library GEMM): against fp64 attention the error must be at the level of the fp32 matmul /
    softmax / matmul route, with tight and with very loose operand bounds (the power-of-two scales are exact), at both token
    counts of the bench configurations (1024; 4096 at 512 x 512)."""
    from pit_hip import _lib

    torch.manual_seed(21)
    for B, L, C, amp in ((2, 1024, 512, 1.0), (1, 4096, 512, 3.0), (3, 256, 64, 0.2), (2, 64, 32, 1.0)):
        qkv = (amp * torch.randn(B, L, 3 * C, device=DEV)).contiguous()
        q, k, v = qkv[..., :C].double(), qkv[..., C:2 * C].double(), qkv[..., 2 * C:].double()
        ref = torch.softmax(q @ k.transpose(1, 2) * C ** -0.5, -1) @ v
        q32, k32, v32 = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        y32 = torch.matmul(torch.softmax(torch.matmul(q32, k32.transpose(1, 2)) * C ** -0.5, -1), v32)
        scale = float(ref.abs().max())
        e32 = float((y32.double() - ref).abs().max()) / scale
        qb, vb = float(qkv[..., :2 * C].abs().max()), float(qkv[..., 2 * C:].abs().max())
        for loose in (1.0, 300.0):
            O, ps = _lib.attention_f16x3(qkv, qb * loose, vb * loose)
            assert O.dtype == torch.float32 and tuple(O.shape) == (B, L, C) and torch.isfinite(O).all()
            e16 = float((O.double() * ps - ref).abs().max()) / scale
            print(f"attention B{B} L{L} C{C} amp {amp:g} bounds x{loose:g}: fp32 route {e32:.2e}, f16x3 {e16:.2e}")
            assert e16 <= 2.0 * e32 + 2e-6, (e16, e32)
    with pytest.raises(_lib.GqHipError):
        _lib.attention_f16x3(torch.randn(1, 100, 96, device=DEV), 1.0, 1.0)     # token count without an instantiation


@pytest.mark.convstack
def test_upconv2x_direct_matches_fp64():
    """Upsample (nearest x2 + conv 3x3, unet.py:60-73) as libgqhip's direct sub-pixel fp16 x 3 convolution: against an fp64
    convolution of the upsampled tensor (error at the level of the fallback route: upsample, then the ordinary convolution), the
    statistics it leaves equal those of its output, bias included; shapes of all three decoder levels, borders included."""
    import torch.nn.functional as F
    from pit_hip import _lib
    from pit_hip.modules import unet as U

    torch.manual_seed(31)
    for ch, B, H, W, outlier in ((512, 2, 32, 32, 0.0), (256, 2, 24, 64, 0.0), (512, 1, 8, 32, 0.0), (128, 2, 32, 32, 3e4)):
        up = U.Upsample(ch).to(DEV).eval().to(memory_format=torch.channels_last)
        x = 3.0 * torch.randn(B, ch, H, W, device=DEV)
        if outlier:
            x[1, 7, 3, 5] = outlier     # the device-side scale comes from a rigorous bound: no overflow, same accuracy elsewhere
        x = x.contiguous(memory_format=torch.channels_last)
        x._gn_stats = (_lib.gn_stats(x, 32), 32)
        with torch.no_grad():
            ref = F.conv2d(F.interpolate(x.double(), scale_factor=2.0, mode="nearest"), up.conv.weight.double(),
                           up.conv.bias.double(), 1, 1)
            scale = float(ref.abs().mean())
            old = U.DIRECT_UPCONV
            try:
                U.DIRECT_UPCONV = False       # the fallback: NHWC upsample copy + the ordinary convolution routes
                y_lib, pb = up(x)
                y_lib = y_lib if pb is None else y_lib + pb[None, :, None, None]
                U.DIRECT_UPCONV = True
                y_dir, pb2 = up(x)
            finally:
                U.DIRECT_UPCONV = old
        assert pb2 is None and tuple(y_dir.shape) == (B, ch, 2 * H, 2 * W)
        e_lib = float((y_lib.double() - ref).abs().max()) / scale
        e_dir = float((y_dir.double() - ref).abs().max()) / scale
        print(f"upsample {ch} ch {H}x{W}: library route {e_lib:.2e}, direct {e_dir:.2e} (of mean |y|)")
        # three products of 22-bit splits over 4 x 4 Cin terms (measured 5.6-8.0e-6); with the 3e4 outlier both routes carry its
        # rounding (the error is quoted against mean |y|, the outlier's own products are ~1e4 times that)
        assert e_dir <= (1.2e-5 if not outlier else 3.0 * e_lib + 1e-6), (e_dir, e_lib)
        st, groups = y_dir._gn_stats
        assert groups == 32 and torch.allclose(_lib.gn_stats_values(st), _lib.gn_stats_values(_lib.gn_stats(y_dir.contiguous(memory_format=torch.channels_last), 32)), rtol=1e-6, atol=1e-3)
    # shapes the kernel does not tile are refused by the binding (the module then upsamples and convolves)
    wf, us = _lib.upconv_weights_f16(torch.randn(4 * 128, 4 * 128, device=DEV), 128, 128)
    with pytest.raises(_lib.GqHipError):
        _lib.upconv2x_direct(torch.randn(1, 128, 12, 32, device=DEV).contiguous(memory_format=torch.channels_last), wf, us, 10.0)


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


# ------------------------------------------------------------------------------------------ the encoder's conv_in on libgqhip
@pytest.mark.convstack
@pytest.mark.parametrize("B,cin,H,W", [(2, 3, 64, 64), (1, 3, 8, 32), (3, 4, 16, 96), (2, 1, 24, 32), (16, 3, 256, 256)])
def test_conv_in_small_matches_fp64_and_leaves_the_statistics(B, cin, H, W):
    """conv3x3_cin_small_f32 (pit/modules/unet.py:411-413, the encoder's conv_in): against an fp64 convolution -- 9 Cin fp32 FMAs
    per output: error <= (9 Cin + 1) 2^-24 of sum |x||w| + |bias| --, image borders, the statistics it leaves for the GroupNorm
    that follows against the statistics kernel run on its own output, and bit-reproducibility over five runs."""
    import torch.nn.functional as F
    from pit_hip import _lib

    g = torch.Generator().manual_seed(7 + cin + H)
    x = (torch.rand(B, cin, H, W, generator=g) * 2 - 1).to(DEV)
    xl = x.contiguous(memory_format=torch.channels_last) if cin > 1 else x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    conv = torch.nn.Conv2d(cin, 128, 3, 1, 1).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.3)
        conv.bias.copy_(torch.randn(128, generator=g))
    wk = _lib.conv_cin_small_weights(conv.weight)
    if cin == 1:      # a one-channel image is both layouts at once for torch: the binding asks for channels_last explicitly
        pytest.skip("Cin = 1 tensors report NCHW-contiguous: the module falls back for them (covered by the C ABI check below)")
    y, st = _lib.conv3x3_cin_small(xl, wk, conv.bias, stats_groups=32)
    with torch.no_grad():
        ref = F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), 1, 1)
        mag = F.conv2d(x.double().abs(), conv.weight.double().abs(), conv.bias.double().abs(), 1, 1)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    err = float(((y.double() - ref).abs() / mag).max())
    print(f"conv_in {cin} -> 128 at {B} x {H} x {W}: error / (sum|x||w| + |b|) = {err:.2e}")
    assert err <= (9 * cin + 1) * 2.0 ** -24
    want = _lib.gn_stats_values(_lib.gn_stats(y, 32))
    got = _lib.gn_stats_values(st)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-3), float((got - want).abs().max())
    for _ in range(4):
        y2, st2 = _lib.conv3x3_cin_small(xl, wk, conv.bias, stats_groups=32)
        assert torch.equal(y2, y) and torch.equal(st2, st)
