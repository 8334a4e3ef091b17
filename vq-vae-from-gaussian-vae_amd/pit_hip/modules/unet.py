"""SD3-style conv encoder / decoder as ``torch.nn`` modules over libgqhip's conv-stack kernels.

Mirror of the reference's ``pit/modules/unet.py`` interface for the hot path:
``Encoder(**params)(x) -> [B, 2*z, H/8, W/8]`` (unet.py:317-436) and
``Decoder(**params)(z) -> [B, 3, H, W]`` (unet.py:439-587).  The module tree is
laid out so that ``state_dict()`` keys equal the reference's (``conv_in``,
``down.{i}.block.{j}.norm1`` ... ``up.{i}.upsample.conv``), so a reference
checkpoint's ``encoder.*`` / ``decoder.*`` tensors load unchanged.

Two execution paths.  A module left in NCHW runs ATen / MIOpen convolutions with libgqhip's fused GroupNorm and residual
kernels (the drop-in default).  A module converted with ``.to(memory_format=torch.channels_last)`` -- the bench
configuration -- runs every convolution but ``conv_in`` in libgqhip (direct fp16 x 3 implicit GEMMs, Winograd transforms
around fp16 GEMMs, fp32 matrix-core convolutions for the layers next to z) and is bit-reproducible run to run; which kernel
serves which layer: profiles/r03/route_table.txt.  Training / autograd / CPU tensors always take the ATen ops.

Only what the shipped SD3-UNet configs use is built: ``attn_type`` "vanilla"
(single-head SDPA, unet.py:166-206) or "none"; ``temb_channels`` is always 0 on
this path (unet.py:342, :473) so the time-embedding projection is not created.
"""
from __future__ import annotations

import math
import os

from typing import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


def _gn(ch: int) -> nn.GroupNorm:
    # unet.py:54-57
    return nn.GroupNorm(32, ch, eps=1e-6, affine=True)


def _silu(x: torch.Tensor) -> torch.Tensor:
    # unet.py:49-51 (x * sigmoid(x)); F.silu is the same function, one kernel.
    return F.silu(x)


def _use_fused(x: torch.Tensor, norm: nn.GroupNorm) -> bool:
    """HIP fused GroupNorm(+SiLU) applies to inference on HIP devices, NCHW fp32, HW % 4 == 0;
    training / autograd / CPU tensors stay on the ATen ops (this stack is PyTorch by design)."""
    if not (FUSED_GN and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()
            and x.dim() == 4 and norm.affine):
        return False
    from .. import _lib

    layout = _lib.image_layout(x)
    if layout == 0:
        return (x.shape[2] * x.shape[3]) % 4 == 0
    return layout == 1 and _lib.gn_nhwc_ok(x.shape[1], norm.num_groups)


def _materialize(x: torch.Tensor, pre_bias) -> torch.Tensor:
    return x if pre_bias is None else x + pre_bias[None, :, None, None]


def _norm_act(norm: nn.GroupNorm, x: torch.Tensor, act: bool = True, pre_bias=None) -> torch.Tensor:
    """GroupNorm followed by swish (unet.py:140-141, :146-147, :432-433) -- one fused HIP pass pair.
    ``pre_bias``: per-channel bias still pending on ``x`` (see ``_conv``)."""
    if _use_fused(x, norm):
        from .. import _lib

        st = getattr(x, "_gn_stats", None)   # left behind by the residual add that produced x (see _add)
        if st is not None and pre_bias is None and st[1] == norm.num_groups and _lib.image_layout(x) == 1:
            y = _lib.gn_apply(x, norm.weight, norm.bias, norm.num_groups, norm.eps, act, st[0])
        else:
            y = _lib.gn_silu(x, norm.weight, norm.bias, norm.num_groups, norm.eps, silu=act, pre_bias=pre_bias)
        if WINOGRAD_F16X3:
            y._act_bound = _gn_act_bound(norm, x)   # lets the Winograd GEMMs that consume y run as fp16 x 3 (see _f16_args)
        return y
    y = norm(_materialize(x, pre_bias))
    return _silu(y) if act else y


def _defer_ok(x: torch.Tensor, conv: nn.Conv2d) -> bool:
    """Deferred-bias path: inference on HIP, NCHW fp32, zero padding, output HW % 4 == 0."""
    return (FUSED_GN and DEFER_BIAS and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()
            and conv.bias is not None and conv.padding_mode == "zeros")


def _f16_gemm_ok(x: torch.Tensor) -> bool:
    """The library-GEMM half of the fp16 x 3 routes needs bmm(out_dtype=fp32) (recent PyTorch-ROCm): probed once."""
    from .. import _lib

    return _lib.bmm_out_dtype_ok(x.device)


def _f16_args(conv: nn.Conv2d, x: torch.Tensor, f4: bool):
    """(U3, u_scale, bound) for _lib.wino_conv3x3's fp16 x 3 route, or None: needs a rigorous bound on |x| (left on the
    tensor by _norm_act: every Winograd convolution of this UNet is fed by GroupNorm + swish)."""
    bound = getattr(x, "_act_bound", None)
    if not WINOGRAD_F16X3 or bound is None or not _f16_gemm_ok(x):
        return None
    u3, u_scale, wf2 = _wino_weights_f16(conv, f4)
    return u3, u_scale, bound, (wf2 if WINOGRAD_OWN_GEMM else None)


def _f16_args_gn(conv: nn.Conv2d, norm: nn.GroupNorm, x: torch.Tensor, f4: bool):
    """_f16_args for a convolution whose GroupNorm + swish runs inside the input transform: the bound is that of the
    activated tensor, which is never materialised."""
    if not WINOGRAD_F16X3 or not _f16_gemm_ok(x):
        return None
    u3, u_scale, wf2 = _wino_weights_f16(conv, f4)
    return u3, u_scale, _gn_act_bound(norm, x), (wf2 if WINOGRAD_OWN_GEMM else None)


def _conv(conv: nn.Conv2d, x: torch.Tensor, want_stats: bool = False):
    """Run ``conv`` and return (y, pending_bias).  ATen's MIOpen path adds the bias in a separate
    elementwise pass over the whole output; on the deferred path the conv runs bias-free and the
    bias is handed to the consumer (the next fused GroupNorm or residual add), which folds it in.
    ``want_stats``: the consumer is a GroupNorm(32) -- on the Winograd path its statistics come with the output."""
    if _defer_ok(x, conv):
        if _wino_ok(conv, x):
            from .. import _lib

            f4 = WINOGRAD_F4 and getattr(conv, "_gq_wino4", False) and x.shape[2] % 4 == 0 and x.shape[3] % 4 == 0
            if want_stats and FUSED_WINO_TAIL and _lib.gn_nhwc_ok(conv.out_channels, GN_GROUPS):
                # the consumer is a GroupNorm: add the bias here and leave the statistics with the output
                y, stats = _lib.wino_conv3x3(x, _wino_weights(conv, f4), bias=conv.bias, stats_groups=GN_GROUPS,
                                             f16=_f16_args(conv, x, f4))
                y._gn_stats = (stats, GN_GROUPS)
                return y, None
            return _lib.wino_conv3x3(x, _wino_weights(conv, f4), f16=_f16_args(conv, x, f4)), conv.bias
        return F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups), conv.bias
    return conv(x), None


def _match_layout(x: torch.Tensor, conv: nn.Conv2d) -> torch.Tensor:
    """A module converted with ``.to(memory_format=torch.channels_last)`` gets its input in NHWC too (the fast conv
    stack -- Winograd, sub-pixel upsample, NHWC GroupNorm kernels -- is the channels_last one)."""
    if (x.is_cuda and x.dim() == 4 and conv.weight.is_contiguous(memory_format=torch.channels_last)
            and not conv.weight.is_contiguous() and not x.is_contiguous(memory_format=torch.channels_last)):
        return x.contiguous(memory_format=torch.channels_last)
    return x


def _wino_ok(conv: nn.Conv2d, x: torch.Tensor) -> bool:
    return (WINOGRAD and getattr(conv, "_gq_wino", False) and conv.in_channels >= WINOGRAD_MIN_CH
            and conv.out_channels >= WINOGRAD_MIN_CH and conv.out_channels % 4 == 0 and x.shape[2] % 2 == 0
            and x.shape[3] % 2 == 0 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous())


def _direct_ok(conv: nn.Conv2d, x: torch.Tensor) -> bool:
    """3x3 stride-1 padding-1 convolutions of the widest levels run as libgqhip's direct fp16 x 3 convolution: into 128
    channels (the 256 x 256 level), where every Winograd route is HBM-bound on its transformed tensors, and -- for modules
    marked F(2x2,3x3)-only (the encoder) -- into 256 channels too (gq_conv3.h)."""
    if not (DIRECT_CONV and getattr(conv, "_gq_wino", False) and conv.in_channels % 16 == 0 and x.shape[2] % 8 == 0
            and x.shape[3] % 32 == 0 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()):
        return False
    if conv.out_channels == 128:
        return True
    return conv.out_channels == 256 and not (WINOGRAD_F4 and getattr(conv, "_gq_wino4", False))


def _direct_weights(conv: nn.Conv2d):
    """(Wf, u_scale) of conv3x3_direct / conv1x1_direct, cached until the weight changes."""
    from .. import _lib

    return _cached(conv, "direct_wf", _wkey(conv.weight), lambda: _lib.conv3_weights_f16(conv.weight))


def _stats_of(x: torch.Tensor, pre_bias, groups: int):
    """GroupNorm statistics of x + pre_bias: left by the producer of x, computed earlier for the same pending bias, or
    computed (and remembered on x) here."""
    from .. import _lib

    st = getattr(x, "_gn_stats", None)
    if st is not None and pre_bias is None and st[1] == groups:
        return st[0]
    sp = getattr(x, "_gn_stats_pb", None)
    if sp is not None and sp[1] == groups and sp[2] is pre_bias:
        return sp[0]
    stats = _lib.gn_stats(x, groups, pre_bias)
    x._gn_stats_pb = (stats, groups, pre_bias)
    return stats


def _gn_tuple(norm: nn.GroupNorm, x: torch.Tensor, pre_bias):
    """The ``gn`` argument of wino_conv3x3 / conv3x3_direct: statistics left by the producer of x, or computed here."""
    return (norm.weight, norm.bias, norm.num_groups, norm.eps, True, _stats_of(x, pre_bias, norm.num_groups), pre_bias)


def _pointwise_ok(conv: nn.Conv2d, x: torch.Tensor) -> bool:
    """1x1 convolutions into 128 / 256 / 512 channels (ResnetBlock shortcuts, attention proj_out) as libgqhip's fp16 x 3
    GEMM over the pixels (gq_conv3.h: conv1x1_f16x3) instead of MIOpen's fp32 implicit GEMM (~100 TFLOP/s)."""
    return (DIRECT_CONV_1X1 and FUSED_GN and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()
            and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1
            and conv.out_channels in (128, 256, 512) and conv.in_channels % 128 == 0 and x.dim() == 4
            and (x.shape[2] * x.shape[3]) % 256 == 0 and x.is_contiguous(memory_format=torch.channels_last)
            and not x.is_contiguous())


def _norm_act_conv_small(norm: nn.GroupNorm, conv: nn.Conv2d, x: torch.Tensor, pre_bias=None) -> torch.Tensor:
    """conv(swish(norm(x + pre_bias))) for a 3x3 convolution into <= 4 channels (the decoder's conv_out, unet.py:585-587):
    one libgqhip kernel -- GroupNorm + swish applied while staging, fp32 FMAs -- instead of a normalisation pass and
    MIOpen's implicit GEMM (0.2 + 0.96 ms at 16 x 256 x 256 x 128)."""
    if (FUSED_CONV_OUT and _use_fused(x, norm) and conv.out_channels <= 4 and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
            and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == "zeros"
            and x.shape[1] % 32 == 0 and x.shape[1] <= 512 and x.shape[2] % 16 == 0 and x.shape[3] % 16 == 0):
        from .. import _lib

        if _lib.image_layout(x) == 1:
            ohwi = _cached(conv, "ohwi", _wkey(conv.weight), lambda: conv.weight.detach().permute(0, 2, 3, 1).contiguous())
            return _lib.conv3x3_gn_small(x, ohwi, conv.bias, _gn_tuple(norm, x, pre_bias))
    return conv(_norm_act(norm, x, pre_bias=pre_bias))


def _wkey(*params):
    return tuple((p.data_ptr(), p._version, p.device) for p in params)


def _cached(module: nn.Module, name: str, key, build):
    """Weight-derived data of ``module`` (operand-order copies, fp16 splits, bounds): one dict per module, entries keyed on
    the parameters' (data_ptr, _version, device), dropped together by ``invalidate_caches``."""
    cache = module.__dict__.setdefault("_gq_cache", {})
    ent = cache.get(name)
    if ent is None or ent[0] != key:
        ent = (key, build())
        cache[name] = ent
    return ent[1]


def _conv_f32_ok(conv: nn.Conv2d, x: torch.Tensor, gn: bool) -> bool:
    """The narrow ends of the stack (encoder conv_out, decoder conv_in) on libgqhip's fp32 matrix-core convolution: a fixed
    summation order, hence bit-reproducible -- MIOpen's pick for 512 -> 32 channels is a split-K kernel with atomics."""
    if not (CONV_F32 and FUSED_GN and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and x.dim() == 4
            and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.padding_mode == "zeros" and x.is_contiguous(memory_format=torch.channels_last)
            and not x.is_contiguous()):
        return False
    from .. import _lib

    return _lib.conv_f32_ok(conv.in_channels, conv.out_channels, x.shape[2], x.shape[3], gn)


def _conv_f32(conv: nn.Conv2d, x: torch.Tensor, gn=None) -> torch.Tensor:
    from .. import _lib

    wk = _cached(conv, "f32_wk", _wkey(conv.weight), lambda: _lib.conv_f32_weights(conv.weight))
    return _lib.conv3x3_f32(x, wk, conv.out_channels, bias=conv.bias, gn=gn)


def _conv_in_small(conv: nn.Conv2d, x: torch.Tensor):
    """The encoder's conv_in (unet.py:411-413, 3 -> 128 channels) -> (y, pending_bias).  On the channels_last path: libgqhip's
    fixed-order fp32 kernel with the bias and the first GroupNorm's statistics in its epilogue (round 4; it was the last MIOpen
    convolution of the bench shapes: 0.24 ms per call once MIOpen's CK solver runs, 4 ms for each of a process's first eight
    calls).  Other shapes / layouts: ``_conv``."""
    if (CONV_IN_SMALL and FUSED_GN and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and x.dim() == 4
            and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.padding_mode == "zeros" and GN_GROUPS == 32):
        from .. import _lib

        if _lib.image_layout(x) == 1 and _lib.conv_cin_small_ok(conv.in_channels, conv.out_channels, x.shape[2], x.shape[3]):
            wk = _cached(conv, "cin_small_wk", _wkey(conv.weight), lambda: _lib.conv_cin_small_weights(conv.weight))
            y, st = _lib.conv3x3_cin_small(x, wk, conv.bias, stats_groups=GN_GROUPS)
            y._gn_stats = (st, GN_GROUPS)
            return y, None
    return _conv(conv, x)


def _norm_act_conv_f32(norm: nn.GroupNorm, conv: nn.Conv2d, x: torch.Tensor, pre_bias=None) -> torch.Tensor:
    """conv(swish(norm(x + pre_bias))) for the encoder's conv_out (unet.py:432-436): GroupNorm + swish applied while the
    patch is staged, fp32 matrix cores, fixed summation order."""
    if _conv_f32_ok(conv, x, True) and _use_fused(x, norm):
        from .. import _lib

        if _lib.image_layout(x) == 1:
            return _conv_f32(conv, x, gn=_gn_tuple(norm, x, pre_bias))
    return conv(_norm_act(norm, x, pre_bias=pre_bias))


def _norm_act_conv(norm: nn.GroupNorm, conv: nn.Conv2d, x: torch.Tensor, pre_bias=None, want_stats: bool = False):
    """conv(swish(norm(x + pre_bias))) -> (y, pending_bias).  The GroupNorm(+SiLU) is applied inside the Winograd input
    transform, so the normalised tensor is never written (FUSED_WINO_GN / FUSED_WINO_GN_F4)."""
    if _defer_ok(x, conv) and _direct_ok(conv, x) and _use_fused(x, norm):
        from .. import _lib

        if _lib.image_layout(x) == 1:
            wf, us = _direct_weights(conv)
            gn = _gn_tuple(norm, x, pre_bias)
            if want_stats and FUSED_WINO_TAIL and _lib.gn_nhwc_ok(conv.out_channels, GN_GROUPS):
                y, ostats = _lib.conv3x3_direct(x, wf, us, _gn_act_bound(norm, x), gn=gn, bias=conv.bias,
                                                stats_groups=GN_GROUPS)
                y._gn_stats = (ostats, GN_GROUPS)
                return y, None
            return _lib.conv3x3_direct(x, wf, us, _gn_act_bound(norm, x), gn=gn), conv.bias
    if _defer_ok(x, conv) and _wino_ok(conv, x) and _use_fused(x, norm):
        from .. import _lib

        f4 = WINOGRAD_F4 and getattr(conv, "_gq_wino4", False) and x.shape[2] % 4 == 0 and x.shape[3] % 4 == 0
        if _lib.image_layout(x) == 1 and (FUSED_WINO_GN_F4 if f4 else FUSED_WINO_GN):
            st = getattr(x, "_gn_stats", None)
            if st is not None and pre_bias is None and st[1] == norm.num_groups:
                stats = st[0]
            else:
                stats = _lib.gn_stats(x, norm.num_groups, pre_bias)
            gn = (norm.weight, norm.bias, norm.num_groups, norm.eps, True, stats, pre_bias)
            U = _wino_weights(conv, f4)
            f16 = _f16_args_gn(conv, norm, x, f4)
            if want_stats and FUSED_WINO_TAIL and _lib.gn_nhwc_ok(conv.out_channels, GN_GROUPS):
                y, ostats = _lib.wino_conv3x3(x, U, gn=gn, bias=conv.bias, stats_groups=GN_GROUPS, f16=f16)
                y._gn_stats = (ostats, GN_GROUPS)
                return y, None
            return _lib.wino_conv3x3(x, U, gn=gn, f16=f16), conv.bias
    return _conv(conv, _norm_act(norm, x, pre_bias=pre_bias), want_stats)


_WINO_G2 = [[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]]
_WINO_G4 = [[1 / 4, 0.0, 0.0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
            [1 / 24, -1 / 12, 1 / 6], [0.0, 0.0, 1.0]]


def _wino_weights(conv: nn.Conv2d, f4: bool = False) -> torch.Tensor:
    """U = G g G^T of a 3x3 kernel, [16, Cin, Cout] for F(2x2,3x3) or [36, Cin, Cout] for F(4x4,3x3) (formed in
    fp64, rounded once); cached until the weight changes."""
    def build():
        w = conv.weight
        G = torch.tensor(_WINO_G4 if f4 else _WINO_G2, dtype=torch.float64, device=w.device)
        u = torch.einsum("ik,ockl,jl->ijco", G, w.double(), G).float()
        return u.reshape(-1, w.shape[1], w.shape[0]).contiguous()

    return _cached(conv, "wino_u", _wkey(conv.weight) + (f4,), build)


def _wino_weights_f16(conv: nn.Conv2d, f4: bool = False):
    """(U3, u_scale, Wf2): U = G g G^T scaled by a power of two into fp16's comfortable range and split into two fp16 terms,
    stacked along K as [U_h; U_l; U_h] ([T, 3 Cin, Cout] fp16) -- the weight side of the f16x3 GEMM (gqhip.h:
    wino_in_nhwc_f16x3) -- and, where libgqhip's own GEMM applies (Cin % 32 == 0, Cout % 128 == 0), (U_h, U_l) in MFMA operand
    order for wino_gemm_f16x2 (the [h | l] route).  Cached like (and keyed like) the fp32 U."""
    def build():
        U = _wino_weights(conv, f4)
        amax = float(U.abs().max())
        u_scale = 2.0 ** math.floor(math.log2(16384.0 / max(amax, 1e-30))) if amax > 0 else 1.0
        us = U * u_scale
        h = us.half()
        l = (us - h.float()).half()
        wf2 = None
        if WINOGRAD_OWN_GEMM and U.shape[1] % 32 == 0 and U.shape[2] % 128 == 0:
            from .. import _lib

            wf2 = _lib.wino_weights_operand_order(h, l)
        return torch.cat([h, l, h], 1).contiguous(), u_scale, wf2

    return _cached(conv, "wino_u3", _wkey(conv.weight) + (f4, WINOGRAD_OWN_GEMM), build)


def _gn_act_bound(norm: nn.GroupNorm, x: torch.Tensor) -> float:
    """A rigorous bound on |SiLU(GroupNorm(x))|: a group of n elements has |(x - mean) / sqrt(var + eps)| <= sqrt(n - 1),
    so |y| <= sqrt(n - 1) max|gamma| + max|beta| and |SiLU(y)| <= |y|.  (max|gamma|, max|beta| cached per weight.)"""
    gb = _cached(norm, "gb_max", _wkey(norm.weight, norm.bias),
                 lambda: (float(norm.weight.detach().abs().max()), float(norm.bias.detach().abs().max())))
    n = (x.shape[1] // norm.num_groups) * x.shape[2] * x.shape[3]
    return math.sqrt(max(n - 1, 1)) * gb[0] + gb[1]


class _WeightGuard:
    """Notices parameter writes that bump no version counter (``param.data.mul_()``, EMA swaps through ``.data``): the
    weight-derived caches below are keyed on (data_ptr, _version), which such writes leave unchanged.  One libgqhip launch per
    forward hashes the bytes of every parameter of the module (gqhip.h:gqhip_checksum_tensors, ~0.1-0.2 GB read); the forward
    runs speculatively on the cached data while the sums travel to pinned host memory; before the result is returned they are
    compared with the sums the caches were built from, and on a difference the caches are dropped and the forward runs again.
    The wait is for a copy queued at the START of this forward, so the host stays at most one forward ahead of the device
    and the device never idles."""

    def __init__(self) -> None:
        self.key = None

    def begin(self, module: nn.Module) -> bool:
        from .. import _lib

        params = [p for p in module.parameters()]
        # the checksum kernel reads numel() words from data_ptr(): only DENSE parameters (contiguous, or a dense channels_last
        # layout) -- an expanded or strided view would be read past its own storage, so such a module runs unguarded
        dense = lambda p: p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
        if not params or not all(p.is_cuda and p.dtype == torch.float32 and dense(p) for p in params):
            return False
        key = tuple(p.data_ptr() for p in params)
        if key != self.key:
            try:
                self.table = _lib.checksum_table([p.detach() for p in params])
            except _lib.GqHipError:
                return False
            self.sums = torch.empty(len(params), dtype=torch.int64, device=params[0].device)
            self.host = torch.empty(len(params), dtype=torch.int64).pin_memory()
            self.event = torch.cuda.Event()
            self.baseline = None
            self.key = key
        self.versions = tuple(p._version for p in params)
        _lib.checksum_tensors(self.table, self.sums)
        self.host.copy_(self.sums, non_blocking=True)
        self.event.record()
        return True

    def changed(self) -> bool:
        """True when some parameter's bytes changed WITHOUT its version counter moving (the only case the caches cannot see by
        themselves: an update that bumps ``_version`` -- load_state_dict, an optimizer step -- already changed the cache keys, so the
        speculative run rebuilt what it needed and is not repeated)."""
        self.event.synchronize()
        cur = self.host.clone()
        base, base_v = self.baseline, getattr(self, "baseline_versions", None)
        self.baseline, self.baseline_versions = cur, self.versions
        if base is None or torch.equal(cur, base):
            return False
        moved = (cur != base).tolist()
        return any(m and v == bv for m, v, bv in zip(moved, self.versions, base_v))


def _with_stats_arena(module: nn.Module, run, x: torch.Tensor):
    """run(x) with every GroupNorm statistics record of the forward carved out of an arena of the module (one fill per forward
    instead of one small memset launch per producing kernel: _lib.StatsArena).  One arena per (thread, stream): the buffer is
    re-zeroed and reused in stream order, so two streams (or threads) running the same module must not share it.  Never under a
    graph capture: a captured graph would bake in the arena's address, and a later eager forward with a bigger batch reallocates
    the buffer -- the replay would then zero and accumulate into memory the allocator may have handed to someone else; captured
    forwards take per-call records from the caching allocator, which the capture keeps alive."""
    if not (STATS_ARENA and FUSED_GN and x.is_cuda and not torch.is_grad_enabled()) or torch.cuda.is_current_stream_capturing():
        return run(x)
    import threading

    from .. import _lib

    key = (threading.get_ident(), torch.cuda.current_stream(x.device).cuda_stream)
    arenas = module.__dict__.setdefault("_gq_stats_arenas", {})
    arena = arenas.pop(key, None)
    if arena is None:
        arena = _lib.StatsArena()
        module.__dict__.setdefault("_gq_stats_arena", arena)      # (the first one: what tests / tools look at)
        # bounded: workloads that churn threads or streams would otherwise keep one device arena per (thread, stream)
        # they ever used; the least recently used one goes (dicts keep insertion order; a hit re-inserts below).  The device
        # memory of an evicted arena is released stream-ordered by the caching allocator, so a forward still in flight on it is safe.
        while len(arenas) >= _MAX_STATS_ARENAS:
            arenas.pop(next(iter(arenas)))
    arenas[key] = arena
    with _lib.stats_arena(arena, x.device):
        return run(x)


_MAX_STATS_ARENAS = 8


def _guarded(module: nn.Module, run0, x: torch.Tensor):
    """run(x) with the module's weight caches verified against the parameters' bytes (see _WeightGuard).  Costs one host wait per
    forward (for a copy queued at the forward's start: the host stays at most one module ahead of the device).  A caller that
    never writes parameters through ``.data`` and wants a fully asynchronous forward opts out per module:
    ``encoder.weight_guard = False`` (or process-wide: GQHIP_WEIGHT_GUARD=0)."""
    run = lambda t: _with_stats_arena(module, run0, t)
    if not (WEIGHT_GUARD and getattr(module, "weight_guard", True) and x.is_cuda and not torch.is_grad_enabled()
            and not torch.cuda.is_current_stream_capturing()):
        return run(x)
    guard = module.__dict__.get("_gq_guard")
    if guard is None:
        guard = module.__dict__["_gq_guard"] = _WeightGuard()
    if not guard.begin(module):
        return run(x)
    y = run(x)
    if guard.changed():
        invalidate_caches(module)
        y = run(x)
    return y


def invalidate_caches(module: nn.Module) -> None:
    """Drop every weight-derived cache under ``module`` (Winograd U matrices, operand-order fp16 splits, the sub-pixel phase
    matrices, the fused q/k/v matrix, operand bounds): every one of them lives in the owning module's ``_gq_cache`` dict
    (see ``_cached``), keyed on the parameters' (data_ptr, _version, device) -- which follows ``load_state_dict``, optimizer
    steps, ``.to()`` and any in-place op on the parameter.  Writes through ``param.data`` bump no version counter: those are
    caught by the content-hash guard of every inference forward (``_WeightGuard``), which calls this function itself, so
    nobody has to remember to."""
    for m in module.modules():
        m.__dict__.pop("_gq_cache", None)


def _drop_caches_after_load(module, incompatible_keys) -> None:
    invalidate_caches(module)


def mark_winograd(module: nn.Module, f4: bool = False) -> None:
    """Flag the stride-1, padding-1 3x3 convolutions of ``module`` for the Winograd path (see ``_conv``);
    ``f4``: F(4x4,3x3) where the spatial size allows it (decoder only: larger rounding error)."""
    for m in module.modules():
        if (isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1)
                and m.dilation == (1, 1) and m.groups == 1 and m.padding_mode == "zeros"):
            m._gq_wino = True
            m._gq_wino4 = f4


def _add(a: torch.Tensor, b: torch.Tensor, bias=None) -> torch.Tensor:
    """a + b (+ bias[c]) -- the residual add with the pending biases folded in."""
    if (bias is not None and a.is_cuda and a.shape == b.shape and a.dtype == torch.float32
            and not torch.is_grad_enabled()):
        from .. import _lib

        la, lb = _lib.image_layout(a), _lib.image_layout(b)
        if FUSED_ADD_STATS and la == 1 and lb == 1 and _lib.gn_nhwc_ok(a.shape[1], GN_GROUPS):
            # every residual add of this UNet feeds a GroupNorm(32): leave its statistics with the sum, the
            # consumer (_norm_act) then skips its own statistics pass over the tensor
            y, stats = _lib.add_bias_stats(a, b, bias, GN_GROUPS)
            y._gn_stats = (stats, GN_GROUPS)
            return y
        if la is not None and la == lb and ((la == 0 and (a.shape[2] * a.shape[3]) % 4 == 0) or
                                            (la == 1 and a.shape[1] % 4 == 0)):
            return _lib.add_bias(a, b, bias)
    return _materialize(a + b, bias)


FUSED_GN = True    # module-level switches (tests / A-B timing)
WINOGRAD = True          # decoder 3x3 convs with >= WINOGRAD_MIN_CH channels: Winograd F(2x2,3x3) + 16 hipBLASLt GEMMs
WINOGRAD_MIN_CH = 128
FUSED_WINO_TAIL = True   # Winograd output transform + bias + residual add + next GroupNorm's statistics in one pass
WINOGRAD_F4 = True       # decoder: F(4x4,3x3) (36 GEMMs on 6x6 tiles) instead of F(2x2,3x3)
# The 16 / 36 Winograd GEMMs as ONE fp16 batched GEMM with fp32 accumulation whose K axis carries the three products of
# two-term fp16 splits of both operands (22-bit significands): measured error 2.5-2.9e-7 of sum|a||b| against 2.6-3.5e-7
# for hipBLASLt's fp32 GEMM (itself a split-bf16 emulation on gfx950), at 1.2x (128 channels) to 2.5x (512) its speed
# (tools/bmm_bf16x3.py).  False: the library's fp32 GEMM.
WINOGRAD_F16X3 = True
# the 256- / 512-channel Winograd GEMMs through libgqhip's wino_gemm_f16x2 on the [h | l] operand (weights in MFMA
# operand order, three products in the kernel) where its grid fills the chip (_lib.own_gemm_fits: everything but the
# 32 x 32 levels' 36 x 1024-tile GEMMs); else the library GEMM over [h | h | l]
WINOGRAD_OWN_GEMM = True
# 3x3 convolutions into 128 channels (256 x 256 level) -- and, where the alternative is F(2x2,3x3) (the encoder), into 256
# channels (128 x 128 level) -- as a direct fp16 x 3 implicit GEMM instead of Winograd: reads the activation once and
# writes the result once where Winograd moves 6.4 / 10.7 GB of transformed tensors per convolution at 256 x 256
DIRECT_CONV = True
DIRECT_CONV_S2 = True      # Downsample (pad + 3x3 stride 2) as an fp16 x 3 convolution on the four phase images of x
DIRECT_CONV_1X1 = True     # 1x1 shortcut / proj_out convolutions as an fp16 x 3 GEMM with the split of x inside the kernel
FUSED_CONV_OUT = True      # decoder conv_out (128 -> 3) with norm_out + swish fused in: one VALU kernel
# encoder conv_out (512 -> 2 z, norm_out + swish fused in) and decoder conv_in (z -> 512) on the fp32 matrix cores with a fixed
# summation order (bit-reproducible; MIOpen's pick for the former combines split-K partial sums with atomics)
CONV_F32 = True
# every inference forward of Encoder / Decoder on a HIP device checks its weight-derived caches against a content hash of the
# parameters (catches ``param.data`` writes, which bump no version counter); see _WeightGuard
WEIGHT_GUARD = os.environ.get("GQHIP_WEIGHT_GUARD", "1") != "0"   # (the env switch: A/B timing)
# GroupNorm+SiLU applied inside the Winograd input transforms (F(2x2,3x3) / F(4x4,3x3)): the normalised tensor is never
# written or re-read.  Bit-identical V to gn_apply + plain transform (same folded scale / shift, same silu_f32); pays
# since the loads of a tile are issued ahead of the activations (branch-free borders): 53.4 -> 50.8 ms / step.
FUSED_WINO_GN = True
FUSED_WINO_GN_F4 = True
# also in the encoder: measured perturbation of z 4.2e-6 vs the CPU reference (direct MIOpen convs: 3.4e-6), no index
# change on the CPU golden nor on 16 384 rows against the direct-conv encoder (tools/encoder_winograd_check.py)
WINOGRAD_ENCODER = True
# Upsample: nearest x2 + conv3x3 as four 2x2 phase convolutions of the low-resolution input (2.25x fewer flops), computed directly by
# libgqhip's upconv2x_f16x3 (one kernel); shapes it does not tile: NHWC upsample copy + the ordinary convolution routes
DIRECT_UPCONV = True
FUSED_QKV = True         # attention: q, k, v as one GEMM with fused biases (channels_last)
FUSED_ADD_STATS = True   # residual add also produces the next GroupNorm's statistics (channels_last only)
STATS_ARENA = True       # the statistics records of a forward from one arena zeroed by one fill (not ~60 memset launches per step)
CONV_IN_SMALL = True     # the encoder's conv_in (3 -> 128) on libgqhip's fixed-order fp32 kernel (+ bias + statistics), not MIOpen
GN_GROUPS = 32     # unet.py:54-57: every Normalize is GroupNorm(32, C, eps=1e-6)
DEFER_BIAS = True
ATTN_F16X3 = True   # both attention GEMMs as fp16 x 3 library GEMMs (split of q, k, v and softmax + split in libgqhip kernels)
ATTN_MATH = "auto"  # explicit matmul/softmax/matmul instead of the fused SDPA kernel: "auto" = on HIP devices


def _sdpa(q, k, v):
    # single head, head_dim = C = 512, fp32: two hipBLASLt GEMMs + a softmax beat the fused attn_fwd kernel
    # (3.3 -> 1.9 ms per step at bs 16; A/B-timed, +2 % end to end); same function, fp32 rounding-level differences
    if ATTN_MATH is True or (ATTN_MATH == "auto" and q.is_cuda):
        w = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * (q.shape[-1] ** -0.5), dim=-1)
        return torch.matmul(w, v)
    return F.scaled_dot_product_attention(q, k, v)  # scale c**-0.5


def _conv3(cin: int, cout: int, padding_mode: str = "zeros", stride: int = 1, padding: int = 1) -> nn.Conv2d:
    return nn.Conv2d(cin, cout, 3, stride, padding, padding_mode=padding_mode)


class ResnetBlock(nn.Module):
    """GN -> swish -> conv3 -> GN -> swish -> conv3 (+ 1x1 shortcut); unet.py:100-153."""

    def __init__(self, cin: int, cout: int, dropout: float = 0.0, padding_mode: str = "zeros") -> None:
        super().__init__()
        self.in_channels, self.out_channels = cin, cout
        self.norm1 = _gn(cin)
        self.conv1 = _conv3(cin, cout, padding_mode)
        self.norm2 = _gn(cout)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = _conv3(cout, cout, padding_mode)
        if cin != cout:
            self.nin_shortcut = nn.Conv2d(cin, cout, 1)

    def forward(self, x: torch.Tensor, pre_bias=None) -> torch.Tensor:
        """``pre_bias``: bias of the conv that produced ``x``, not yet added (deferred path)."""
        h, b1 = _norm_act_conv(self.norm1, self.conv1, x, pre_bias, want_stats=True)   # norm2 follows
        if self.in_channels != self.out_channels and _pointwise_ok(self.nin_shortcut, x) and _use_fused(x, self.norm1):
            # nin(x + pb) as one fp16 x 3 GEMM over the pixels; x + pb is split inside the kernel, its power-of-two scale
            # comes (on the device) from the statistics norm1 needed anyway: |x + pb| <= sqrt(group sum of squares)
            from .. import _lib

            wf, us = _direct_weights(self.nin_shortcut)
            scales = _lib.f16_scales(_stats_of(x, pre_bias, GN_GROUPS), 1.0, us)
            xs, bs = _lib.conv1x1_direct(x, wf, us, scales, pre_bias=pre_bias), self.nin_shortcut.bias
        elif self.in_channels != self.out_channels:
            # nin(x + pb) = nin_nobias(x) + W.pb + nin.bias : every constant goes into the fused add
            xs, bs = _conv(self.nin_shortcut, x)
            if pre_bias is not None:
                wpb = self.nin_shortcut.weight.reshape(self.out_channels, self.in_channels) @ pre_bias
                bs = wpb if bs is None else bs + wpb
        else:
            xs, bs = x, pre_bias
        if self.dropout.p > 0.0 and self.training:
            h, bias = _conv(self.conv2, self.dropout(_norm_act(self.norm2, h, pre_bias=b1)))
        else:   # dropout is the identity (unet.py:148 with p = 0 / eval)
            if (FUSED_WINO_TAIL and _defer_ok(h, self.conv2) and _wino_ok(self.conv2, h) and _use_fused(h, self.norm2)
                    and xs.shape[1] == self.out_channels and xs.is_contiguous(memory_format=torch.channels_last)
                    and not xs.is_contiguous()):
                from .. import _lib

                if (_direct_ok(self.conv2, h) and _lib.gn_nhwc_ok(self.out_channels, GN_GROUPS)
                        and _lib.image_layout(h) == 1):
                    bias = self.conv2.bias if bs is None else self.conv2.bias + bs
                    wf, us = _direct_weights(self.conv2)
                    y, ostats = _lib.conv3x3_direct(h, wf, us, _gn_act_bound(self.norm2, h),
                                                    gn=_gn_tuple(self.norm2, h, b1), residual=xs, bias=bias,
                                                    stats_groups=GN_GROUPS)
                    y._gn_stats = (ostats, GN_GROUPS)
                    return y
                if _lib.gn_nhwc_ok(self.out_channels, GN_GROUPS) and _lib.image_layout(h) == 1:
                    # conv2, its bias, the shortcut's constants, the residual add and the next GroupNorm's statistics
                    # in one output-transform pass (and, with F(4x4,3x3), norm2 + swish inside the input transform)
                    bias = self.conv2.bias if bs is None else self.conv2.bias + bs
                    f4 = (WINOGRAD_F4 and getattr(self.conv2, "_gq_wino4", False) and h.shape[2] % 4 == 0
                          and h.shape[3] % 4 == 0)
                    if FUSED_WINO_GN_F4 if f4 else FUSED_WINO_GN:
                        st = getattr(h, "_gn_stats", None)
                        if st is not None and b1 is None and st[1] == self.norm2.num_groups:
                            stats = st[0]
                        else:
                            stats = _lib.gn_stats(h, self.norm2.num_groups, b1)
                        gn = (self.norm2.weight, self.norm2.bias, self.norm2.num_groups, self.norm2.eps, True, stats, b1)
                        src = h
                        f16 = _f16_args_gn(self.conv2, self.norm2, h, f4)
                    else:
                        gn, src = None, _norm_act(self.norm2, h, pre_bias=b1)
                        f16 = _f16_args(self.conv2, src, f4)
                    y, ostats = _lib.wino_conv3x3(src, _wino_weights(self.conv2, f4), gn=gn, residual=xs, bias=bias,
                                                  stats_groups=GN_GROUPS, f16=f16)
                    y._gn_stats = (ostats, GN_GROUPS)
                    return y
            h, bias = _conv(self.conv2, _norm_act(self.norm2, h, pre_bias=b1))
        if bs is not None:
            bias = bs if bias is None else bias + bs
        return _add(xs, h, bias)


class AttnBlock(nn.Module):
    """Single-head self attention over the h*w positions; unet.py:166-206."""

    def __init__(self, ch: int) -> None:
        super().__init__()
        self.norm = _gn(ch)
        self.q = nn.Conv2d(ch, ch, 1)
        self.k = nn.Conv2d(ch, ch, 1)
        self.v = nn.Conv2d(ch, ch, 1)
        self.proj_out = nn.Conv2d(ch, ch, 1)

    def _qkv_weights(self):
        """([c, 3c] GEMM matrix, [3c] bias) of the q / k / v 1x1 convolutions; cached until a weight changes."""
        ps = (self.q.weight, self.k.weight, self.v.weight, self.q.bias, self.k.bias, self.v.bias)

        def build():
            c = self.q.weight.shape[0]
            return (torch.cat([p.detach().reshape(c, c) for p in ps[:3]], 0).t().contiguous(),
                    torch.cat([p.detach() for p in ps[3:]], 0).contiguous())

        return _cached(self, "qkv", _wkey(*ps), build)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        b, c, h, w = x.shape
        a_scale = 1.0      # power of two still pending on the attention output (fp16 x 3 route)
        y = _norm_act(self.norm, x, act=False)
        # [b, c, h, w] -> [b, 1, hw, c]
        if y.is_contiguous(memory_format=torch.channels_last) and not y.is_contiguous():
            # channels_last: [b, hw, c] is a free view of the conv output and of the result
            if FUSED_QKV and y.is_cuda and y.dtype == torch.float32 and not torch.is_grad_enabled():
                # q, k, v = three 1x1 convolutions of the same tensor = ONE GEMM [b*hw, c] x [c, 3c] with the biases in
                # its epilogue (instead of 3 MIOpen launches + 3 bias-add passes); the thirds are strided views
                wqkv, bqkv = self._qkv_weights()
                if DIRECT_CONV_1X1 and c == 512 and (h * w) % 256 == 0:
                    # ... as libgqhip's fp16 x 3 GEMM over the pixels (the fp32 library GEMM runs at ~130 TFLOP/s)
                    from .. import _lib

                    qkv_wf, qkv_us = _cached(self, "qkv_wf", _wkey(self.q.weight, self.k.weight, self.v.weight),
                                             lambda: _lib.conv3_weights_f16(wqkv.t().reshape(3 * c, c, 1, 1)))
                    qkv = _lib.conv1x1_direct(y, qkv_wf, qkv_us, _gn_act_bound(self.norm, x), bias=bqkv)
                    qkv = qkv.permute(0, 2, 3, 1).reshape(b, 1, h * w, 3 * c)
                else:
                    qkv = torch.addmm(bqkv, y.permute(0, 2, 3, 1).reshape(b * h * w, c), wqkv).view(b, 1, h * w, 3 * c)
                q, k, v = qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:]
            else:
                qkv = None
                q, k, v = (f(y).permute(0, 2, 3, 1).reshape(b, 1, h * w, c) for f in (self.q, self.k, self.v))
            if ATTN_F16X3 and qkv is not None and qkv.is_contiguous() and c % 4 == 0 and _f16_gemm_ok(qkv):
                from .. import _lib

                if (h * w) in _lib.ATTN_L_OK:
                    # both attention GEMMs as fp16 GEMMs over K axes of two-term fp16 splits (fp32 accumulation): the fp32
                    # GEMMs are a split-bf16 emulation at ~120 TFLOP/s; operand scales from rigorous bounds (GroupNorm bound x
                    # weight row sums), the softmax writes the split operand of the second GEMM directly
                    yb = _gn_act_bound(self.norm, x)
                    a, a_scale = _lib.attention_f16x3(qkv.view(b, h * w, 3 * c), self._qk_bound(yb), self._v_bound(yb))
            if a_scale == 1.0:
                a = _sdpa(q, k, v)
            a = a.reshape(b, h, w, c).permute(0, 3, 1, 2)   # (times a_scale: folded into proj_out's scale below)
        else:
            q, k, v = (f(y).reshape(b, 1, c, h * w).transpose(2, 3).contiguous() for f in (self.q, self.k, self.v))
            a = _sdpa(q, k, v)
            a = a.transpose(2, 3).reshape(b, c, h, w)
        if (_pointwise_ok(self.proj_out, a) and FUSED_ADD_STATS and x.is_contiguous(memory_format=torch.channels_last)
                and not x.is_contiguous() and self.proj_out.bias is not None):
            # proj_out + bias + residual add + the next GroupNorm's statistics in one fp16 x 3 GEMM over the pixels.  Scale:
            # the attention output is a convex combination of the rows of v, so |a| <= max|v| <= max|y| max_j ||W_v[j]||_1 +
            # max|b_v|, with |y| bounded by the GroupNorm bound
            from .. import _lib

            if _lib.image_layout(a) == 1 and _lib.gn_nhwc_ok(c, GN_GROUPS):
                wf, us = _direct_weights(self.proj_out)
                out, st = _lib.conv1x1_direct(a, wf, us, self._v_bound(_gn_act_bound(self.norm, x)) / a_scale, residual=x,
                                              bias=self.proj_out.bias, stats_groups=GN_GROUPS, post_scale=a_scale)
                out._gn_stats = (st, GN_GROUPS)
                return out
        if a_scale != 1.0:
            a = a * a_scale
        p, pb = _conv(self.proj_out, a)
        return _add(x, p, pb)

    def _qk_bound(self, y_bound: float) -> float:
        """max(|q|, |k|) for |y| <= y_bound: q = W_q y + b_q, k = W_k y + b_k."""
        ps = (self.q.weight, self.k.weight, self.q.bias, self.k.bias)

        def build():
            c = self.q.weight.shape[0]
            rs = max(float(self.q.weight.detach().reshape(c, -1).abs().sum(1).max()),
                     float(self.k.weight.detach().reshape(c, -1).abs().sum(1).max()))
            return rs, max(float(self.q.bias.detach().abs().max()), float(self.k.bias.detach().abs().max()))

        qkb = _cached(self, "qk_bound", _wkey(*ps), build)
        return y_bound * qkb[0] + qkb[1]

    def _v_bound(self, y_bound: float) -> float:
        """max|v| for |y| <= y_bound: v = W_v y + b_v."""
        def build():
            wv = self.v.weight.detach().reshape(self.v.weight.shape[0], -1)
            return float(wv.abs().sum(1).max()), float(self.v.bias.detach().abs().max())

        vb = _cached(self, "v_bound", _wkey(self.v.weight, self.v.bias), build)
        return y_bound * vb[0] + vb[1]


class Downsample(nn.Module):
    """Asymmetric (0,1,0,1) pad + stride-2 conv3; unet.py:76-97."""

    def __init__(self, ch: int, with_conv: bool = True, padding_mode: str = "zeros") -> None:
        super().__init__()
        self.with_conv = with_conv
        self.mode = "constant" if padding_mode == "zeros" else padding_mode
        if with_conv:
            self.conv = _conv3(ch, ch, stride=2, padding=0)

    def forward(self, x: torch.Tensor):
        """Returns (y, pending_bias) -- see ``_conv``."""
        if not self.with_conv:
            return F.avg_pool2d(x, 2, 2), None
        conv = self.conv
        if (DIRECT_CONV_S2 and self.mode == "constant" and FUSED_GN and x.is_cuda and x.dtype == torch.float32
                and not torch.is_grad_enabled() and conv.out_channels in (128, 256, 512) and conv.in_channels % 16 == 0
                and conv.bias is not None and x.dim() == 4 and x.shape[2] % 16 == 0 and x.shape[3] % 64 == 0
                and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()):
            # pad + stride-2 convolution + bias + the next GroupNorm's statistics as ONE fp16 x 3 kernel on the four phase
            # images of x (MIOpen: a padding pass + an fp32 implicit GEMM, 0.7-1.0 ms); the scale of x comes, on the device,
            # from the statistics its producer left behind: |x| <= sqrt(group sum of squares)
            from .. import _lib

            if _lib.gn_nhwc_ok(x.shape[1], GN_GROUPS) and _lib.gn_nhwc_ok(conv.out_channels, GN_GROUPS):
                s2_wf, s2_us = _cached(conv, "s2_wf", _wkey(conv.weight), lambda: _lib.conv3s2_weights_f16(conv.weight))
                scales = _lib.f16_scales(_stats_of(x, None, GN_GROUPS), 1.0, s2_us)
                y, st = _lib.conv3x3s2_direct(x, s2_wf, s2_us, scales, bias=conv.bias, stats_groups=GN_GROUPS)
                y._gn_stats = (st, GN_GROUPS)
                return y, None
        if self.mode == "constant":
            x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
        else:
            x = F.pad(x, (0, 1, 0, 1), mode=self.mode)
        return _conv(self.conv, x)


class Upsample(nn.Module):
    """Nearest x2 + conv3; unet.py:60-73."""

    def __init__(self, ch: int, with_conv: bool = True, padding_mode: str = "zeros") -> None:
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = _conv3(ch, ch, padding_mode)

    def _phase_weights(self) -> torch.Tensor:
        """[4*Cin, 4*Cout] matrix of the 3x3 kernel folded onto the low-resolution grid: one 2x2 kernel per output phase
        (a, b); tap u of phase a collects the kernel rows that land on source row i-1+a+u: a=0: {0}, {1,2};  a=1: {0,1}, {2}
        (same for columns).  Row index (2u+v)*Cin + ci, column (2a+b)*Cout + co.  Cached until the weight changes."""
        def build():
            w = self.conv.weight.detach()
            rows = (((0,), (1, 2)), ((0, 1), (2,)))
            cout, cin = w.shape[0], w.shape[1]
            m = w.new_zeros(2, 2, cin, 2, 2, cout)          # [u, v, ci, a, b, co]
            for a in range(2):
                for b in range(2):
                    for u in range(2):
                        for v in range(2):
                            for kh in rows[a][u]:
                                for kw in rows[b][v]:
                                    m[u, v, :, a, b, :] += w[:, :, kh, kw].t()
            return m.reshape(4 * cin, 4 * cout).contiguous()

        return _cached(self, "phase_w", _wkey(self.conv.weight), build)

    def forward(self, x: torch.Tensor):
        """Returns (y, pending_bias) -- see ``_conv``."""
        fast = (FUSED_GN and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()
                and x.shape[1] % 4 == 0 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous())
        if (fast and DIRECT_UPCONV and self.with_conv and _defer_ok(x, self.conv) and self.conv.kernel_size == (3, 3)
                and self.conv.stride == (1, 1) and self.conv.padding == (1, 1)):
            # nearest x2 then conv3x3 == four 2x2 convolutions of the low-resolution input (one per output phase): 16 instead
            # of 36 tap evaluations per source pixel, no upsampled tensor -- computed directly by libgqhip's upconv2x_f16x3 (one
            # kernel per Upsample), with the bias added and the statistics of the next ResnetBlock's norm1 left behind; the
            # scale of x comes, on the device, from the statistics its producer left (sqrt of a group's sum of squares bounds
            # its largest element)
            from .. import _lib

            b, c, h, w = x.shape
            st = getattr(x, "_gn_stats", None)
            cout = self.conv.out_channels
            if (st is not None and cout in (128, 256, 512) and c % 16 == 0 and h % 8 == 0 and w % 32 == 0
                    and self.conv.bias is not None and _lib.gn_nhwc_ok(cout, GN_GROUPS)):
                wf, wf_us = _cached(self, "phase_wf", _wkey(self.conv.weight),
                                    lambda: _lib.upconv_weights_f16(self._phase_weights(), c, cout))
                scales = _lib.f16_scales(st[0], 1.0, wf_us)
                y, ostats = _lib.upconv2x_direct(x, wf, wf_us, scales, bias=self.conv.bias, stats_groups=GN_GROUPS)
                y._gn_stats = (ostats, GN_GROUPS)
                return y, None
        if fast:   # shapes the direct kernel does not tile: upsample (a plain NHWC copy kernel), then the convolution
            from .. import _lib

            x = _lib.upsample2x_nhwc(x)   # ATen's NHWC nearest kernel runs at ~1.6 TB/s; this one is a plain copy
        else:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        return _conv(self.conv, x) if self.with_conv else (x, None)


def _make_attn(ch: int, attn_type: str) -> nn.Module:
    if attn_type == "vanilla":
        return AttnBlock(ch)
    if attn_type == "none":
        return nn.Identity()
    raise NotImplementedError(f"attn_type={attn_type!r} is not on the SD3-UNet hot path (reference unet.py:291-314)")


class _Level(nn.Module):
    """One resolution level: res blocks (+ per-block attention) and the optional resampler."""

    def __init__(self) -> None:
        super().__init__()
        self.block = nn.ModuleList()
        self.attn = nn.ModuleList()

    def run(self, h: torch.Tensor, pre_bias=None) -> torch.Tensor:
        has_attn = len(self.attn) > 0
        for i, blk in enumerate(self.block):
            h = blk(h, pre_bias if i == 0 else None)
            if has_attn:
                h = self.attn[i](h)
        return h


class _Mid(nn.Module):
    def __init__(self, ch: int, dropout: float, padding_mode: str) -> None:
        super().__init__()
        self.block_1 = ResnetBlock(ch, ch, dropout, padding_mode)
        self.block_2 = ResnetBlock(ch, ch, dropout, padding_mode)  # no mid attention (unet.py:391, :500)

    def forward(self, h: torch.Tensor, pre_bias=None) -> torch.Tensor:
        return self.block_2(self.block_1(h, pre_bias))


class Encoder(nn.Module):
    def __init__(self, *, ch: int, out_ch: int = 3, ch_mult: Sequence[int] = (1, 2, 4, 8), num_res_blocks: int,
                 attn_resolutions: Sequence[int], dropout: float = 0.0, resamp_with_conv: bool = True,
                 in_channels: int, resolution: int, z_channels: int, double_z: bool = True,
                 use_linear_attn: bool = False, attn_type: str = "vanilla", padding_mode: str = "zeros",
                 **ignore_kwargs) -> None:
        super().__init__()
        self.register_load_state_dict_post_hook(_drop_caches_after_load)
        if use_linear_attn:
            attn_type = "linear"
        self.ch, self.resolution, self.in_channels = ch, resolution, in_channels
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.conv_in = _conv3(in_channels, ch, padding_mode)
        widths = [ch] + [ch * m for m in ch_mult]
        res = resolution
        self.down = nn.ModuleList()
        for lvl in range(self.num_resolutions):
            level = _Level()
            cin = widths[lvl]
            for _ in range(num_res_blocks):
                level.block.append(ResnetBlock(cin, widths[lvl + 1], dropout, padding_mode))
                cin = widths[lvl + 1]
                if res in attn_resolutions:
                    level.attn.append(_make_attn(cin, attn_type))
            if lvl != self.num_resolutions - 1:
                level.downsample = Downsample(cin, resamp_with_conv, padding_mode)
                res //= 2
            self.down.append(level)
        top = widths[-1]
        self.mid = _Mid(top, dropout, padding_mode)
        self.norm_out = _gn(top)
        self.conv_out = _conv3(top, 2 * z_channels if double_z else z_channels, padding_mode)
        if WINOGRAD_ENCODER:
            mark_winograd(self)   # F(2x2,3x3) only: the encoder's rounding decides indices

    def invalidate_caches(self) -> None:
        invalidate_caches(self)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _guarded(self, self._forward, x)

    def _forward(self, x: torch.Tensor) -> torch.Tensor:
        x = _match_layout(x, self.conv_in)
        h, pb = _conv_in_small(self.conv_in, x)
        for lvl, level in enumerate(self.down):
            h, pb = level.run(h, pb), None
            if lvl != self.num_resolutions - 1:
                h, pb = level.downsample(h)
        h = self.mid(h, pb)
        return _norm_act_conv_f32(self.norm_out, self.conv_out, h)


class Decoder(nn.Module):
    def __init__(self, *, ch: int, out_ch: int, ch_mult: Sequence[int] = (1, 2, 4, 8), num_res_blocks: int,
                 attn_resolutions: Sequence[int], dropout: float = 0.0, resamp_with_conv: bool = True,
                 in_channels: int, resolution: int, z_channels: int, give_pre_end: bool = False,
                 tanh_out: bool = False, use_linear_attn: bool = False, attn_type: str = "vanilla",
                 padding_mode: str = "zeros", **ignore_kwargs) -> None:
        super().__init__()
        self.register_load_state_dict_post_hook(_drop_caches_after_load)
        if use_linear_attn:
            attn_type = "linear"
        self.ch, self.resolution, self.in_channels = ch, resolution, in_channels
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.give_pre_end, self.tanh_out = give_pre_end, tanh_out
        top = ch * ch_mult[-1]
        res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, res, res)
        self.conv_in = _conv3(z_channels, top, padding_mode)
        self.mid = _Mid(top, dropout, padding_mode)
        levels = []
        cin = top
        for lvl in reversed(range(self.num_resolutions)):
            level = _Level()
            cout = ch * ch_mult[lvl]
            for _ in range(num_res_blocks + 1):
                level.block.append(ResnetBlock(cin, cout, dropout, padding_mode))
                cin = cout
                if res in attn_resolutions:
                    level.attn.append(_make_attn(cin, attn_type))
            if lvl != 0:
                level.upsample = Upsample(cin, resamp_with_conv, padding_mode)
                res *= 2
            levels.append(level)
        self.up = nn.ModuleList(reversed(levels))  # index = resolution level, like the reference
        self.norm_out = _gn(cin)
        self.conv_out = _conv3(cin, out_ch, padding_mode)
        mark_winograd(self, f4=True)

    def get_last_layer(self, **kwargs) -> torch.Tensor:
        return self.conv_out.weight

    def invalidate_caches(self) -> None:
        invalidate_caches(self)

    def forward(self, z: torch.Tensor, **kwargs) -> torch.Tensor:
        self.last_z_shape = z.shape
        return _guarded(self, self._forward, z)

    def _forward(self, z: torch.Tensor) -> torch.Tensor:
        z = _match_layout(z, self.conv_in)
        if _conv_f32_ok(self.conv_in, z, False):
            h, pb = _conv_f32(self.conv_in, z), None
        else:
            h, pb = _conv(self.conv_in, z)
        h, pb = self.mid(h, pb), None
        for lvl in reversed(range(self.num_resolutions)):
            h, pb = self.up[lvl].run(h, pb), None
            if lvl != 0:
                h, pb = self.up[lvl].upsample(h)
        if self.give_pre_end:
            return h
        h = _norm_act_conv_small(self.norm_out, self.conv_out, h, pb)
        return torch.tanh(h) if self.tanh_out else h
