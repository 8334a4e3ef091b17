"""fp64 references and error-model gates for the conv-stack tests (test infrastructure, `-m gpu` only).

Rule: no test compares two device routes with each other under a measured tolerance, and nothing is compared
through upstream library layers.  A route is compared with an fp64 evaluation of THE SAME FUNCTION ON THE SAME INPUT
(``twin64``: the module itself, deep-copied to double -- every fast path of pit_hip.modules.unet requires float32, so the twin
runs plain ATen ops) and the gate is one of

* a **linear-op gate**: ``|y - ref| <= c * (sum |x||w| + |bias| + |residual|)`` per output element (``lin_gate``), with a charged
  coefficient ``c`` per kind of kernel (``C_*`` below: what the kernel's arithmetic can lose, with headroom, never a number read
  off a passing run of the comparison it gates);
* a **propagated first-order model** for a block of several layers (``resnet_ref_and_bound``, ``attn_ref_and_bound``): every
  linear op charged as above, GroupNorm / swish / softmax propagated with their derivatives, 25 % slack for the second order;
* the **product contract** for whole encoders / decoders (``contract_z``, ``contract_x``): the tolerance DESIGN.md states for the
  drop-in against the reference path -- |z - z_ref| <= 5e-5 of max(1, |z|max) for the encoder (the layer that decides tokens),
  2e-4 of max(1, |x|max) and PSNR >= 60 dB for reconstructions -- applied against fp64, i.e. every route has to meet the contract
  by itself.
"""
from __future__ import annotations

import copy

import torch
import torch.nn.functional as F

U24 = 2.0 ** -24

# charged coefficients, relative to sum |x||w| (+ |bias| + |residual|) of one output element
C_FP32 = 16 * U24          # an fp32 FMA / fp32-MFMA chain in any order, K <= a few thousand, + bias / residual adds
C_F16X3 = 20 * U24         # three fp16 products of two-term splits (dropped l*l: 2^-22 = 4u; fp32 accumulation; sub-fp16-normal l parts)
C_WINO_F2 = 24 * U24       # F(2x2,3x3): input transform sums 4 terms (x2 per axis), output transform 9 -> a few more roundings per term
C_WINO_F4 = 200 * U24      # F(4x4,3x3): transform matrices with entries up to 8 / down to 1/24: the known ~x10 amplification of F(2x2)
C_LIB_FP32 = 32 * U24      # the library's fp32 convolution / GEMM (on gfx950 a split-bf16 emulation, 2.6-3.5e-7 measured by its vendor tests)
SECOND_ORDER = 1.25

Z_TOL = 5e-5               # DESIGN.md "stated tolerance": encoder output against the reference path
X_TOL = 2e-4               # ... reconstructions (max-abs, of max(1, |x|max)); and PSNR >= 60 dB
PSNR_MIN = 60.0


def twin64(module: torch.nn.Module) -> torch.nn.Module:
    """The same module (same class, same parameters) in double and in contiguous layout, without the weight-derived caches."""
    saved = []
    for m in module.modules():
        drop = {k: m.__dict__.pop(k) for k in [k for k in m.__dict__ if k.startswith("_gq_") and k not in ("_gq_wino", "_gq_wino4")]}
        saved.append((m, drop))
    try:
        t = copy.deepcopy(module)
    finally:
        for m, drop in saved:
            m.__dict__.update(drop)
    return t.double().to(memory_format=torch.contiguous_format).eval()


def d64(x: torch.Tensor) -> torch.Tensor:
    return x.detach().double().contiguous()


def lin_gate(got: torch.Tensor, ref64: torch.Tensor, mag64: torch.Tensor, c: float, what: str = "") -> float:
    """max |got - ref| / mag <= c; returns the fraction of the gate that was used."""
    assert got.shape == ref64.shape, (what, tuple(got.shape), tuple(ref64.shape))
    assert torch.isfinite(got).all(), what
    tiny = 1e-30
    used = float(((got.double() - ref64).abs() / (mag64 + tiny)).max()) / c
    print(f"[gate] {what}: {used:.3f} of c = {c / U24:.0f} u")
    assert used <= 1.0, (what, used)
    return used


def bound_gate(got: torch.Tensor, ref64: torch.Tensor, bound64: torch.Tensor, what: str = "") -> float:
    assert got.shape == ref64.shape and torch.isfinite(got).all(), what
    used = float(((got.double() - ref64).abs() / bound64).max())
    print(f"[gate] {what}: {used:.3f} of the propagated bound")
    assert used <= 1.0, (what, used)
    return used


def conv_ref_and_mag(x64, w64, b64=None, stride=1, padding=1, res64=None):
    ref = F.conv2d(x64, w64, b64, stride, padding)
    mag = F.conv2d(x64.abs(), w64.abs(), None if b64 is None else b64.abs(), stride, padding)
    if res64 is not None:
        ref = ref + res64
        mag = mag + res64.abs()
    return ref, mag


def contract_z(z: torch.Tensor, z64: torch.Tensor, what: str = "") -> float:
    scale = max(1.0, float(z64.abs().max()))
    used = float((z.double() - z64).abs().max()) / (Z_TOL * scale)
    print(f"[contract] {what}: |z - z64| = {used:.3f} of {Z_TOL:g} x {scale:.3g}")
    assert torch.isfinite(z).all() and used <= 1.0, (what, used)
    return used


def contract_x(x: torch.Tensor, x64: torch.Tensor, what: str = "") -> float:
    scale = max(1.0, float(x64.abs().max()))
    used = float((x.double() - x64).abs().max()) / (X_TOL * scale)
    mse = float(((x.double() - x64) ** 2).mean())
    psnr = 10.0 * torch.log10(torch.tensor(4.0 * scale * scale / max(mse, 1e-300))).item()
    print(f"[contract] {what}: |x - x64| = {used:.3f} of {X_TOL:g} x {scale:.3g}; PSNR {psnr:.1f} dB")
    assert torch.isfinite(x).all() and used <= 1.0 and psnr >= PSNR_MIN, (what, used, psnr)
    return used


# ------------------------------------------------------------------------------------------ propagated models
def _group_view(t: torch.Tensor, groups: int):
    b, c = t.shape[0], t.shape[1]
    return t.reshape(b, groups, -1)


def gn_parts(norm: torch.nn.GroupNorm, x64: torch.Tensor):
    """(xhat, rstd per element, mean per element) of GroupNorm(x) in fp64."""
    g = _group_view(x64, norm.num_groups)
    mean = g.mean(2, keepdim=True)
    var = g.var(2, unbiased=False, keepdim=True)
    rstd = (var + norm.eps).rsqrt()
    xhat = ((g - mean) * rstd).reshape(x64.shape)
    return xhat, rstd.expand_as(g).reshape(x64.shape), mean.expand_as(g).reshape(x64.shape)


def silu_lip() -> float:
    return 1.1      # max |d silu / dx| = 1.0998


def gn_own_error(norm: torch.nn.GroupNorm, x64: torch.Tensor, act: bool) -> torch.Tensor:
    """What an fp32 evaluation of [swish](GroupNorm(x)) can lose by itself: the statistics (fp32 partial sums: relative 8u of mean
    and of var), the folded scale / shift, the sigmoid (v_exp + one Newton step: a few ulp)."""
    xhat, rstd, mean = gn_parts(norm, x64)
    gam = norm.weight.detach().abs()[None, :, None, None]
    bet = norm.bias.detach().abs()[None, :, None, None]
    e = 16 * U24 * (gam * ((x64 - mean).abs() * rstd + mean.abs() * rstd + xhat.abs()) + bet)
    return e * (silu_lip() if act else 1.0) + (8 * U24 * (gam * xhat.abs() + bet) if act else 0.0)


def gn_propagate(norm: torch.nn.GroupNorm, x64: torch.Tensor, dx: torch.Tensor, act: bool) -> torch.Tensor:
    """First-order bound on the change of [swish](GroupNorm(x)) for an elementwise perturbation bound dx of x:
    |d xhat| <= rstd (|dx| + mean_g |dx| + |xhat| rms_g(dx))   (mean shift; d rstd / rstd = -rstd mean(xhat dx), Cauchy-Schwarz)."""
    xhat, rstd, _ = gn_parts(norm, x64)
    g = _group_view(dx, norm.num_groups)
    mean_d = g.mean(2, keepdim=True).expand_as(g).reshape(dx.shape)
    rms_d = (g * g).mean(2, keepdim=True).sqrt().expand_as(g).reshape(dx.shape)
    gam = norm.weight.detach().abs()[None, :, None, None]
    d = gam * rstd * (dx + mean_d + xhat.abs() * rms_d)
    return d * (silu_lip() if act else 1.0)


def resnet_ref_and_bound(t, x64: torch.Tensor, pb64=None, c1: float = C_F16X3, c2: float = C_F16X3, cs: float = C_F16X3):
    """fp64 output of a ResnetBlock twin ``t`` (pit/modules/unet.py:137-153 in the reference) and an elementwise first-order bound
    on what an fp32 route with per-convolution coefficients (c1, c2, shortcut cs) may differ by."""
    xin = x64 if pb64 is None else x64 + pb64[None, :, None, None]
    a1 = F.silu(t.norm1(xin))
    da1 = gn_own_error(t.norm1, xin, True)
    h1, m1 = conv_ref_and_mag(a1, t.conv1.weight, t.conv1.bias)
    dh1 = c1 * m1 + F.conv2d(da1, t.conv1.weight.abs(), None, 1, 1)
    a2 = F.silu(t.norm2(h1))
    da2 = gn_propagate(t.norm2, h1, dh1, True) + gn_own_error(t.norm2, h1, True)
    if t.in_channels != t.out_channels:
        xs, ms = conv_ref_and_mag(xin, t.nin_shortcut.weight, t.nin_shortcut.bias, 1, 0)
        dxs = cs * ms
    else:
        xs, dxs = xin, 2 * U24 * xin.abs()
    y, m2 = conv_ref_and_mag(a2, t.conv2.weight, t.conv2.bias, 1, 1, xs)
    dy = c2 * m2 + F.conv2d(da2, t.conv2.weight.abs(), None, 1, 1) + dxs
    return y, SECOND_ORDER * dy


def attn_ref_and_bound(t, x64: torch.Tensor, c_proj: float = C_F16X3, c_gemm: float = C_F16X3):
    """fp64 output of an AttnBlock twin (reference unet.py:185-206) and the propagated first-order bound: q / k / v / proj_out charged
    c_proj of sum |in||w| + |bias|, the two attention GEMMs c_gemm of sum |q||k| resp. sum p|v|, softmax propagated by its derivative
    (|dp_ij| <= p_ij (|ds_ij| + sum_j p_ij |ds_ij|)) and charged 8u (1 + |s_ij - max_i s|) p_ij for its own exp / normalisation."""
    b, c, h, w = x64.shape
    y = t.norm(x64)
    dy = gn_own_error(t.norm, x64, False)
    def proj(conv):
        o, m = conv_ref_and_mag(y, conv.weight, conv.bias, 1, 0)
        d = c_proj * m + F.conv2d(dy, conv.weight.abs(), None, 1, 0)
        tok = lambda z: z.reshape(b, c, h * w).transpose(1, 2)     # [b, L, c]
        return tok(o), tok(d)
    (q, dq), (k, dk), (v, dv) = proj(t.q), proj(t.k), proj(t.v)
    sc = c ** -0.5
    s = torch.bmm(q, k.transpose(1, 2)) * sc
    ds = (torch.bmm(dq, k.abs().transpose(1, 2)) + torch.bmm(q.abs(), dk.transpose(1, 2))
          + c_gemm * torch.bmm(q.abs(), k.abs().transpose(1, 2))) * sc + 2 * U24 * s.abs()
    p = torch.softmax(s, -1)
    own = 8 * U24 * (1.0 + (s - s.max(-1, keepdim=True).values).abs()) * p
    dp = p * (ds + (p * ds).sum(-1, keepdim=True)) + own
    a = torch.bmm(p, v)
    da = torch.bmm(dp, v.abs()) + torch.bmm(p, dv) + c_gemm * torch.bmm(p, v.abs())
    img = lambda z: z.transpose(1, 2).reshape(b, c, h, w)
    out, m = conv_ref_and_mag(img(a), t.proj_out.weight, t.proj_out.bias, 1, 0, x64)
    dout = c_proj * m + F.conv2d(img(da), t.proj_out.weight.abs(), None, 1, 0)
    return out, SECOND_ORDER * dout


def stats_of(y64: torch.Tensor, groups: int = 32) -> torch.Tensor:
    """(sum, sum of squares) per (image, group) of an NCHW fp64 tensor, flattened like _lib.gn_stats_values."""
    g = _group_view(y64, groups)
    return torch.stack([g.sum(2), (g * g).sum(2)], -1).flatten()
