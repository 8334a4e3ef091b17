"""Test infrastructure: parameters of an Encoder / Decoder re-drawn the way trained SD-VAE stacks look, and a calibration of the
encoder's last convolution that puts z at the trained operating point of the reference's quantiser (about 16 bits per 16-dim group:
mu ~ 0.9 N(0, 1), logvar ~ -1.5 +- 0.3).  Shared by tests/golden/make_golden_r4.py (reference, CPU) and the -m gpu
tests, so both build bit-identical weights from (seed, recipe, the stored per-channel calibration)."""
import numpy as np
import torch


def checkpoint_like_(module, seed):
    """GroupNorm gamma in [0.05, 8] (log-uniform, a few at the ends), beta in +-3, conv weights with a 10^3 dynamic range across
    output channels (per-channel log-uniform gains on the default init) and a few large biases."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in module.modules():
            if isinstance(m, torch.nn.GroupNorm):
                gam = torch.exp(torch.rand(m.weight.shape, generator=g) * (np.log(8.0) - np.log(0.05)) + np.log(0.05))
                gam[0], gam[-1] = 8.0, 0.05
                m.weight.copy_(gam.to(m.weight.device))
                m.bias.copy_(((torch.rand(m.bias.shape, generator=g) * 2 - 1) * 3.0).to(m.bias.device))
            elif isinstance(m, torch.nn.Conv2d):
                gain = torch.exp((torch.rand(m.weight.shape[0], generator=g) * 2 - 1) * np.log(1000.0) / 2)   # 1e-1.5 .. 1e1.5
                gain = gain / gain.mean()
                # keep the layer's overall scale near the init's (a trained net is not exploding): normalise the RMS gain
                gain = gain / float((gain ** 2).mean().sqrt())
                m.weight.mul_(gain.to(m.weight.device)[:, None, None, None])
                if m.bias is not None:
                    m.bias.copy_((torch.randn(m.bias.shape, generator=g) * 0.3).to(m.bias.device))


def operating_point_calibration(z, c):
    """Per-channel (scale, shift) that maps the measured z [B, 2c, h, w] to mu ~ (0, 0.9), logvar ~ (-1.5, 0.3)."""
    m = z.mean(dim=(0, 2, 3)).double()
    s = z.std(dim=(0, 2, 3)).double()
    target_std = torch.cat([torch.full((c,), 0.9), torch.full((c,), 0.3)]).double()
    target_mean = torch.cat([torch.zeros(c), torch.full((c,), -1.5)]).double()
    scale = target_std / s
    shift = target_mean - m * scale
    return scale.float(), shift.float()


def apply_conv_out_calibration_(conv_out, scale, shift):
    """z' = z * scale + shift folded into the convolution (fp32 elementwise: the same bits on every host)."""
    with torch.no_grad():
        sc = scale.to(conv_out.weight.device, torch.float32)
        sh = shift.to(conv_out.weight.device, torch.float32)
        conv_out.weight.mul_(sc[:, None, None, None])
        conv_out.bias.mul_(sc).add_(sh)
