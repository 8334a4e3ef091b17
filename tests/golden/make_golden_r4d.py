"""Round-4 addition (same rules as make_golden*.py: run in the build container only, imports /root/reference, stores DATA -- seeds,
a per-channel calibration and expected outputs -- never source).

  g18_e2e_16x256_trained_like.npz   the BENCH configuration itself against the reference: SIXTEEN 256x256 images (BASELINE configs[1]:
                      bs 16 per GPU) through the reference Encoder -> GaussianQuantRegularizer(backend="torch", 65 536 samples, group 16)
                      -> Decoder on CPU with checkpoint-like weights (tests/ckpt_like.py) and z at the trained operating point, and the
                      reference's per-image PSNR (eval.py:165-169: get_psnr(x, x_rec, zero_mean=True), pit/evaluations/psnr.py:17-28).
                      Small by construction: indices, the reference's own top-2 gaps, PSNR per image and three moments of every
                      reconstruction -- no tensors (g15 / g17 hold z and the reconstructions of this recipe at 2 / 3 images).
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from pit.evaluations.psnr import get_psnr  # noqa: E402
from pit.modules.unet import Decoder as RefDecoder, Encoder as RefEncoder  # noqa: E402
from pit.quantization.gaussian import GaussianQuantRegularizer as RefGQ  # noqa: E402

from ckpt_like import apply_conv_out_calibration_, checkpoint_like_, operating_point_calibration  # noqa: E402
from oracle import gq_oracle as O  # noqa: E402

torch.set_grad_enabled(False)
FULL = dict(attn_type="vanilla", double_z=True, z_channels=16, resolution=256, in_channels=3, out_ch=3, ch=128,
            ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[32], dropout=0.0)
IMAGE_SEED, WEIGHT_SEED, ENC_RECIPE_SEED, DEC_RECIPE_SEED, NIMG = 4258, 1234, 5, 6, 16

t0 = time.time()
gx = torch.Generator().manual_seed(IMAGE_SEED)
x = torch.rand(NIMG, 3, 256, 256, generator=gx) * 2 - 1
torch.manual_seed(WEIGHT_SEED)
renc, rdec = RefEncoder(**FULL).eval(), RefDecoder(**FULL).eval()
checkpoint_like_(renc, ENC_RECIPE_SEED)
checkpoint_like_(rdec, DEC_RECIPE_SEED)
z0 = renc(x)
scale, shift = operating_point_calibration(z0, 16)
apply_conv_out_calibration_(renc.conv_out, scale, shift)
ze = renc(x)
print(f"encoder {time.time() - t0:.0f}s: mu std {float(ze[:, :16].std()):.3f}; logvar mean {float(ze[:, 16:].mean()):+.3f}")
ref = RefGQ("bchw", 65536, group=16, backend="torch").eval()
zh, info = ref(ze)
print(f"quantiser {time.time() - t0:.0f}s")
xr = torch.cat([rdec(zh[i:i + 4]) for i in range(0, NIMG, 4)])
psnr = get_psnr(x, xr, zero_mean=True)
print(f"decoder {time.time() - t0:.0f}s; PSNR {psnr.numpy().round(3)}")

b_, c2, h, w = ze.shape
zf = ze.reshape(b_, c2, h * w).transpose(1, 2)
mu, lv = zf.chunk(2, 2)
std = torch.exp(0.5 * torch.clamp(lv, -30.0, 20.0))
mu_r, std_r = mu.reshape(-1, 16).contiguous(), std.reshape(-1, 16).contiguous()
oi, _, best, second = O.argmax_rows(mu_r.numpy(), std_r.numpy(), ref.prior_samples.numpy(), 1.0, logstd=std_r.log().numpy(),
                                    with_gap=True)
want = info["indices"].permute(0, 2, 3, 1).reshape(-1).numpy()
assert np.array_equal(oi, want), "oracle != reference"
gap = (best - second).astype(np.float32)
moments = np.stack([xr.mean(dim=(1, 2, 3)).numpy(), xr.std(dim=(1, 2, 3)).numpy(), xr.abs().amax(dim=(1, 2, 3)).numpy()], 1).astype(np.float64)
zmom = np.array([float(ze.abs().max()), float(ze[:, :16].std()), float(ze[:, 16:].mean())])
path = os.path.join(HERE, "g18_e2e_16x256_trained_like.npz")
np.savez_compressed(path, indices=info["indices"].numpy().astype(np.uint16), gap=gap, psnr=psnr.numpy().astype(np.float32),
                    x_rec_moments=moments, z_moments=zmom, conv_out_scale=scale.numpy(), conv_out_shift=shift.numpy())
print(f"wrote {os.path.basename(path)} ({os.path.getsize(path) / 1024:.0f} KiB): {oi.size} rows, min gap {float(gap.min()):.2e}, "
      f"rows with gap < 2e-5: {int((gap < 2e-5).sum())} ({time.time() - t0:.0f}s)")

meta_path = os.path.join(HERE, "meta.json")
meta = json.load(open(meta_path))
meta["cases_r4"]["G18"] = {"image_seed": IMAGE_SEED, "weights_seed": WEIGHT_SEED, "encoder_recipe_seed": ENC_RECIPE_SEED,
                           "decoder_recipe_seed": DEC_RECIPE_SEED, "images": NIMG, "size": 256}
with open(meta_path, "w") as f:
    json.dump(meta, f, indent=1)
