"""Round-4 addition (same rules as make_golden*.py: run in the build container only, imports /root/reference, stores DATA -- seeds,
a per-channel calibration and expected outputs -- never source).

  g17_e2e_nonsquare_trained_like.npz   THREE 192x320 images (odd batch, non-square: a 24x40 latent, 960 rows per image -- not a
                      multiple of the filter's 512-row blocks, 2880 rows in all) through the reference Encoder ->
                      GaussianQuantRegularizer(backend="torch", 65 536 samples, group 16) -> Decoder on CPU, checkpoint-like
                      weights (tests/ckpt_like.py) and conv_out calibrated to the trained operating point, as g15.  The only
                      end-to-end golden that is neither square nor an even batch: the GroupNorm statistics, the attention over 960
                      tokens, the Winograd tile edges (24 and 40 are not multiples of the 4x4 / 6x6 output tiles' super-tiles at
                      every level) and the ragged last row block of the quantiser all meet the REFERENCE's values here.
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

from pit.modules.unet import Decoder as RefDecoder, Encoder as RefEncoder  # noqa: E402
from pit.quantization.gaussian import GaussianQuantRegularizer as RefGQ  # noqa: E402

from ckpt_like import apply_conv_out_calibration_, checkpoint_like_, operating_point_calibration  # noqa: E402
from oracle import gq_oracle as O  # noqa: E402

torch.set_grad_enabled(False)
FULL = dict(attn_type="vanilla", double_z=True, z_channels=16, resolution=256, in_channels=3, out_ch=3, ch=128,
            ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[32], dropout=0.0)
IMAGE_SEED, WEIGHT_SEED, ENC_RECIPE_SEED, DEC_RECIPE_SEED, NIMG, H, W = 4257, 1234, 5, 6, 3, 192, 320

t0 = time.time()
gx = torch.Generator().manual_seed(IMAGE_SEED)
x = torch.rand(NIMG, 3, H, W, generator=gx) * 2 - 1
torch.manual_seed(WEIGHT_SEED)
renc, rdec = RefEncoder(**FULL).eval(), RefDecoder(**FULL).eval()
checkpoint_like_(renc, ENC_RECIPE_SEED)
checkpoint_like_(rdec, DEC_RECIPE_SEED)
z0 = renc(x)
scale, shift = operating_point_calibration(z0, 16)
apply_conv_out_calibration_(renc.conv_out, scale, shift)
ze = renc(x)
assert tuple(ze.shape) == (NIMG, 32, H // 8, W // 8)
print(f"encoder {time.time() - t0:.0f}s: mu std {float(ze[:, :16].std()):.3f}; logvar mean {float(ze[:, 16:].mean()):+.3f}")
ref = RefGQ("bchw", 65536, group=16, backend="torch").eval()
zh, info = ref(ze)
xr = rdec(zh)
print(f"quantiser + decoder {time.time() - t0:.0f}s")

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
kl_bits = float((0.5 * (mu_r ** 2 + std_r ** 2 - 1.0 - 2.0 * std_r.log())).sum(1).mean() / np.log(2.0))
path = os.path.join(HERE, "g17_e2e_nonsquare_trained_like.npz")
np.savez_compressed(path, z_enc=ze.numpy(), indices=info["indices"].numpy().astype(np.int32), gap=gap,
                    x_rec=xr.numpy().astype(np.float16), conv_out_scale=scale.numpy(), conv_out_shift=shift.numpy())
print(f"wrote {os.path.basename(path)} ({os.path.getsize(path) / 1024:.0f} KiB): {oi.size} rows, {kl_bits:.1f} bits / group, min gap "
      f"{float(gap.min()):.2e}, |x_rec| max {float(xr.abs().max()):.3g} ({time.time() - t0:.0f}s)")

meta_path = os.path.join(HERE, "meta.json")
meta = json.load(open(meta_path))
meta["cases_r4"]["G17"] = {"image_seed": IMAGE_SEED, "weights_seed": WEIGHT_SEED, "encoder_recipe_seed": ENC_RECIPE_SEED,
                           "decoder_recipe_seed": DEC_RECIPE_SEED, "images": NIMG, "height": H, "width": W,
                           "kl_bits_per_group": round(kl_bits, 2)}
with open(meta_path, "w") as f:
    json.dump(meta, f, indent=1)
