"""Round-4 addition to the golden vectors (same rules as make_golden*.py: run in the build container only, imports
/root/reference, stores DATA -- seeds, a per-channel calibration and expected outputs -- never source).

  g15_e2e_trained_like.npz   two 256x256 images through the reference Encoder -> GaussianQuantRegularizer(backend="torch",
                      65 536 samples, group 16) -> Decoder on CPU with CHECKPOINT-LIKE weights (tests/ckpt_like.py: GroupNorm
                      gamma over two decades, beta of a few units, conv gains over three decades) and the encoder's conv_out
                      calibrated so that z sits at the quantiser's TRAINED operating point (mu ~ 0.9 N(0,1), logvar ~ -1.5 +- 0.3:
                      about 16 bits per group, sigma ~ 0.47 -- the regime eval.py:112-116 runs trained checkpoints in; every
                      earlier end-to-end golden used seeded-random weights, logvar ~ 0).
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
IMAGE_SEED, WEIGHT_SEED, ENC_RECIPE_SEED, DEC_RECIPE_SEED, NIMG = 4256, 1234, 5, 6, 2

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
assert tuple(ze.shape) == (NIMG, 32, 32, 32)
mu_s, lv_s = ze[:, :16], ze[:, 16:]
print(f"encoder {time.time() - t0:.0f}s: mu mean {float(mu_s.mean()):+.3f} std {float(mu_s.std()):.3f}; logvar mean {float(lv_s.mean()):+.3f} "
      f"std {float(lv_s.std()):.3f}")
ref = RefGQ("bchw", 65536, group=16, backend="torch").eval()
zh, info = ref(ze)
print(f"quantiser {time.time() - t0:.0f}s")
xr = rdec(zh)
print(f"decoder {time.time() - t0:.0f}s")

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
# bits per group at this operating point (gaussian.py:89-93: KL to the prior in bits), for the record
kl_bits = float((0.5 * (mu_r ** 2 + std_r ** 2 - 1.0 - 2.0 * std_r.log())).sum(1).mean() / np.log(2.0))
path = os.path.join(HERE, "g15_e2e_trained_like.npz")
np.savez_compressed(path, z_enc=ze.numpy(), indices=info["indices"].numpy().astype(np.int32), gap=gap,
                    x_rec=xr.numpy().astype(np.float16), conv_out_scale=scale.numpy(), conv_out_shift=shift.numpy(),
                    x_rec_stats=np.array([float(xr.mean()), float(xr.std()), float(xr.abs().max())], np.float64))
print(f"wrote g15_e2e_trained_like.npz ({os.path.getsize(path) / 1024:.0f} KiB): {oi.size} rows, {kl_bits:.1f} bits / group, min gap "
      f"{float(gap.min()):.2e}, rows with gap < 1e-3: {int((gap < 1e-3).sum())}, |x_rec| max {float(xr.abs().max()):.3g} ({time.time() - t0:.0f}s)")

meta_path = os.path.join(HERE, "meta.json")
meta = json.load(open(meta_path))
meta["cases_r4"] = {"G15": {"image_seed": IMAGE_SEED, "weights_seed": WEIGHT_SEED, "encoder_recipe_seed": ENC_RECIPE_SEED,
                            "decoder_recipe_seed": DEC_RECIPE_SEED, "images": NIMG, "size": 256, "kl_bits_per_group": round(kl_bits, 2),
                            "recipe": "tests/ckpt_like.py: checkpoint_like_ on the seeded init, then conv_out calibrated with the stored "
                                      "per-channel (scale, shift)"}}
with open(meta_path, "w") as f:
    json.dump(meta, f, indent=1)
print("round-4 goldens written")
