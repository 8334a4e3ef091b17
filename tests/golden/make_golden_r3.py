"""Round-3 addition to the golden vectors (same rules as make_golden.py / make_golden_r2.py: run in the build container
only, imports /root/reference, stores DATA -- inputs' seed and expected outputs -- never source).

  g14_e2e_8x256.npz   eight 256x256 images (one batch, the way eval.py:144-151 feeds them) through the reference
                      Encoder -> GaussianQuantRegularizer(backend="torch", 65 536 samples, group 16) -> Decoder on CPU:
                      8 192 rows of end-to-end index parity for the GPU path (round 2 had one image = 1 024 rows).
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

from pit.modules.unet import Decoder as RefDecoder, Encoder as RefEncoder  # noqa: E402
from pit.quantization.gaussian import GaussianQuantRegularizer as RefGQ  # noqa: E402

from oracle import gq_oracle as O  # noqa: E402

torch.set_grad_enabled(False)
FULL = dict(attn_type="vanilla", double_z=True, z_channels=16, resolution=256, in_channels=3, out_ch=3, ch=128,
            ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[32], dropout=0.0)
IMAGE_SEED, WEIGHT_SEED, NIMG = 3256, 1234, 8

t0 = time.time()
gx = torch.Generator().manual_seed(IMAGE_SEED)
x = torch.rand(NIMG, 3, 256, 256, generator=gx) * 2 - 1
torch.manual_seed(WEIGHT_SEED)
renc, rdec = RefEncoder(**FULL).eval(), RefDecoder(**FULL).eval()
ze = renc(x)
assert tuple(ze.shape) == (NIMG, 32, 32, 32)
print(f"encoder {time.time() - t0:.0f}s")
ref = RefGQ("bchw", 65536, group=16, backend="torch").eval()
zh, info = ref(ze)
print(f"quantiser {time.time() - t0:.0f}s")
xr = rdec(zh)
print(f"decoder {time.time() - t0:.0f}s")

# the oracle on the very rows the reference saw -> index parity + top-2 gaps
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
path = os.path.join(HERE, "g14_e2e_8x256.npz")
np.savez_compressed(path, z_enc=ze.numpy(), indices=info["indices"].numpy().astype(np.int32), gap=gap,
                    x_rec=xr.numpy().astype(np.float16),
                    x_rec_stats=np.array([float(xr.mean()), float(xr.std()), float(xr.abs().max())], np.float64))
print(f"wrote g14_e2e_8x256.npz ({os.path.getsize(path) / 1024:.0f} KiB): {oi.size} rows, min gap {float(gap.min()):.2e}, "
      f"rows with gap < 1e-3: {int((gap < 1e-3).sum())} ({time.time() - t0:.0f}s)")

meta_path = os.path.join(HERE, "meta.json")
meta = json.load(open(meta_path))
meta["cases_r3"] = {"G14": {"image_seed": IMAGE_SEED, "weights_seed": WEIGHT_SEED, "images": NIMG, "size": 256}}
with open(meta_path, "w") as f:
    json.dump(meta, f, indent=1)
print("round-3 goldens written")
