"""Shared by the -m gpu test files: golden loader, the bench configuration's UNet parameters, engine builders, row order."""
import json
import math
import os
import numpy as np
import pytest
import torch
from oracle import gq_oracle as O
import convstack_ref as R
from ckpt_like import apply_conv_out_calibration_, checkpoint_like_

G = os.path.join(os.path.dirname(__file__), "golden")


META = json.load(open(os.path.join(G, "meta.json")))


DEV = "cuda:0"


FULL = dict(attn_type="vanilla", double_z=True, z_channels=16, resolution=256, in_channels=3, out_ch=3, ch=128,
            ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[32], dropout=0.0)


def load(name):
    return np.load(os.path.join(G, name))


def _rows(ind):   # [B, K, h, w] -> rows (b, l, k)
    return np.asarray(ind).transpose(0, 2, 3, 1).reshape(-1)


def _stv(stats):
    from pit_hip import _lib

    return _lib.gn_stats_values(stats)


def _psnr(a, b):
    mse = float(((a - b) ** 2).mean())
    return 10 * np.log10(4.0 / max(mse, 1e-20))


def _engine(reg_target="pit.quantization.gaussian.GaussianQuantRegularizer", reg_params=None, unet=FULL, seed=1234):
    from pit_hip.models.autoencoder import AutoencodingEngine

    torch.manual_seed(seed)
    reg_params = reg_params or {"format": "bchw", "group": 16, "n_samples": 65536, "backend": "hip"}
    return AutoencodingEngine(encoder_config={"target": "pit.modules.unet.Encoder", "params": unet},
                              decoder_config={"target": "pit.modules.unet.Decoder", "params": unet},
                              regularizer_config={"target": reg_target, "params": reg_params}).eval()


# ------------------------------------------------------------------------------------------ configs[4]: 512 x 512
def _x512():
    gx = torch.Generator().manual_seed(1512)
    return torch.rand(1, 3, 512, 512, generator=gx) * 2 - 1


# ------------------------------------------------------------------------------------------ checkpoint-shaped weights
_checkpoint_like_ = checkpoint_like_      # (shared with tests/golden/make_golden_r4.py)


def _trained_like_engine(d):
    from pit_hip.models.autoencoder import AutoencodingEngine

    torch.manual_seed(1234)
    vae = AutoencodingEngine(encoder_config={"target": "pit.modules.unet.Encoder", "params": FULL},
                             decoder_config={"target": "pit.modules.unet.Decoder", "params": FULL},
                             regularizer_config={"target": "pit.quantization.gaussian.GaussianQuantRegularizer",
                                                 "params": {"format": "bchw", "group": 16, "n_samples": 65536, "backend": "hip"}}).eval()
    checkpoint_like_(vae.encoder, 5)
    checkpoint_like_(vae.decoder, 6)
    apply_conv_out_calibration_(vae.encoder.conv_out, torch.from_numpy(d["conv_out_scale"]), torch.from_numpy(d["conv_out_shift"]))
    return vae


def _e2e_vs_golden(d, x, channels_last, filt, tag):
    """Gates = bench.GATES (the ONE definition): the golden z through the GPU quantiser -> the reference's indices except where
    its own top-2 gap is below the libm difference; end to end (GPU encoder in front): |dz| inside the gate, at most 2 per 1024
    indices differing and only at near-ties of the reference's own score; decoder: reconstruction of the images whose tokens all
    agree within the fp16 golden's resolution at this output scale."""
    from bench import GATES
    from pit_hip import _lib

    nimg = x.shape[0]
    per = d["indices"].size // nimg
    vae = _trained_like_engine(d).to(DEV)
    x = x.to(DEV)
    if channels_last:
        vae = vae.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)
    _lib.set_filter(filt)
    try:
        want, gap = _rows(d["indices"]), d["gap"]
        # (1) the reference's own z through the GPU quantiser: the bit-exact contract at the module boundary
        zhat, info = vae.regularization(torch.from_numpy(d["z_enc"]).to(DEV))
        diff2 = _rows(info["indices"].cpu().numpy()) != want
        assert diff2.sum() == 0 or np.all(gap[diff2] < GATES["same_z_gap"]), (int(diff2.sum()), gap[diff2])
        # (2) end to end
        with torch.no_grad():
            z_enc = vae.encode(x, unregularized=True)[0]
            z, ind = vae.quant(x)
            rec = vae.dequant(ind)
    finally:
        _lib.set_filter("auto")
    zr = torch.from_numpy(d["z_enc"])
    dz = float((z_enc.cpu() - zr).abs().max())
    got = _rows(ind.cpu().numpy())
    diff = got != want
    print(f"{tag} (channels_last={channels_last}, filter {filt}): |dz| {dz:.2e} (|z| max {float(zr.abs().max()):.2f}), "
          f"{int(diff2.sum())} of {want.size} indices differ on the golden z, {int(diff.sum())} end to end"
          f"{' at gaps ' + str(gap[diff]) if diff.any() else ''}; smallest golden gap {float(gap.min()):.2e}")
    assert dz <= GATES["z_enc_max_abs"], dz
    per_image = diff.reshape(nimg, per).sum(1)
    assert per_image.max() <= GATES["indices_differing_per_1024"] and np.all(gap[diff] < GATES["near_tie_gap"]), (per_image, gap[diff])
    ref = torch.from_numpy(d["x_rec"].astype(np.float32))
    same = ~diff.reshape(nimg, per).any(1)
    scale = float(ref.abs().max())                     # |x_rec| reaches ~6.5 with these weights: fp16 ulp of the golden 3.9e-3 there
    if same.any():
        err = float((rec.cpu()[same] - ref[same]).abs().max())
        assert err <= GATES["recon_max_abs_if_indices_equal"] * max(1.0, scale), (err, scale)


# ------------------------------------------------------------------------------------------ 16-bit split-relative ids in the records
_BIG_N_SCRIPT = r"""
import os, sys, json
import numpy as np, torch
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "vq-vae-from-gaussian-vae_amd"))
from pit_hip import _lib
from oracle import gq_oracle as O
dim, n, rows, filt = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
_lib.set_filter(filt)
g = torch.Generator().manual_seed(dim + n)
mu = 0.9 * torch.randn(rows, dim, generator=g)
sd = torch.exp(0.5 * (-1.5 + 0.3 * torch.randn(rows, dim, generator=g)))
cb = torch.randn(n, dim, generator=g)
mu[: rows // 2] = cb[-(rows // 2):]      # winners in the LAST groups of the codebook: rows sitting on those codes, small sigma
sd[: rows // 2] = 0.02
dev = torch.device("cuda:0")
ws = _lib.Workspace()
idx, _ = _lib.gq_argmax(mu.to(dev), sd.to(dev), cb.to(dev), 1.0, ws=ws)
torch.cuda.synchronize()
pl = _lib.debug_plan(rows, n, dim)
sel = np.r_[0:8, rows - 8:rows]
ref, _ = O.argmax_rows(mu.numpy()[sel], sd.numpy()[sel], cb.numpy(), 1.0)
got = idx.cpu().numpy()[sel]
print(json.dumps({"plan": pl, "equal": bool(np.array_equal(got, ref)), "max_index": int(got.max()), "n": n}))
"""
