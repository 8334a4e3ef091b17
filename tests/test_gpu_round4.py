"""-m gpu, round 4: end-to-end index parity at a TRAINED operating point (golden g15, tests/golden/make_golden_r4.py: the reference's
CPU path with checkpoint-like weights and z at ~17.8 bits per group -- the regime eval.py:112-116 runs in), and the pieces
round 4 added to the quantiser call."""
import os

import numpy as np
import pytest
import torch

from ckpt_like import apply_conv_out_calibration_, checkpoint_like_

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"
FULL = dict(attn_type="vanilla", double_z=True, z_channels=16, resolution=256, in_channels=3, out_ch=3, ch=128,
            ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[32], dropout=0.0)


def _rows(ind):   # [B, K, h, w] -> rows (b, l, k)
    return np.asarray(ind).transpose(0, 2, 3, 1).reshape(-1)


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


@pytest.mark.parametrize("channels_last", [True, False])
@pytest.mark.parametrize("filt", ["auto", "bf16", "fp32", "mixed"])
def test_g15_trained_operating_point_end_to_end_vs_reference_golden(channels_last, filt):
    """VERDICT r3 missing #3 / next #1d.  Gates = bench.GATES (the ONE definition): the golden z through the GPU quantiser ->
    the reference's indices except where its own top-2 gap is below the libm difference; end to end (GPU encoder in front):
    |dz| inside the gate, at most 2 per 1024 indices differing and only at near-ties of the reference's own score; decoder:
    reconstruction of the images whose tokens all agree within the fp16 golden's resolution at this output scale."""
    from bench import GATES
    from pit_hip import _lib

    d = np.load(os.path.join(G, "g15_e2e_trained_like.npz"))
    vae = _trained_like_engine(d).to(DEV)
    gx = torch.Generator().manual_seed(4256)
    x = (torch.rand(2, 3, 256, 256, generator=gx) * 2 - 1).to(DEV)
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
    print(f"g15 trained-like e2e (channels_last={channels_last}, filter {filt}): |dz| {dz:.2e} (|z| max {float(zr.abs().max()):.2f}), "
          f"{int(diff2.sum())} of 2048 indices differ on the golden z, {int(diff.sum())} end to end"
          f"{' at gaps ' + str(gap[diff]) if diff.any() else ''}; smallest golden gap {float(gap.min()):.2e}")
    assert dz <= GATES["z_enc_max_abs"], dz
    per_image = diff.reshape(2, 1024).sum(1)
    assert per_image.max() <= GATES["indices_differing_per_1024"] and np.all(gap[diff] < GATES["near_tie_gap"]), (per_image, gap[diff])
    ref = torch.from_numpy(d["x_rec"].astype(np.float32))
    same = ~diff.reshape(2, 1024).any(1)
    scale = float(ref.abs().max())                     # |x_rec| reaches ~6.5 with these weights: fp16 ulp of the golden 3.9e-3 there
    if same.any():
        err = float((rec.cpu()[same] - ref[same]).abs().max())
        assert err <= GATES["recon_max_abs_if_indices_equal"] * max(1.0, scale), (err, scale)
