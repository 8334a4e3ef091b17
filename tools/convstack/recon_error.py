"""Reconstruction error of the GPU path vs the reference's CPU end-to-end golden (tests/golden/g7_full_e2e.npz)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
import bench
dev = torch.device("cuda:0")
d = np.load(os.path.join(ROOT, "tests", "golden", "g7_full_e2e.npz"))
gx = torch.Generator().manual_seed(1000)
x = (torch.rand(1, 3, 256, 256, generator=gx) * 2 - 1).to(dev)
ref = torch.from_numpy(d["x_rec"].astype(np.float32))
for cl in (0, 1):
    vae = bench.build_model(dev, bench.CONFIGS["gq_0.25"])
    xx = x
    if cl:
        vae = vae.to(memory_format=torch.channels_last); xx = x.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        z_enc = vae.encode(xx, unregularized=True)[0]
        z, ind = vae.quant(xx)
        rec = vae.dequant(ind)
    same = (ind.cpu().numpy().astype(np.int32) == d["indices"]).mean()
    err = (rec.cpu() - ref).abs()
    mse = float(((rec.cpu() - ref) ** 2).mean())
    print(f"channels_last={cl}: |z_enc - cpu| max {float((z_enc.cpu() - torch.from_numpy(d['z_enc'])).abs().max()):.2e}; "
          f"indices equal {same * 100:.2f}%; recon max-abs-err {float(err.max()):.2e} (incl. fp16 storage of the golden), "
          f"mean {float(err.mean()):.2e}, PSNR vs CPU recon {10 * np.log10(4.0 / mse):.1f} dB")
