"""What the encoder's convolution routes do to the indices: perturbation of z against the CPU golden and against the
all-MIOpen (fp32 implicit GEMM) encoder, and the number of index flips on bs-16 random batches.  Variants: MIOpen only; the
default (direct fp16 x 3 convolution at the two widest levels, Winograd F(2x2,3x3) below, 1x1 as fp16 x 3 GEMMs); the same with
F(4x4,3x3) at the 512-channel levels and the middle block."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
import bench
from pit_hip.modules import unet

dev = torch.device("cuda:0")
vae = bench.build_model(dev, bench.CONFIGS["gq_0.25"]).to(memory_format=torch.channels_last)
d = np.load(os.path.join(ROOT, "tests", "golden", "g7_full_e2e.npz"))
gx = torch.Generator().manual_seed(1000)
x1 = (torch.rand(1, 3, 256, 256, generator=gx) * 2 - 1).to(dev).contiguous(memory_format=torch.channels_last)
g = torch.Generator().manual_seed(5)
NB = int(os.environ.get("GQ_CHECK_BATCHES", "8"))   # bs-16 batches for the flip statistics
xs = [(torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).to(dev).contiguous(memory_format=torch.channels_last) for _ in range(NB)]
res = {}
with torch.no_grad():
    for name in ("direct", "default", "F(4,3) deep levels"):
        unet.WINOGRAD = unet.DIRECT_CONV = unet.DIRECT_CONV_1X1 = name != "direct"
        if name == "F(4,3) deep levels":
            for part in (vae.encoder.down[2], vae.encoder.down[3], vae.encoder.mid):
                for m in part.modules():
                    if getattr(m, "_gq_wino", False):
                        m._gq_wino4 = True
        z1 = vae.encoder(x1)
        _, i1 = vae.quant(x1)
        z16 = torch.cat([vae.encoder(xb).cpu() for xb in xs])
        i16 = torch.cat([vae.quant(xb)[1].cpu() for xb in xs])
        torch.cuda.synchronize(); import time; t0 = time.perf_counter()
        for _ in range(5): vae.encoder(xs[0])
        torch.cuda.synchronize(); t_enc = (time.perf_counter() - t0) / 5 * 1e3
        res[name] = (z1.cpu(), i1.cpu(), z16, i16, t_enc)
unet.WINOGRAD = unet.DIRECT_CONV = unet.DIRECT_CONV_1X1 = True
zc = torch.from_numpy(d["z_enc"])
for name in res:
    z1, i1 = res[name][0], res[name][1]
    print(f"{name:20s} (encoder {res[name][4]:.1f} ms): max|z - z_cpu| = {float((z1 - zc).abs().max()):.3e}; index mismatches vs CPU golden: "
          f"{int((i1.numpy() != d['indices']).sum())} / {i1.numel()}")
zd = res["direct"][2]
for name in ("default", "F(4,3) deep levels"):
    zw = res[name][2]
    flips = (res["direct"][3] != res[name][3])
    print(f"{NB} x bs16 {name}: max|z - z_direct| = {float((zw - zd).abs().max()):.3e} (mean |z| {float(zd.abs().mean()):.3f}); "
          f"index flips vs direct: {int(flips.sum())} / {flips.numel()}")
