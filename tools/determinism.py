import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/vq-vae-from-gaussian-vae_amd")
import bench
from pit_hip.modules import unet
dev = torch.device("cuda:0")
vae = bench.build_model(dev, bench.CONFIGS["gq_0.25"]).to(memory_format=torch.channels_last)
for B in (1, 4, 16):
    x = (torch.rand(B, 3, 256, 256) * 2 - 1).to(dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for wino in (True, False):
            unet.WINOGRAD = wino; unet.SUBPIXEL_UPCONV = wino
            z1 = vae.encoder(x); z2 = vae.encoder(x)
            zh, info = vae.regularization(z1)
            r1 = vae.decode(zh); r2 = vae.decode(zh)
            print(f"B={B} winograd/subpixel={wino}: encoder run-to-run equal {torch.equal(z1, z2)} (max diff {float((z1-z2).abs().max()):.2e}), "
                  f"decoder equal {torch.equal(r1, r2)} (max diff {float((r1-r2).abs().max()):.2e})")
