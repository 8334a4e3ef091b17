"""512x512 and odd-sized inputs through the channels_last (Winograd / sub-pixel) stack vs the NCHW (MIOpen direct) one."""
import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/vq-vae-from-gaussian-vae_amd")
import bench
dev = torch.device("cuda:0")
vae = bench.build_model(dev, bench.CONFIGS["gq_0.25"])
x = (torch.rand(2, 3, 512, 512) * 2 - 1).to(dev)
with torch.no_grad():
    z0 = vae.encoder(x); zh0, i0 = vae.regularization(z0); r0 = vae.decode(zh0)
    vae = vae.to(memory_format=torch.channels_last)
    z1 = vae.encoder(x); zh1, i1 = vae.regularization(z1); r1 = vae.decode(zh0.contiguous(memory_format=torch.channels_last))
print("512x512: z", tuple(z1.shape), "max|z_cl - z_nchw|", float((z1 - z0).abs().max()), "index mismatches", int((i0["indices"] != i1["indices"]).sum()), "/", i0["indices"].numel(),
      "recon max diff", float((r1 - r0).abs().max()), "peak mem GB", torch.cuda.max_memory_allocated() / 1e9)
x = (torch.rand(3, 3, 250, 198) * 2 - 1).to(dev)   # odd sizes: fallbacks
with torch.no_grad():
    r = vae.decode(vae.regularization(vae.encoder(x))[0])
print("odd size ok:", tuple(r.shape))
