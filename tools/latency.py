"""Single-request latency of encode -> quantize -> decode: eager launches vs one HIP-graph replay."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
import bench
from pit_hip.graphed import GraphedAutoencoder

dev = torch.device("cuda:0")
vae = bench.build_model(dev, bench.CONFIGS["gq_0.25"]).to(memory_format=torch.channels_last)
for B in (1, 4, 16):
    x = (torch.rand(B, 3, 256, 256) * 2 - 1).to(dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for _ in range(3):
            z, ind = vae.quant(x); rec = vae.decode(z)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            z, ind = vae.quant(x); rec = vae.decode(z)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 10
    gvae = GraphedAutoencoder(vae, x)
    rec_g, ind_g = gvae(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        rec_g, ind_g = gvae(x)
    torch.cuda.synchronize()
    graphed = (time.perf_counter() - t0) / 10
    same = torch.equal(ind_g, ind) and torch.allclose(rec_g, rec, atol=1e-5)
    print(f"B={B}: eager {eager*1e3:.2f} ms, HIP graph {graphed*1e3:.2f} ms ({eager/graphed:.2f}x), outputs equal: {same}")
