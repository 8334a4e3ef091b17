"""One encode->quantize->decode step repeated a few times (for rocprofv3 --kernel-trace --stats)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
import bench
from pit_hip.modules import unet as U
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    setattr(U, k, v == "1")
dev = torch.device("cuda:0")
vae = bench.build_model(dev, bench.CONFIGS["gq_0.25"]).to(memory_format=torch.channels_last)
g = torch.Generator().manual_seed(1000)
x = (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).to(dev).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(6):
        zh, info = vae.encode(x, return_reg_log=True)
        rec = vae.decode(zh)
torch.cuda.synchronize()
