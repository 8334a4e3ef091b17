import os, sys, time, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "vq-vae-from-gaussian-vae_amd"))
from pit_hip import _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(123)
dim, n, rows = 16, 65536, 262144
noise = torch.randn(n, dim, generator=g).to(dev)
mu = torch.randn(rows, dim, generator=g).to(dev)
for name, sd in (("abs(randn)+1e-3", torch.abs(torch.randn(rows, dim, generator=g)) + 1e-3), ("lognormal(-0.75,0.15)", torch.exp(0.5 * (-1.5 + 0.3 * torch.randn(rows, dim, generator=g))))):
    sd = sd.to(dev)
    ws = _lib.Workspace()
    for _ in range(2): _lib.gq_argmax(mu, sd, noise, 1.0, ws=ws)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): _lib.gq_argmax(mu, sd, noise, 1.0, ws=ws)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    _lib.debug_enable(True); _lib.gq_argmax(mu, sd, noise, 1.0, ws=ws); torch.cuda.synchronize()
    fb, rr = _lib.debug_counters(ws); _lib.debug_enable(False)
    print(f"sd={name}: {dt*1e3:.2f} ms per call, fallback rows {fb} ({100*fb/rows:.2f}%), candidates/row {rr/rows:.2f}")
