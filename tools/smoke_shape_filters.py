import os, sys, torch, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/vq-vae-from-gaussian-vae_amd")
from pit_hip import _lib
dev = torch.device("cuda:0")
dim, n, rows = 16, 65536, 1024 * 4 * 8 * 8
g = torch.Generator().manual_seed(123)
noise = torch.randn(n, dim, generator=g).to(dev)
mu = torch.randn(rows, dim, generator=g).to(dev)
sd = (torch.abs(torch.randn(rows, dim, generator=g)) + 1e-3).to(dev)
for filt in ("auto", "fp32"):
    _lib.set_filter(filt)
    ws = _lib.Workspace()
    _lib.debug_enable(True)
    _lib.gq_argmax(mu, sd, noise, 1.0, ws=ws); torch.cuda.synchronize()
    fb, rr = _lib.debug_counters(ws)
    _lib.debug_enable(False)
    t0 = time.perf_counter()
    for _ in range(5):
        _lib.gq_argmax(mu, sd, noise, 1.0, ws=ws)
    torch.cuda.synchronize()
    print(f"smoke-loop shape ({rows} rows, sd=|randn|) filter={filt}: {(time.perf_counter()-t0)/5*1e3:.2f} ms, undecided rows (in-block scan) {fb} ({100*fb/rows:.1f} %), candidates/row {rr/rows:.2f}")
