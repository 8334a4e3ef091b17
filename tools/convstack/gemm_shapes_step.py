"""Every library GEMM of one encode -> quantise -> decode step at the bench shape: operand shapes, dtype, time per call
(events around each call; one step), executed TFLOP/s and bytes moved -- the list a hand-written GEMM would have to beat."""
import os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
import bench
dev = torch.device("cuda:0")
vae = bench.build_model(dev, bench.CONFIGS["gq_0.25"]).to(memory_format=torch.channels_last)
g = torch.Generator().manual_seed(1000)
x = (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).to(dev).contiguous(memory_format=torch.channels_last)
def run():
    with torch.no_grad():
        z = vae.encoder(x); zh, info = vae.regularization(z); return vae.decode(zh)
for _ in range(3): run()
torch.cuda.synchronize()
log = []
def wrap(name):
    orig = getattr(torch, name)
    def f(*a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); out = orig(*a, **k); e.record()
        ts = [t for t in a if isinstance(t, torch.Tensor)]
        log.append((name, tuple(tuple(t.shape) for t in ts), str(ts[-1].dtype).replace("torch.", ""), s, e, out.numel() * out.element_size(),
                    sum(t.numel() * t.element_size() for t in ts[-2:])))
        return out
    setattr(torch, name, f)
for n in ("bmm", "mm", "matmul", "addmm"): wrap(n)
run(); torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, shapes, dt, s, e, ob, ib in log:
    A, Bm = shapes[-2], shapes[-1]
    flops = 2.0 * Bm[-1] * (torch.tensor(A).prod().item())
    k = (name, A, Bm, dt)
    c = agg.setdefault(k, [0, 0.0, flops, ob + ib])
    c[0] += 1; c[1] += s.elapsed_time(e)
tot = 0.0
for (name, A, Bm, dt), (n, ms, flops, byts) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += ms
    print(f"{ms:7.3f} ms  x{n:2d}  {name:6s} {dt:8s} {str(A):24s} x {str(Bm):20s} {ms/n*1e3:7.0f} us/call  {flops/(ms/n)/1e9:7.0f} TFLOP/s executed  {byts/(ms/n)/1e9:5.2f} TB/s")
print(f"total {tot:.3f} ms in {len(log)} GEMM calls")
