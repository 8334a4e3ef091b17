"""Do hipBLASLt's other solutions beat its heuristic's first pick on the step's fp16 -> fp32 GEMM shapes?  Times torch.bmm /
torch.mm (out_dtype fp32) with PyTorch's TunableOp off and on (tuning enabled: every solution is benchmarked on first use)."""
import os, sys, time, torch
import torch.cuda.tunable as T
dev = torch.device("cuda:0")
SHAPES = [("bmm", 36, 16384, 768, 256), ("bmm", 36, 4096, 1536, 512), ("bmm", 16, 16384, 1536, 512), ("bmm", 16, 4096, 1536, 512),
          ("bmm", 36, 1024, 1536, 512), ("bmm", 36, 16384, 1536, 256), ("bmm", 16, 16384, 768, 512),
          ("mm", 1, 266256, 3072, 1024), ("mm", 1, 67600, 6144, 2048), ("mm", 1, 17424, 6144, 2048)]
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
ops = []
for kind, b, m, k, n in SHAPES:
    if kind == "bmm":
        A = torch.randn(b, m, k, device=dev).half(); B = torch.randn(b, k, n, device=dev).half()
        ops.append((kind, b, m, k, n, (lambda A=A, B=B: torch.bmm(A, B, out_dtype=torch.float32))))
    else:
        A = torch.randn(m, k, device=dev).half(); B = torch.randn(k, n, device=dev).half()
        ops.append((kind, b, m, k, n, (lambda A=A, B=B: torch.mm(A, B, out_dtype=torch.float32))))
base = [bench(f) for *_, f in ops]
T.enable(True); T.tuning_enable(True); T.set_max_tuning_duration(200); T.set_max_tuning_iterations(30)
T.set_filename(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "tunableop.csv"))
t0 = time.time()
tuned = []
for (kind, b, m, k, n, f), t in zip(ops, base):
    try:
        tuned.append(bench(f))
    except Exception as ex:  # noqa
        tuned.append(float("nan")); print("tuning failed:", kind, b, m, k, n, repr(ex)[:200])
print(f"tuning took {time.time()-t0:.1f} s")
for (kind, b, m, k, n, f), t0_, t1 in zip(ops, base, tuned):
    fl = 2.0 * b * m * k * n
    print(f"{kind} {b:3d} x {m:6d} x {k:5d} x {n:5d}: default {t0_:7.1f} us ({fl/t0_/1e6:6.0f} TF)  tuned {t1:7.1f} us ({fl/t1/1e6:6.0f} TF)  {t0_/t1:5.2f}x")
T.write_file()
print(T.get_results()[:12] if hasattr(T, "get_results") else "")
