import torch, time
dev = torch.device("cuda:0")
torch.manual_seed(0)
print("float32_matmul_precision:", torch.get_float32_matmul_precision(), "allow_tf32:", torch.backends.cuda.matmul.allow_tf32)
for (M, K, N) in ((67600, 2048, 2048), (16384, 512, 512), (4096, 4608, 512)):
    a = torch.randn(M, K, device=dev); b = torch.randn(K, N, device=dev)
    c = a @ b
    idx = torch.randint(0, M, (512,), device=dev)
    ref = (a[idx].double() @ b.double())
    err = (c[idx].double() - ref).abs()
    scale = (a[idx].double().abs() @ b.double().abs())     # sum |a||b|
    print((M, K, N), "max err / sum|a||b| =", float((err / scale).max()), " max rel to |c| typical:", float(err.max() / ref.abs().mean()))
    # CPU-style fp32 reference error for comparison
    c32 = (a[idx].cpu() @ b.cpu())
    e32 = (c32.double() - ref.cpu()).abs()
    print("    torch-CPU fp32 same rows: max err / sum|a||b| =", float((e32 / scale.cpu()).max()))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): c = a @ b
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"    {2*M*K*N/dt/1e12:.1f} TFLOP/s")
