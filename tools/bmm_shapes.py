"""hipBLASLt / rocBLAS rate of the batched GEMMs the Winograd convolutions issue (random data)."""
import torch, time, sys
dev = torch.device("cuda:0")
lib = sys.argv[1] if len(sys.argv) > 1 else "default"
if lib != "default":
    torch.backends.cuda.preferred_blas_library(lib)
print("preferred blas:", torch.backends.cuda.preferred_blas_library())
for (nb, T, K, N) in ((36, 65536, 128, 128), (16, 262144, 128, 128), (36, 65536, 256, 128), (36, 16384, 512, 512), (36, 16384, 512, 256), (16, 16384, 512, 512)):
    V = torch.randn(nb, T, K, device=dev) * 0.1; U = torch.randn(nb, K, N, device=dev) * 0.1
    for _ in range(3): M = torch.bmm(V, U)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): M = torch.bmm(V, U)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    gb = (V.numel() + M.numel()) * 4 / 1e9
    print(f"  bmm [{nb},{T},{K}]x[{K},{N}]: {dt*1e3:.3f} ms, {2*nb*T*K*N/dt/1e12:.0f} TFLOP/s, {gb/dt/1e3:.2f} TB/s")
