"""Winograd batched GEMMs: hipBLASLt fp32 (itself a split-bf16 emulation on gfx950) against ONE bf16 GEMM with fp32
output whose K axis carries the split-bf16 products:  A' = [A_h | A_h | A_l] (x [A_m ...] for the 6-product variant),
B' = [B_h ; B_l ; B_h]  ->  A'B' = A_h B_h + A_h B_l + A_l B_h  (error ~3 * 2^-18 per product, fp32 accumulation in the MFMA).
Prints time, fp32-equivalent TFLOP/s and the error against an fp64 product for both."""
import sys
import torch

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def split2(x):
    h = x.bfloat16()
    l = (x - h.float()).bfloat16()
    return h, l


def split3(x):
    h = x.bfloat16()
    r = x - h.float()
    m = r.bfloat16()
    l = (r - m.float()).bfloat16()
    return h, m, l


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


shapes = [("dec 256^2 c128 F4", 36, 65536, 128, 128), ("dec 128^2 c256 F4", 36, 16384, 256, 256),
          ("dec 64^2 c512 F4", 36, 4096, 512, 512), ("dec 32^2 c512 F4", 36, 1024, 512, 512),
          ("enc 256^2 c128 F2", 16, 262144, 128, 128), ("enc 128^2 c256 F2", 16, 65536, 256, 256),
          ("enc 64^2 c512 F2", 16, 16384, 512, 512), ("dec 128^2 256->512? c512->256 F4", 36, 16384, 512, 256)]
for name, nb, m, k, n in shapes:
    A = torch.randn(nb, m, k, generator=g).to(dev)
    B = (torch.randn(nb, k, n, generator=g) / k ** 0.5).to(dev)
    flops = 2.0 * nb * m * k * n
    t32 = timed(lambda: torch.bmm(A, B))
    ah, al = split2(A)
    bh, bl = split2(B)
    A3 = torch.cat([ah, ah, al], 2).contiguous()
    B3 = torch.cat([bh, bl, bh], 1).contiguous()
    t3 = timed(lambda: torch.bmm(A3, B3, out_dtype=torch.float32))
    a3 = split3(A)
    b3 = split3(B)
    # products of weight >= 2^-16: hh, hm, mh, hl, lh, mm
    A6 = torch.cat([a3[0], a3[0], a3[1], a3[0], a3[2], a3[1]], 2).contiguous()
    B6 = torch.cat([b3[0], b3[1], b3[0], b3[2], b3[0], b3[1]], 1).contiguous()
    t6 = timed(lambda: torch.bmm(A6, B6, out_dtype=torch.float32))
    # fp16 two-term splits (11-bit significands: 22 bits, 3 products), operands scaled into fp16's range
    sa, sb = 1.0 / 8, 256.0
    fh = (A * sa).half()
    fl = (A * sa - fh.float()).half()
    gh = (B * sb).half()
    gl = (B * sb - gh.float()).half()
    F3 = torch.cat([fh, fh, fl], 2).contiguous()
    G3 = torch.cat([gh, gl, gh], 1).contiguous()
    tf = timed(lambda: torch.bmm(F3, G3, out_dtype=torch.float32))
    ef = float((((torch.bmm(F3[:1, :2048], G3[:1], out_dtype=torch.float32) / (sa * sb)).double()
                 - torch.bmm(A[:1, :2048].double(), B[:1].double())).abs()
                / torch.bmm(A[:1, :2048].abs().double(), B[:1].abs().double())).max())
    print(f"{name:34s} fp16x3 {tf:7.3f} ms {flops / tf / 1e9:6.1f} TF-eq err {ef:.1e}", flush=True)
    del F3, G3
    # error on one batch against fp64
    ref = torch.bmm(A[:1, :2048].double(), B[:1].double())
    scale = torch.bmm(A[:1, :2048].abs().double(), B[:1].abs().double())
    e32 = float(((torch.bmm(A[:1, :2048], B[:1]).double() - ref).abs() / scale).max())
    e3 = float(((torch.bmm(A3[:1, :2048], B3[:1], out_dtype=torch.float32).double() - ref).abs() / scale).max())
    e6 = float(((torch.bmm(A6[:1, :2048], B6[:1], out_dtype=torch.float32).double() - ref).abs() / scale).max())
    print(f"{name:34s} [{nb}x{m}x{k}x{n}] fp32 {t32:7.3f} ms {flops / t32 / 1e9:6.1f} TF err {e32:.1e} | bf16x3 {t3:7.3f} ms "
          f"{flops / t3 / 1e9:6.1f} TF-eq err {e3:.1e} | bf16x6 {t6:7.3f} ms {flops / t6 / 1e9:6.1f} TF-eq err {e6:.1e}", flush=True)
    del A, B, A3, B3, A6, B6
