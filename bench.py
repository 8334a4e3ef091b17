#!/usr/bin/env python3
"""bench.py -- images/sec of encode -> quantize -> decode (256x256, codebook 2^16).

A "step" is one pass of the hot path over one batch of synthetic images already
resident in HBM: SD3-UNet encoder (PyTorch-ROCm) -> GaussianQuantRegularizer
(fused HIP kernels through libgqhip.so) -> decoder, followed -- exactly like the
reference's eval.py loop -- by the per-batch PSNR and ONE packed all-gather of
(indices, PSNR) across ranks.  Workload = BASELINE.json configs[1]:
sd3unet_gq_0.25 (codebook 2^16, dim 16, 1 group), bs = 16 per GPU, fp32.

Launch: `python bench.py --gpus 1 --steps K --warmup W`, or for N > 1
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
 --master-port P bench.py --gpus N ...` (one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))

def _use_shipped_miopen_db():
    """Point MIOpen at a private copy of the find-db/perf-db tuned for this workload on MI355X
    (vq-vae-from-gaussian-vae_amd/miopen_db, produced by `bench.py --miopen-benchmark 1`): with
    MIOPEN_FIND_MODE=FAST a hit returns the tuned solver at once and a miss falls back to the
    immediate-mode heuristic, so there is no search at start-up.  Must run before torch loads MIOpen."""
    import shutil
    import tempfile

    src = os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd", "miopen_db")
    if not os.path.isdir(src) or os.environ.get("MIOPEN_USER_DB_PATH"):
        return
    dst = tempfile.mkdtemp(prefix=f"gq_miopen_db_{os.environ.get('LOCAL_RANK', '0')}_")
    for f in os.listdir(src):
        shutil.copy(os.path.join(src, f), dst)
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")


if "--miopen-db" in sys.argv and sys.argv[sys.argv.index("--miopen-db") + 1] == "1":
    _use_shipped_miopen_db()

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

UNET = dict(attn_type="vanilla", double_z=True, z_channels=16, resolution=256, in_channels=3, out_ch=3, ch=128,
            ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[32], dropout=0.0)
N_CODES, DIM = 65536, 16
PEAK_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 MFMA dense peak
PEAK_BF16_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (2.5 PFLOP/s)


def build_model(device):
    from pit_hip.models.autoencoder import AutoencodingEngine

    torch.manual_seed(1234)  # no checkpoint offline: seeded random init of the real architecture
    vae = AutoencodingEngine(
        encoder_config={"target": "pit.modules.unet.Encoder", "params": UNET},
        decoder_config={"target": "pit.modules.unet.Decoder", "params": UNET},
        regularizer_config={"target": "pit.quantization.gaussian.GaussianQuantRegularizer",
                            "params": {"format": "bchw", "group": DIM, "n_samples": N_CODES, "backend": "hip"}},
    )
    return vae.eval().to(device)


def pmc_traffic_bytes(kernel="gq_filter_bf16_kernel"):
    """HBM bytes per filter launch from the committed rocprofv3 PMC passes (profiles/r01/pmc_*.csv,
    separate FETCH_SIZE / WRITE_SIZE runs of tools/kbench.py at this shape).  Units are KiB; gfx950
    reports half of a wide coalesced read stream, so FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM)."""
    import csv

    vals = {}
    for name in ("FETCH_SIZE", "WRITE_SIZE"):
        path = os.path.join(ROOT, "profiles", "r01", f"pmc_{name}.csv")
        if not os.path.exists(path):
            return None
        rows = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
                if kernel in r["Kernel_Name"] and r["Counter_Name"] == name]
        if not rows:
            return None
        vals[name] = sum(rows) / len(rows)
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0


def cpu_baseline(bs_sample: int = 1):
    """The CPU restatement (oracle + the same torch modules on CPU) on a bounded sample:
    `bs_sample` 256x256 images through encoder -> oracle quantiser -> decoder, all host cores."""
    import numpy as np

    from oracle import gq_oracle as O
    from pit_hip.modules.unet import Decoder, Encoder

    cores = min(os.cpu_count() or 1, 64)  # torch-CPU convs stop scaling (and SMT hurts) beyond that
    torch.set_num_threads(cores)
    torch.manual_seed(1234)
    enc, dec = Encoder(**UNET).eval(), Decoder(**UNET).eval()
    g = torch.Generator().manual_seed(1000)
    x = torch.rand(bs_sample, 3, 256, 256, generator=g) * 2 - 1
    cb = O.codebook(N_CODES, DIM, 42)
    O.lib()
    with torch.no_grad():
        enc(x[:1]); dec(torch.zeros(1, 16, 32, 32))  # warm the CPU kernels
        t0 = time.perf_counter()
        z = enc(x)
        t1 = time.perf_counter()
        zhat, ind = O.gq1_forward(z.numpy(), cb, DIM, threads=cores)
        t2 = time.perf_counter()
        dec(torch.from_numpy(zhat))
        t3 = time.perf_counter()
    total = t3 - t0
    return {"value": round(bs_sample / total, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{bs_sample} image(s) 256x256: encoder {t1 - t0:.2f}s + oracle quantiser "
                      f"({bs_sample * 1024} rows x 65536 codes, OpenMP) {t2 - t1:.2f}s + decoder {t3 - t2:.2f}s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL over xGMI) for real runs; gloo only to exercise the N>1 control flow on one GPU")
    ap.add_argument("--miopen-benchmark", type=int, default=int(os.environ.get("GQ_MIOPEN_BENCHMARK", "0")))
    ap.add_argument("--miopen-db", type=int, default=0, help="1: use the shipped MIOpen find-db (implies find API, FAST mode)")
    ap.add_argument("--channels-last", type=int, default=int(os.environ.get("GQ_CHANNELS_LAST", "1")),
                    help="1: conv stack in torch channels_last (NHWC) -- MIOpen's fp32 igemm kernels run without the "
                         "NCHW<->NHWC transposes and the fused GroupNorm/bias kernels have NHWC variants (+8%)")
    args = ap.parse_args()

    from pit_hip import _lib
    from pit_hip.eval_dist import StepRecord, gather_step, init_from_env, psnr_zero_mean

    env = init_from_env(args.dist_backend)
    rank, world = env["rank"], env["world"]
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    device = torch.device("cuda", env["local_rank"] % torch.cuda.device_count())
    torch.cuda.set_device(device)
    # 0 = MIOpen immediate mode (measured: same steady-state img/s as find mode, 35 s vs 248 s of warm-up on a
    # fresh box); 1 = find mode like the reference's trainer.benchmark: True
    torch.backends.cudnn.benchmark = bool(args.miopen_benchmark) or bool(args.miopen_db)

    vae = build_model(device)
    g = torch.Generator().manual_seed(1000 + rank)
    x = (torch.rand(args.batch, 3, args.size, args.size, generator=g) * 2 - 1).to(device)
    if args.channels_last:
        vae = vae.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)
    tokens = (args.size // 8) ** 2
    layout = StepRecord(args.batch, tokens, n_metrics=1)

    @torch.no_grad()
    def step():
        zhat, info = vae.encode(x, return_reg_log=True)
        rec = vae.decode(zhat)
        record = layout.pack(info["indices"], psnr_zero_mean(x, rec)[:, None])
        return gather_step(record, world)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Initialisation (not a benchmark step): one encode+decode so that MIOpen compiles / loads its kernels before
    # any warm-up or timed step.  With N ranks, rank 0 goes first and fills the shared on-disk kernel cache; N ranks
    # compiling the same kernels concurrently on a fresh box serialise on that cache (measured: 186 s vs 33 s).
    def init_pass():
        with torch.no_grad():
            zhat, info = vae.encode(x, return_reg_log=True)
            layout.pack(info["indices"], psnr_zero_mean(x, vae.decode(zhat))[:, None])
        torch.cuda.synchronize()

    if world > 1:
        if rank == 0:
            init_pass()
        dist.barrier()
        if rank != 0:
            init_pass()
        dist.barrier()
    else:
        init_pass()
    step()   # and one complete step (incl. the gather) so that even --warmup 0 times steady-state steps
    sync()
    for _ in range(args.warmup):
        step()
    sync()
    _lib.profile_enable(True)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # per-step spread (async, ~free)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        out = step()
        marks[i + 1].record()
    sync()
    elapsed = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    pick = lambda q: round(per_step[min(len(per_step) - 1, int(q * len(per_step)))], 3)
    step_ms = {"p10": pick(0.10), "p50": pick(0.50), "p90": pick(0.90), "note": "device time between per-step events, rank 0"}
    launches, kernel_ms = _lib.profile_collect()
    _lib.profile_enable(False)

    # Per-stage split, measured AFTER the timed region on this rank's stream, no collective.
    def stage_split(reps=5):
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(reps)]
        with torch.no_grad():
            for e in ev:
                e[0].record()
                z = vae.encoder(x)
                e[1].record()
                zhat, info = vae.regularization(z)
                e[2].record()
                rec = vae.decode(zhat)
                e[3].record()
                layout.pack(info["indices"], psnr_zero_mean(x, rec)[:, None])
                e[4].record()
        torch.cuda.synchronize()
        med = lambda i: sorted(e[i].elapsed_time(e[i + 1]) for e in ev)[reps // 2]
        return {"encoder": round(med(0), 3), "quantiser": round(med(1), 3), "decoder": round(med(2), 3),
                "psnr+pack": round(med(3), 3), "note": "median of 5 untimed extra steps, torch events"}

    stages = stage_split()

    # The fp32 MFMA filter on the same rows, for comparison (same indices; untimed extra quantiser calls).
    def fp32_filter_us(reps=5):
        with torch.no_grad():
            z = vae.encoder(x)
            _lib.set_filter("fp32")
            try:
                vae.regularization(z)
                torch.cuda.synchronize()
                _lib.profile_enable(True)
                for _ in range(reps):
                    vae.regularization(z)
                torch.cuda.synchronize()
                n_l, ms = _lib.profile_collect()
                _lib.profile_enable(False)
            finally:
                _lib.set_filter("auto")
        return ms / max(n_l, 1) * 1e3

    fp32_us = fp32_filter_us() if _lib.get_filter() == "auto" else None

    t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.dist_backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        rows = args.batch * tokens
        flops = 4.0 * DIM * N_CODES * rows  # 4*dim*N flops per row
        avg_ms = kernel_ms / max(launches, 1)
        achieved = flops / (avg_ms * 1e-3) / 1e12 if launches else 0.0
        bf16 = _lib.debug_plan(rows, N_CODES, DIM)["bf16"] == 1
        if bf16:
            # split-bf16 filter: every algorithmic fp32 MAC is executed as 3 bf16 MACs (A_h s_h + A_h s_l + A_l s_h)
            roofline = {"kernel": "gq_filter_bf16_kernel<NV=2,RT=2,CT=16,GT=1,WAVES=8> (split-bf16 MFMA filter of the fused quantiser)",
                        "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                        "executed": round(3 * achieved, 2), "executed_frac": round(3 * achieved / PEAK_BF16_TFLOPS, 4),
                        "vs_fp32_mfma_peak": round(achieved / PEAK_F32_TFLOPS, 3),
                        "note": "achieved = algorithmic fp32-equivalent flops (SURVEY 8d: 4*dim*N per row) / launch time; "
                                "the kernel executes 3 bf16 MACs per algorithmic MAC (two-term bf16 splits, exact re-rank "
                                "keeps the indices bit-identical), so executed = 3 x achieved is what the dense bf16 MFMA "
                                "peak bounds; the algorithmic rate is vs_fp32_mfma_peak x the fp32 MFMA peak (157.3)",
                        "traffic": pmc_traffic_bytes("gq_filter_bf16_kernel"),
                        "traffic_note": "bytes/launch = (2*FETCH_SIZE + WRITE_SIZE) KiB from profiles/r01 PMC passes; "
                                        "algorithmic bytes 3.3e6 + 4.2e6 (+8.4e6 bf16 codebook image, +4.2e6 candidate records)"}
            if fp32_us:
                roofline["fp32_filter"] = {"kernel": "gq_filter_kernel<16,2,8,GQ,2> (GQHIP_FILTER=fp32)",
                                           "avg_launch_us": round(fp32_us, 2),
                                           "achieved": round(flops / (fp32_us * 1e-6) / 1e12, 2), "peak": PEAK_F32_TFLOPS,
                                           "frac": round(flops / (fp32_us * 1e-6) / 1e12 / PEAK_F32_TFLOPS, 4)}
        else:
            roofline = {"kernel": "gq_filter_kernel<16,2,8,GQ,2> (fp32 MFMA filter of the fused quantiser)",
                        "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / PEAK_F32_TFLOPS, 4), "traffic": pmc_traffic_bytes("gq_filter_kernel"),
                        "traffic_note": "bytes/launch = (2*FETCH_SIZE + WRITE_SIZE) KiB from profiles/r01 PMC passes; "
                                        "algorithmic bytes 7.5e6 (+4.2e6 candidate records)"}
        roofline.update({"launches": launches, "avg_launch_us": round(avg_ms * 1e3, 2),
                         "timing": "hipEvents attached to the dispatch (hipExtLaunchKernelGGL) on the launch stream",
                         "algorithmic_flops_per_launch": flops})
        line = {
            "metric": "images/sec encode+quantize+decode, 256x256, codebook 2^16",
            "value": round(args.batch * world * args.steps / elapsed, 3),
            "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "sd3unet_gq_0.25 encode->quantize->decode, bs=16/GPU, 256x256, "
                                   "codebook 2^16 x dim 16, 1 group (BASELINE configs[1])",
                       "global_batch": args.batch * world, "rows_per_step_per_gpu": rows,
                       "weights": "seeded random init (seed 1234), no checkpoint offline",
                       "parallelism": f"dp{world} image-sharded, one packed all_gather/step"},
            "roofline": roofline,
            "stages_ms": stages,
            "step_ms": step_ms,
            "quantiser_rows_per_s": round(rows / (stages["quantiser"] * 1e-3)),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(4)   # ~10 s of host work
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
