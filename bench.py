#!/usr/bin/env python3
"""bench.py -- images/sec of encode -> quantize -> decode (256x256, codebook 2^16).

A "step" is one pass of the hot path over one batch of synthetic images already
resident in HBM: SD3-UNet encoder (PyTorch-ROCm) -> GaussianQuantRegularizer
(fused HIP kernels through libgqhip.so) -> decoder, followed -- exactly like the
reference's eval.py loop -- by the per-batch PSNR and ONE packed all-gather of
(indices, PSNR) across ranks.  Default workload = BASELINE.json configs[1]:
sd3unet_gq_0.25 (codebook 2^16, dim 16, 1 group), bs = 16 per GPU, fp32.
`--config gq_0.50|gq_1.00|gq2_0.25|vq_16|lfq_16 [--size 512]` runs configs[3] / configs[4].

Launch: `python bench.py --gpus N --steps K --warmup W`.  For N > 1 the process either is one rank of an
external launcher (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`: RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment) or -- when RANK is not set -- starts the N ranks itself as fresh child
processes (the way the reference is started once per node, Readme.md:119-126 / eval.py:78-91) BEFORE anything
touches the GPU, relays rank 0's JSON line and exits with the children's return code.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))

# ----------------------------------------------------------------------------- arguments
# BASELINE.json configs: the regularizer blocks of the reference's shipped YAMLs (configs/sd3unet_*.yaml, lines 27-33),
# backend switched to the fused HIP path.  `double_z` / `z_channels` as in each YAML's encoder_config.
CONFIGS = {
    "gq_0.25": dict(target="pit.quantization.gaussian.GaussianQuantRegularizer",
                    params={"format": "bchw", "group": 16, "n_samples": 65536, "backend": "hip"},
                    double_z=True, dim=16, K=1, family="gq", baseline="configs[1]"),
    "gq_0.50": dict(target="pit.quantization.gaussian.GaussianQuantRegularizer",
                    params={"format": "bchw", "group": 8, "n_samples": 65536, "backend": "hip"},
                    double_z=True, dim=8, K=2, family="gq", baseline="configs[3]"),
    "gq_1.00": dict(target="pit.quantization.gaussian.GaussianQuantRegularizer",
                    params={"format": "bchw", "group": 4, "n_samples": 65536, "backend": "hip"},
                    double_z=True, dim=4, K=4, family="gq", baseline="configs[3]"),
    "gq2_0.25": dict(target="pit.quantization.gaussian.GaussianQuantRegularizer2",
                     params={"dim": 16, "codebook_size": 65536, "backend": "hip"},
                     double_z=True, dim=16, K=1, family="gq2", baseline="configs[3]"),
    "vq_16": dict(target="pit.quantization.vq.VQQuantizer", params={"format": "bchw", "n": 65536, "dim": 16},
                  double_z=False, dim=16, K=1, family="vq", baseline="configs[4]"),
    "lfq_16": dict(target="pit.quantization.lfq.LFQQuantizer",
                   params={"format": "bchw", "codebook_size": 256, "num_codebooks": 2},
                   double_z=False, dim=16, K=1, family="lfq", baseline="configs[4]"),
}
N_CODES = 65536
PEAK_F32_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32 MFMA dense peak
PEAK_BF16_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak (2.5 PFLOP/s)
PEAK_HBM_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E spec peak


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--config", default="gq_0.25", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-gpu", action="store_true",
                    help="skip the same-run leg that times the REFERENCE's own GPU call sequence (ATen / MIOpen convolutions, "
                         "gq_cuda op -> argmax -> index_select) on this device")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL over xGMI) for real runs; gloo only to exercise the N>1 control flow on one GPU")
    ap.add_argument("--miopen-benchmark", type=int, default=int(os.environ.get("GQ_MIOPEN_BENCHMARK", "0")))
    ap.add_argument("--channels-last", type=int, default=int(os.environ.get("GQ_CHANNELS_LAST", "1")),
                    help="1: conv stack in torch channels_last (NHWC) -- MIOpen's fp32 igemm kernels run without the "
                         "NCHW<->NHWC transposes and the fused GroupNorm/bias kernels have NHWC variants (+8%)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------- self-launch (N > 1, no launcher)
def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_plan(argv, n: int, port: int, base_env=None):
    """One (cmd, env) per rank: this script again, with the env:// rendezvous variables torch.distributed.run would
    set (eval.py:78-91 reads LOCAL_RANK / WORLD_SIZE; init_method env:// reads RANK / MASTER_*)."""
    base = dict(os.environ if base_env is None else base_env)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL / tensor sharing across processes)
    plan = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        plan.append(([sys.executable, os.path.abspath(__file__)] + list(argv), env))
    return plan


def self_launch(argv, n: int) -> int:
    """Parent of an N-rank run.  Never touches the GPU (no torch import, no HIP call) and never exec()s: it starts N
    fresh processes, relays rank 0's stdout (the ONE JSON line) and returns the first failing child's return code.
    If a rank fails, the remaining ranks (exact PIDs) are terminated so that a crash is not a hang."""
    import threading

    plan = launcher_plan(argv, n, free_port())
    procs = []
    for r, (cmd, env) in enumerate(plan):
        # ranks != 0 print nothing on stdout by contract; whatever they do print goes to stderr
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    lines = []

    def relay():
        for line in procs[0].stdout:
            lines.append(line)
            # stdout carries the ONE JSON line; anything else a library prints there (gloo's "[Gloo] Rank 0 is connected ..."
            # banner) goes to stderr
            out = sys.stdout if line.lstrip().startswith("{") else sys.stderr
            out.write(line)
            out.flush()

    th = threading.Thread(target=relay, daemon=True)
    th.start()
    rc = 0
    pending = set(range(n))
    try:
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 128 - code
                    print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                    for q in pending:          # a failed rank leaves the others blocked in a collective
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
    th.join(timeout=10)
    if rc == 0 and not any(l.lstrip().startswith("{") for l in lines):
        print("bench.py: rank 0 produced no JSON line", file=sys.stderr)
        rc = 1
    return rc


if __name__ == "__main__":
    _early = parse_args()
    if _early.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(sys.argv[1:], _early.gpus))


import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def unet_params(cfg):
    return dict(attn_type="vanilla", double_z=cfg["double_z"], z_channels=16, resolution=256, in_channels=3, out_ch=3,
                ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[32], dropout=0.0)


def build_model(device, cfg):
    from pit_hip.models.autoencoder import AutoencodingEngine

    torch.manual_seed(1234)  # no checkpoint offline: seeded random init of the real architecture
    unet = unet_params(cfg)
    vae = AutoencodingEngine(
        encoder_config={"target": "pit.modules.unet.Encoder", "params": unet},
        decoder_config={"target": "pit.modules.unet.Decoder", "params": unet},
        regularizer_config={"target": cfg["target"], "params": dict(cfg["params"])},
    )
    if cfg["family"] == "vq":   # the default uniform(+-1/n) init is degenerate; N(0,1), seed 7
        g = torch.Generator().manual_seed(7)
        with torch.no_grad():
            vae.regularization.embedding.weight.copy_(torch.randn(vae.regularization.embedding.weight.shape, generator=g))
    return vae.eval().to(device)


def pmc_traffic(kernel, config="gq_0.25"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of tools/kbench.py AT THIS CONFIG'S SHAPE
    (profiles/<round>/pmc_FETCH_SIZE_<config>.csv, pmc_WRITE_SIZE_<config>.csv; for gq_0.25 -- BASELINE configs[1] -- also the
    unsuffixed files of earlier rounds; newest round that has the kernel; None when no pass of this shape is committed).  Units are
    KiB; gfx950 reports half of a wide coalesced read stream, so FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM).
    Returns (bytes | None, provenance)."""
    import csv
    import hashlib

    if config is None:
        return None, None
    config = {"gq2_0.25": "gq_0.25"}.get(config, config)     # GQ2 dim 16 / K 1: the same rows x codes x dim, the same filter instantiation
    prof = os.path.join(ROOT, "profiles")
    for rnd in sorted((d for d in os.listdir(prof) if d.startswith("r")), reverse=True) if os.path.isdir(prof) else []:
        vals, src = {}, {}
        for name in ("FETCH_SIZE", "WRITE_SIZE"):
            path = os.path.join(prof, rnd, f"pmc_{name}_{config}.csv")
            if not os.path.exists(path) and config == "gq_0.25":
                path = os.path.join(prof, rnd, f"pmc_{name}.csv")
            if not os.path.exists(path):
                break
            rows = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
                    if kernel in r["Kernel_Name"] and r["Counter_Name"] == name]
            if not rows:
                break
            vals[name] = sum(rows) / len(rows)
            src[os.path.relpath(path, ROOT)] = hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
        if len(vals) == 2:
            return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, {"files_sha256_16": src, "launches_averaged": len(rows)}
    return None, None


# End-to-end parity gates (GPU encoder -> GPU quantiser -> GPU decoder against the reference's CPU path on the same input).
# ONE definition: tests/test_gpu_modules.py, tests/test_gpu_e2e_goldens.py gate on these numbers,
# bench.py prints them beside what it measured, BASELINE.md / DESIGN.md quote them.
GATES = {
    "z_enc_max_abs": 5e-5,               # |z_gpu - z_cpu| at 256^2 (measured 3-5e-6)
    "z_enc_max_abs_512": 2e-4,           # ... at 512^2
    "indices_differing_per_1024": 2,     # end to end, and only where the reference's own top-2 gap is below `near_tie_gap`
    "near_tie_gap": 1e-3,
    "near_tie_gap_vq": 5e-3,             # VQ: top-2 gap of the squared distance (|dz| 2e-4 moves a distance by ~ 2 |z - e| |dz| sqrt(dim))
    "near_tie_absx_lfq": 5e-4,           # LFQ: a sign flips only where |x| <= |dz|
    "same_z_gap": 1e-4,                  # reference z through the GPU quantiser: equal, or gap below the libm difference of exp / log
    "recon_max_abs_if_indices_equal": 5e-3,     # (goldens store x_rec in fp16: ulp 4.9e-4 at |x| ~ 1)
    "recon_psnr_db_if_indices_equal": 60.0,
    "recon_psnr_db": 40.0,               # when an allowed near-tie index differs, one 8 x 8 patch of the image changes
}


# ----------------------------------------------------------------------------- CPU baseline + in-run parity
def _family_legs(cfg, vae, cores):
    """The CPU side of one quantiser family: `torch_leg` = the reference's own arithmetic restated against the same torch calls
    (oracle/gq_torch_ref.py; what the reference would spend on the host cores), `c_leg` = the C / numpy oracle (the checker),
    `gap_rows(z_cpu)` = the reference's own near-tie measure per row (b, l, k), `tie_gate` = below which an end-to-end index
    difference is allowed, `target` = which leg is the bit-exact target of the GPU quantiser."""
    import numpy as np

    from oracle import gq_oracle as O
    from oracle import gq_torch_ref as T

    fam, dim, K = cfg["family"], cfg["dim"], cfg["K"]
    reg = vae.regularization
    if fam in ("gq", "gq2"):
        cb = reg.prior_samples.detach().cpu()
        cbn = cb.numpy()
        strided = fam == "gq"

        def gap_rows(z_cpu):
            b_, c2, h_, w_ = z_cpu.shape
            zf = z_cpu.reshape(b_, c2, h_ * w_).transpose(1, 2)
            mu_c, lv_c = zf.chunk(2, 2)
            sd_c = torch.exp(0.5 * torch.clamp(lv_c, -30.0, 20.0))
            k_ = (c2 // 2) // dim
            if strided:   # column g of sub-codebook k <- channel g K + k (gaussian.py:122-123)
                rows_of = lambda t: t.reshape(b_, h_ * w_, dim, k_).permute(0, 1, 3, 2).reshape(-1, dim).contiguous()
            else:         # channel k dim + g (gaussian.py:286-287)
                rows_of = lambda t: t.reshape(-1, dim).contiguous()
            mu_r, sd_r = rows_of(mu_c), rows_of(sd_c)
            _, _, best, second = O.argmax_rows(mu_r.numpy(), sd_r.numpy(), cbn, 1.0, logstd=sd_r.log().numpy(), with_gap=True,
                                               threads=cores)
            return (best - second).astype(np.float64)

        if strided:
            return dict(torch_leg=lambda z: T.gq1_forward(z, cb, dim), c_leg=lambda zn: O.gq1_forward(zn, cbn, dim, threads=cores) + (None,),
                        gap_rows=gap_rows, tie_gate=GATES["near_tie_gap"], target="torch-restatement",
                        what_torch="oracle/gq_torch_ref.py:gq1_forward: Normal.log_prob - nlp*beta, sum, argmax in 8 chunks (the "
                                   "reference's backend='torch' path, gaussian.py:136-150)",
                        what_c="oracle/gq_oracle.c, OpenMP, same op order", gap_what="top-2 gap of the reference's score")
        return dict(torch_leg=lambda z: T.gq2_forward(z, cb, dim), c_leg=lambda zn: O.gq2_quant_vq(zn, cbn, dim, 1, threads=cores) + (None,),
                    gap_rows=gap_rows, tie_gate=GATES["near_tie_gap"], target="torch-restatement",
                    what_torch="oracle/gq_torch_ref.py:gq2_forward: GaussianQuantRegularizer2.quant_vq's backend='torch' arithmetic "
                               "(gaussian.py:273-331)",
                    what_c="oracle/gq_oracle.c, OpenMP, same op order", gap_what="top-2 gap of the reference's score")
    if fam == "vq":
        emb = reg.embedding.weight.detach().cpu().float()
        embn = emb.numpy()

        def c_leg(zn):
            zq, ind, _, gap = O.vq_forward_eval(zn, embn, K, "bchw", reg.beta, reg.legacy, threads=cores)
            return zq, ind, gap

        def gap_rows(z_cpu):
            g = O.vq_forward(z_cpu.numpy(), embn, K, "bchw", cores, with_gap=True)[2]
            return np.ascontiguousarray(g.transpose(0, 2, 3, 1)).reshape(-1).astype(np.float64)

        return dict(torch_leg=lambda z: T.vq_forward(z, emb, K), c_leg=c_leg, gap_rows=gap_rows, tie_gate=GATES["near_tie_gap_vq"],
                    target="c-oracle",
                    what_torch="oracle/gq_torch_ref.py:vq_forward: sum z^2 + sum e^2 - 2 einsum, argmin, embedding, z + (z_q - z) "
                               "(vq.py:58-89; fp32, BLAS accumulation order)",
                    what_c="oracle/gq_oracle.c:vq_oracle_argmin: the same distance in fp64 (the arbiter the GPU path is pinned to: the "
                           "fp32 einsum's order is undefined, the two agree wherever the top-2 gap > 1e-4)",
                    gap_what="top-2 gap of the fp64 distance")
    if fam == "lfq":
        def c_leg(zn):
            q, ind = O.lfq_forward(zn)
            x = zn.astype(np.float32)
            return x + (q - x), ind, None

        def gap_rows(z_cpu):
            return z_cpu.abs().amin(dim=1).reshape(-1).double().numpy()       # a sign can flip only where |x| is tiny

        return dict(torch_leg=T.lfq_forward, c_leg=c_leg, gap_rows=gap_rows, tie_gate=GATES["near_tie_absx_lfq"], target="torch-restatement",
                    what_torch="oracle/gq_torch_ref.py:lfq_forward: sign, 16-bit Horner pack, x + (q - x) (lfq.py:147-158, 196-208)",
                    what_c="oracle/gq_oracle.py:lfq_forward (numpy)", gap_what="min |x| over the row's 16 channels")
    raise ValueError(fam)


def cpu_baseline_and_parity(vae, x, cfg, channels_last):
    """Rank 0, N = 1, after the timed region.  ONE image (x[:1]) through the CPU path, timed on the host cores:
    torch-CPU encoder -> quantiser -> torch-CPU decoder with this model's weights.  The quantiser runs twice:
    leg "torch-restatement" = the reference's own arithmetic in torch (oracle/gq_torch_ref.py) and leg "c-oracle" = oracle/gq_oracle.c
    (OpenMP) / its numpy parts.  The same image then goes through the GPU path and the two are compared (north_star: indices
    bit-identical, reconstruction within a stated tolerance).  The quantiser alone is then checked on EVERY row of the step: the GPU
    encoder's z of the whole batch through the C oracle (the checker) against the GPU quantiser's indices and output on that same
    z.  Every BASELINE config has this block: gq (GaussianQuantRegularizer), gq2 (GaussianQuantRegularizer2), vq, lfq."""
    import numpy as np

    from oracle import gq_oracle as O
    from pit_hip.modules.unet import Decoder, Encoder

    cores = min(os.cpu_count() or 1, 64)  # torch-CPU convs stop scaling (and SMT hurts) beyond that
    torch.set_num_threads(cores)
    unet = unet_params(cfg)
    enc, dec = Encoder(**unet).eval(), Decoder(**unet).eval()
    enc.load_state_dict({k: v.detach().cpu() for k, v in vae.encoder.state_dict().items()})
    dec.load_state_dict({k: v.detach().cpu() for k, v in vae.decoder.state_dict().items()})
    x1 = x[:1].detach().to("cpu", memory_format=torch.contiguous_format)
    dim, fam = cfg["dim"], cfg["family"]
    legs = _family_legs(cfg, vae, cores)
    O.lib()
    with torch.no_grad():
        enc(x1[:, :, :64, :64])
        dec(torch.zeros(1, 16, 8, 8))   # warm the CPU kernels
        t0 = time.perf_counter()
        z_cpu = enc(x1)
        t1 = time.perf_counter()
        zhat_t, ind_t = legs["torch_leg"](z_cpu)
        t2 = time.perf_counter()
        zhat_c, ind_c, gap_c = legs["c_leg"](z_cpu.numpy())
        t3 = time.perf_counter()
        rec_cpu = dec(zhat_t)
        t4 = time.perf_counter()
    rows = int(ind_t.numel())
    t_enc, t_qt, t_qc, t_dec = t1 - t0, t2 - t1, t3 - t2, t4 - t3
    if gap_c is None:
        legs_agree = bool(np.array_equal(ind_t.numpy(), ind_c) and np.array_equal(zhat_t.numpy(), zhat_c))
        legs_note = "bit for bit"
    else:                      # VQ: the fp32 einsum's accumulation order is BLAS's; the legs must agree wherever the top-2 gap is clear
        clear = gap_c > 1e-4
        legs_agree = bool(np.array_equal(ind_t.numpy()[clear], ind_c[clear]))
        legs_note = f"wherever the fp64 top-2 gap > 1e-4 ({int((~clear).sum())} of {clear.size} rows are nearer ties than that)"
    # the bit-exact target of the GPU quantiser
    if legs["target"] == "c-oracle":
        ind_ref, zhat_ref = torch.from_numpy(np.ascontiguousarray(ind_c)), torch.from_numpy(np.ascontiguousarray(zhat_c))
    else:
        ind_ref, zhat_ref = ind_t, zhat_t
    baseline = {
        "value": round(1.0 / (t_enc + t_qt + t_dec), 5), "unit": "images/s", "cores": cores, "kind": "port",
        "sample": f"1 image {x1.shape[-1]}x{x1.shape[-1]} ({rows} rows x {N_CODES if fam != 'lfq' else 65536} codes x dim {dim}): "
                  f"torch-CPU encoder {t_enc:.2f}s + quantiser, the reference's arithmetic in torch {t_qt:.2f}s + torch-CPU decoder {t_dec:.2f}s",
        "legs": [
            {"kind": "torch-restatement", "what": legs["what_torch"],
             "quantiser_s": round(t_qt, 3), "rows": rows, "rows_per_s": round(rows / max(t_qt, 1e-9), 1),
             "images_per_s_end_to_end": round(1.0 / (t_enc + t_qt + t_dec), 5)},
            {"kind": "c-oracle", "what": legs["what_c"], "quantiser_s": round(t_qc, 3),
             "rows": rows, "rows_per_s": round(rows / max(t_qc, 1e-9), 1),
             "images_per_s_end_to_end": round(1.0 / (t_enc + t_qc + t_dec), 5)},
        ],
        "legs_agree": legs_agree, "legs_agree_on": legs_note, "bit_exact_target": legs["target"],
        "legs_agree_bit_for_bit": legs_agree if gap_c is None else None,
    }

    # the same image through the GPU path
    dev = x.device
    xg = x[:1]
    with torch.no_grad():
        z_gpu = vae.encoder(xg)
        zhat_g, info_g = vae.regularization(z_gpu)
        rec_gpu = vae.decode(zhat_g)
        # the GPU quantiser on the CPU encoder's z: the bit-exact gate (no conv rounding in between)
        zhat_s, info_s = vae.regularization(z_cpu.to(dev))
        # every row of the step: the GPU encoder's z of the WHOLE batch, GPU quantiser vs the C oracle on that same z
        z_all = vae.encoder(x)
        zhat_all, info_all = vae.regularization(z_all)
    torch.cuda.synchronize()
    z_all_c = z_all.float().to("cpu", memory_format=torch.contiguous_format)
    t5 = time.perf_counter()
    zhat_o, ind_o, _ = legs["c_leg"](z_all_c.numpy())
    t6 = time.perf_counter()
    ind_all = info_all["indices"].cpu().numpy()
    zhat_all_c = zhat_all.float().to("cpu", memory_format=torch.contiguous_format).numpy()
    ind_g = info_g["indices"].cpu()
    ind_s = info_s["indices"].cpu()
    rec_g = rec_gpu.float().cpu().contiguous()
    mse = float(((rec_g - rec_cpu) ** 2).mean())
    n_diff = int((ind_g != ind_t).sum())
    psnr = round(10.0 * float(np.log10(4.0 / max(mse, 1e-30))), 2)
    max_abs = float((rec_g - rec_cpu).abs().max())
    dz = float((z_gpu.float().cpu() - z_cpu).abs().max())
    size = int(x.shape[-1])
    gate_dz = GATES["z_enc_max_abs"] if size <= 256 else GATES["z_enc_max_abs_512"]
    # the reference's own near-tie measure on the rows that differ end to end (an index may differ ONLY at a near-tie)
    to_rows = lambda t: t.permute(0, 2, 3, 1).reshape(-1)          # [B, K, h, w] -> rows (b, l, k)
    diff_rows = (to_rows(ind_g) != to_rows(ind_t)).numpy()
    gaps_at_diff = [float(g) for g in legs["gap_rows"](z_cpu)[diff_rows]] if diff_rows.any() else []
    all_rows_ok = bool((ind_all == ind_o).all() and np.array_equal(zhat_all_c, zhat_o))
    same_z_ok = bool((ind_s == ind_ref).all() and torch.equal(zhat_s.float().cpu().contiguous(), zhat_ref))
    ok = (dz <= gate_dz and n_diff <= GATES["indices_differing_per_1024"] * max(1, ind_t.numel() // 1024)
          and all(g < legs["tie_gate"] for g in gaps_at_diff)
          and psnr >= (GATES["recon_psnr_db_if_indices_equal"] if n_diff == 0 else GATES["recon_psnr_db"])
          and (n_diff > 0 or max_abs <= GATES["recon_max_abs_if_indices_equal"])
          and all_rows_ok and same_z_ok and legs_agree)
    parity = {
        "sample": "image 0 of the batch, GPU path vs the CPU path timed above (same weights, same input); quantiser_all_rows: "
                  "the whole batch",
        "quantiser_same_z": {"indices_equal_frac": float((ind_s == ind_ref).float().mean()),
                             "zhat_bit_equal": bool(torch.equal(zhat_s.float().cpu().contiguous(), zhat_ref)),
                             "note": f"GPU quantiser fed the CPU encoder's z, against the {legs['target']} leg: must be 1.0 / true "
                                     "(bit-exact contract)"},
        "quantiser_all_rows": {"rows": int(ind_o.size), "images": int(x.shape[0]),
                               "indices_equal_frac": float((ind_all == ind_o).mean()),
                               "indices_differing": int((ind_all != ind_o).sum()),
                               "zhat_bit_equal": bool(np.array_equal(zhat_all_c, zhat_o)),
                               "oracle_s": round(t6 - t5, 3),
                               "note": "every row of the step: the GPU encoder's z of the whole batch, GPU quantiser vs the c-oracle leg "
                                       "on that same z: must be 1.0 / 0 / true"},
        "indices_equal_frac": float((ind_g == ind_t).float().mean()),
        "indices_differing": n_diff,
        "reference_near_tie_measure_at_differing_rows": gaps_at_diff,
        "reference_top2_gap_at_differing_rows": gaps_at_diff,          # (the same list under its earlier name)
        "near_tie_measure": legs["gap_what"], "near_tie_gate": legs["tie_gate"],
        "z_enc_max_abs_err": dz,
        "recon_max_abs_err": max_abs,
        "recon_psnr_db": psnr,
        "gates": dict(GATES), "within_gates": bool(ok),
        "within_gates_requires": "|dz| gate, count of differing indices AND each one's near-tie measure below its gate, the "
                                 "reconstruction gates, quantiser_same_z exact, quantiser_all_rows exact, the CPU legs in agreement",
        "tolerance": f"end to end the GPU encoder's fp32 rounding differs from the CPU's (|dz| <= {gate_dz:g}), so an index may "
                     f"differ only at a near-tie of the reference's own decision (<= {GATES['indices_differing_per_1024']} per 1024 rows, "
                     f"{legs['gap_what']} < {legs['tie_gate']:g}); reconstruction with all indices equal: max-abs <= "
                     f"{GATES['recon_max_abs_if_indices_equal']:g}, PSNR >= {GATES['recon_psnr_db_if_indices_equal']:g} dB; with an "
                     f"allowed near-tie difference: PSNR >= {GATES['recon_psnr_db']:g} dB (bench.GATES; the -m gpu end-to-end golden "
                     "tests gate on the same numbers)",
    }
    return baseline, parity


# ----------------------------------------------------------------------------- the reference's own GPU path, same run
class reference_ops:
    """Context: the conv stack as the reference's own op sequence -- ATen GroupNorm, F.silu, MIOpen convolutions with their
    bias passes, F.scaled_dot_product_attention (pit/modules/unet.py:49-57, 137-153, 185-206) -- by switching libgqhip's
    fused kernels off in pit_hip.modules.unet (NCHW tensors, no caches, no weight guard)."""

    SWITCHES = {"FUSED_GN": False, "ATTN_MATH": False, "WEIGHT_GUARD": False}

    def __enter__(self):
        from pit_hip.modules import unet as U

        self.U, self.old = U, {k: getattr(U, k) for k in self.SWITCHES}
        for k, v in self.SWITCHES.items():
            setattr(U, k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            setattr(self.U, k, v)


def reference_vq_forward(z, emb):
    """pit/quantization/vq.py:39-96 (bchw, codebook_num 1), the eval-relevant part: distance matrix by einsum, argmin,
    embedding lookup.  Returns (z_q [B, c, h, w], indices [B, 1, h, w])."""
    b, c, h, w = z.shape
    zf = z.permute(0, 2, 3, 1).contiguous().view(-1, c)
    d = torch.sum(zf ** 2, dim=1, keepdim=True) + torch.sum(emb ** 2, dim=1) - 2 * torch.einsum("bd,dn->bn", zf, emb.t())
    ind = torch.argmin(d, dim=1)
    zq = torch.nn.functional.embedding(ind, emb).view(b, h, w, c)
    return zq.permute(0, 3, 1, 2).contiguous(), ind.view(b, h, w, 1).permute(0, 3, 1, 2).contiguous()


def reference_gpu_path(vae, x, cfg, product_indices, product_images_per_s, steps=10, warmup=8, product_step_ms_p50=None):
    """Rank 0, N = 1, after the timed region: what the REFERENCE is on this device.  Same weights and input as the product
    path, but NCHW modules on ATen / MIOpen ops (reference_ops) and the reference's quantiser call sequence: backend="cuda",
    i.e. the extension_cpp::gq op into the persistent rows x 65 536 `perturbed` buffer, torch.argmax, index_select
    (pit/quantization/gaussian.py:124-133 / :289-298; VQ: the einsum distance matrix + argmin of vq.py:58-73; LFQ: elementwise,
    the product module itself).  The op behind `gq_cuda.ops.gq_cuda` is this repo's HIP build of it (the CUDA source cannot be
    built here) -- faster than the reference's one-thread-per-pair kernel would be, so the leg errs in the reference's favour.
    `warmup` (>= 8: MIOpen's immediate mode runs a process's first eight calls of a convolution on a slow generic kernel) untimed
    steps, then `steps` timed ones -- the same treatment as the product loop; the step = encode -> quantise -> decode -> PSNR ->
    pack, as in the product loop.  Reported: the MEDIAN per-step device time (and the wall mean).  MIOpen in immediate mode
    (torch.backends.cudnn.benchmark as the product loop has it, --miopen-benchmark); find mode costs minutes of warm-up on a
    fresh box and measured the same steady state (main(), comment at cudnn.benchmark)."""
    import copy

    from pit_hip.eval_dist import StepRecord, psnr_zero_mean
    from pit_hip.modules.unet import Decoder, Encoder

    dev = x.device
    unet = unet_params(cfg)
    with reference_ops():
        enc, dec = Encoder(**unet).eval().to(dev), Decoder(**unet).eval().to(dev)          # fresh modules: contiguous (NCHW) weights
        enc.load_state_dict({k: v.detach().contiguous() for k, v in vae.encoder.state_dict().items()})
        dec.load_state_dict({k: v.detach().contiguous() for k, v in vae.decoder.state_dict().items()})
        fam = cfg["family"]
        if fam == "gq":
            from pit_hip.quantization.gaussian import GaussianQuantRegularizer

            reg = GaussianQuantRegularizer(**dict(cfg["params"], backend="cuda-compat")).eval().to(dev)
            quant = lambda z: (lambda zh, info: (zh, info["indices"]))(*reg(z))
            seq = "gq_cuda op -> rows x 65536 fp32 matrix in HBM -> torch.argmax -> index_select"
        elif fam == "gq2":
            from pit_hip.quantization.gaussian import GaussianQuantRegularizer2

            reg = GaussianQuantRegularizer2(**dict(cfg["params"], backend="cuda-compat")).eval().to(dev)
            quant = lambda z: (lambda zh, info: (zh, info["indices"]))(*reg(z))
            seq = "quant_gaussian + gq_cuda op -> rows x 65536 fp32 matrix in HBM -> torch.argmax -> index_select"
        elif fam == "vq":
            emb = vae.regularization.embedding.weight.detach().contiguous()
            quant = lambda z: reference_vq_forward(z.float(), emb)
            seq = "einsum distance matrix rows x 65536 in HBM -> torch.argmin -> embedding"
        else:
            reg = copy.copy(vae.regularization)
            quant = lambda z: (lambda zh, info: (zh, info["indices"]))(*reg(z))
            seq = "elementwise sign + pack (the product module: there is no matrix to materialise)"
        xn = x.contiguous()                                                                  # NCHW
        tokens = product_indices[0].numel()
        layout = StepRecord(x.shape[0], tokens, n_metrics=1)

        @torch.no_grad()
        def one(ev=None):
            if ev: ev[0].record()
            z = enc(xn)
            if ev: ev[1].record()
            zhat, ind = quant(z)
            if ev: ev[2].record()
            rec = dec(zhat)
            if ev: ev[3].record()
            layout.pack(ind, psnr_zero_mean(xn, rec)[:, None])
            if ev: ev[4].record()
            return ind

        for _ in range(max(warmup, 8)):           # MIOpen solver selection / kernel load / its slow first eight calls, `perturbed`
            one()
        torch.cuda.synchronize()
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(steps)]
        t0 = time.perf_counter()
        for e in evs:
            ind = one(e)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps
    med = lambda i: sorted(e[i].elapsed_time(e[i + 1]) for e in evs)[steps // 2]
    step_med_ms = sorted(e[0].elapsed_time(e[4]) for e in evs)[steps // 2]
    eq = float((ind.reshape(-1) == product_indices.reshape(-1).to(ind.device)).float().mean())
    ips = x.shape[0] / (step_med_ms * 1e-3)
    return {
        "what": "the reference's own call sequence on this GPU, same weights / input / run: NCHW modules on ATen + MIOpen ops "
                f"(GroupNorm, silu, conv2d + bias, scaled_dot_product_attention), quantiser = {seq} "
                "(pit/quantization/gaussian.py:124-133); the gq op is this repo's HIP build (the CUDA source is unbuildable here)",
        "images_per_s": round(ips, 2), "ms_per_step": round(step_med_ms, 3), "ms_per_step_wall_mean": round(wall * 1e3, 3),
        "steps": steps, "warmup": max(warmup, 8), "statistic": "median per-step device time (torch events around the whole step)",
        "stages_ms": {"encoder": round(med(0), 3), "quantiser": round(med(1), 3), "decoder": round(med(2), 3),
                      "psnr+pack": round(med(3), 3)},
        "indices_equal_frac_vs_product": eq,
        "indices_note": "the product path's indices are the CPU reference's (parity block); the compat op's score matrix is "
                        "2 s + const(r) in another rounding, and the NCHW encoder's z differs by fp32 rounding, so a few "
                        "near-tie rows may differ here",
        # same statistic on both sides: the product's median per-step device time (line["step_ms"]["p50"]) against this leg's
        # median per-step device time; the ratio of the product's wall-clock MEAN (= line["value"], host gaps and the gather
        # included) to this leg's median is kept beside it under its own name
        "product_over_reference": round((x.shape[0] / (product_step_ms_p50 * 1e-3) if product_step_ms_p50 else product_images_per_s) / ips, 3),
        "product_over_reference_statistic": "median device step / median device step" if product_step_ms_p50 else "wall mean / median device step",
        "product_wall_mean_over_reference_median": round(product_images_per_s / ips, 3),
    }


# ----------------------------------------------------------------------------- main
def main():
    args = parse_args()
    cfg = CONFIGS[args.config]

    from pit_hip import _lib
    from pit_hip.eval_dist import StepRecord, gather_step, init_from_env, psnr_zero_mean

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        print(f"bench.py: WORLD_SIZE={world_env} but --gpus {args.gpus}. Either run `python bench.py --gpus {args.gpus}` "
              f"without RANK set (it starts the ranks itself) or launch every rank with `python -m torch.distributed.run "
              f"--nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...`.",
              file=sys.stderr)
        sys.exit(2)
    if args.dist_backend == "nccl" and args.gpus > max(torch.cuda.device_count(), 1):
        print(f"bench.py: --gpus {args.gpus} with RCCL needs {args.gpus} devices, {torch.cuda.device_count()} visible "
              "(use --dist-backend gloo to exercise the N>1 control flow on fewer GPUs).", file=sys.stderr)
        sys.exit(2)
    env = init_from_env(args.dist_backend)
    rank, world = env["rank"], env["world"]
    device = torch.device("cuda", env["local_rank"] % torch.cuda.device_count())
    torch.cuda.set_device(device)
    # 0 = MIOpen immediate mode (measured: same steady-state img/s as find mode, 35 s vs 248 s of warm-up on a
    # fresh box); 1 = find mode like the reference's trainer.benchmark: True
    torch.backends.cudnn.benchmark = bool(args.miopen_benchmark)

    vae = build_model(device, cfg)
    g = torch.Generator().manual_seed(1000 + rank)
    x = (torch.rand(args.batch, 3, args.size, args.size, generator=g) * 2 - 1).to(device)
    if args.channels_last:
        vae = vae.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)
    tokens = (args.size // 8) ** 2 * (cfg["K"] if cfg["family"] in ("gq", "gq2", "vq") else 1)
    layout = StepRecord(args.batch, tokens, n_metrics=1)
    gather_ev = []

    @torch.no_grad()
    def step():
        zhat, info = vae.encode(x, return_reg_log=True)
        rec = vae.decode(zhat)
        record = layout.pack_with_psnr(info["indices"], x, rec)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = gather_step(record, world)
        e1.record()
        gather_ev.append((e0, e1))
        return out

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
            layout.pack_with_psnr(info["indices"], x, vae.decode(zhat))
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
    gather_ev.clear()
    _lib.profile_reserve(4 * args.steps + 64)   # event pairs created here, none inside the timed region
    _lib.profile_enable(True)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # per-step spread (async, ~free)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        step()
        marks[i + 1].record()
    sync()
    elapsed = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    pick = lambda v, q: round(v[min(len(v) - 1, int(q * len(v)))], 3)
    step_ms = {"p10": pick(per_step, 0.10), "p50": pick(per_step, 0.50), "p90": pick(per_step, 0.90),
               "note": "device time between per-step events, rank 0"}
    gather_ms = sorted(a.elapsed_time(b) for a, b in gather_ev)
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
                layout.pack_with_psnr(info["indices"], x, rec)
                e[4].record()
        torch.cuda.synchronize()
        med = lambda i: sorted(e[i].elapsed_time(e[i + 1]) for e in ev)[reps // 2]
        return {"encoder": round(med(0), 3), "quantiser": round(med(1), 3), "decoder": round(med(2), 3),
                "psnr+pack": round(med(3), 3), "note": "median of 5 untimed extra steps, torch events"}

    stages = stage_split()

    # The quantiser alone, back to back (no conv kernels in between): what a tokenizer-only caller pays per call.
    def quantiser_call_us(reps=20):
        with torch.no_grad():
            z = vae.encoder(x)
            for _ in range(3):
                vae.regularization(z)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                vae.regularization(z)
            b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3

    call_us = quantiser_call_us()

    # The same call on SURVEY.md 8(d)'s quantiser-only inputs (mu = 0.9 randn, logvar = -1.5 + 0.3 randn: ~1.1 bit per dimension, near
    # the trained operating point -- the z of this bench's seeded-random encoder has sigma ~ 1, logvar ~ 0, where the score is nearly
    # flat in n): rows, dim and codebook as in the step; through the C ABI's row entry point.
    def quantiser_call_us_8d(reps=20):
        if cfg["family"] not in ("gq", "gq2"):
            return None
        dim_, rows_ = cfg["dim"], args.batch * tokens
        g8 = torch.Generator().manual_seed(0)
        mu8 = (0.9 * torch.randn(rows_, dim_, generator=g8)).to(device)
        sd8 = torch.exp(0.5 * (-1.5 + 0.3 * torch.randn(rows_, dim_, generator=g8))).to(device)
        cb8 = vae.regularization.prior_samples
        ws8 = _lib.Workspace()
        for _ in range(3):
            _lib.gq_argmax(mu8, sd8, cb8, 1.0, ws=ws8)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            _lib.gq_argmax(mu8, sd8, cb8, 1.0, ws=ws8)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3

    call_us_8d = quantiser_call_us_8d()

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

    mfma_family = cfg["family"] in ("gq", "gq2", "vq")
    fp32_us = fp32_filter_us() if (mfma_family and _lib.get_filter() == "auto") else None

    # per-rank medians and the MAX of the elapsed times over ranks
    cpu_coll = args.dist_backend != "nccl"
    t = torch.tensor([elapsed, per_step[len(per_step) // 2], gather_ms[len(gather_ms) // 2]], dtype=torch.float64,
                     device="cpu" if cpu_coll else device)
    rank_step_ms = [round(float(t[1]), 3)]
    rank_gather_ms = [round(float(t[2]), 4)]
    if world > 1:
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        elapsed = max(float(a[0]) for a in allt)
        rank_step_ms = [round(float(a[1]), 3) for a in allt]
        rank_gather_ms = [round(float(a[2]), 4) for a in allt]

    if rank == 0:
        dim = cfg["dim"]
        rows = args.batch * tokens
        per_row = (2.0 if cfg["family"] == "vq" else 4.0) * dim * N_CODES   # 4*dim*N flops per row (VQ: 2*dim*N)
        flops = per_row * rows
        if mfma_family:
            avg_ms = kernel_ms / max(launches, 1)
            achieved = flops / (avg_ms * 1e-3) / 1e12 if launches else 0.0
            plan = _lib.debug_plan(rows, N_CODES, dim)
            # filter kernel the plan selects: 0 fp32 MFMA, 1 split-bf16 (dim 4; GQHIP_FILTER=bf16), 2 fp16 + fp8 (GQHIP_FILTER=mixed,
            # dim 16, Gaussian score), 3 fp16 main product (the default at dims 8 / 16 / 32, Gaussian score and VQ)
            kind = plan["bf16"]
            if kind == 2 and cfg["family"] == "vq":
                kind = 1                 # fp16 + fp8 is the Gaussian score's only: VQ under GQHIP_FILTER=mixed runs split-bf16
            bf16 = kind >= 1
            whole = flops / (stages["quantiser"] * 1e-3) / 1e12
            whole_b2b = flops / (call_us * 1e-6) / 1e12
            grid = bool(_lib.lib().gqhip_grid_search_applies(N_CODES, dim)) and cfg["family"] != "lfq"
            # the committed PMC passes are of tools/kbench.py at the config's bs 16 shape (profiles/rNN/pmc_*_<config>[_<size>].csv):
            # shapes without a committed pass report no traffic
            shape_tag = (args.config if args.size == 256 else f"{args.config}_{args.size}") if args.batch == 16 else None
            launch_names = None
            if grid:
                # dim 4: no filter / re-rank -- a pruned exact search over a cached box tree of the codebook (csrc/gq_grid.h)
                kname = "gq_grid_kernel"
                traffic, prov = pmc_traffic(kname, shape_tag)
                with torch.no_grad():
                    zq_ = vae.encoder(x)
                    _lib.debug_enable(True)
                    try:
                        vae.regularization(zq_)
                        torch.cuda.synchronize()
                        gs = _lib.debug_grid(vae.regularization._ws)
                    finally:
                        _lib.debug_enable(False)
                bf16 = True        # (priced against the dense fp16 / bf16 peak, like the filter it replaces)
                launch_names = ("gq_prep_kernel", "gq_grid_build_kernel", kname)
                roofline = {"kernel": f"{kname} (pruned exact search over a box tree of the codebook: 16 -> 256 -> 1024 leaves held in LDS, "
                                      "4096 sub-leaves of 16 codes fetched from the codebook cache; it replaces filter + re-rank at dim 4: "
                                      "the dense MFMA form is bound by the VALU fold of its own outputs there, "
                                      "profiles/r04/pmc_filter_gq_1.00_dim4_round3_kernels.txt)",
                            "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                            "visited": {"sub_leaves_per_row": round(gs["sub_leaves"] / rows, 2), "sub_leaves_total": 4096,
                                        "codes_per_sub_leaf": N_CODES // 4096,
                                        "fraction_of_the_codebook_scored": round(gs["sub_leaves"] / rows / 4096, 5),
                                        "exactly_scored_codes_per_row": round(gs["exact_codes"] / rows, 3),
                                        "rows_scanned_over_all_codes": int(gs["scanned_rows"])},
                            "note": "achieved = the DENSE algorithmic flops of the shape (SURVEY 8d: 4*dim*N per row) / launch time, against "
                                    "the dense fp16 MFMA peak the replaced filter was priced on -- an equivalence rate, not work done: the "
                                    "kernel scores `visited.fraction_of_the_codebook_scored` of the pairs (fp32 FMAs on the vector ALUs, no "
                                    "matrix instruction) and is bound by dependent round trips and VALU issue (profiles/r05/"
                                    "grid_search_variants.txt); indices bit-identical to the reference's either way",
                            "traffic": traffic, "traffic_source": prov}
            elif bf16:
                kname = "gq_filter_bf16_kernel"
                traffic, prov = pmc_traffic(kname, shape_tag)
                # executed work per algorithmic fp32 MAC, in bf16-rate MACs: split-bf16 = 3 bf16 MACs (A_h s_h + A_h s_l +
                # A_l s_h); fp16 + fp8 = 1 fp16 MAC + 2 fp8 MACs on the block-scaled instruction (twice the bf16 rate) = 2
                ex = {1: 3, 2: 2, 3: 1}[kind]
                what = {1: "split-bf16 MFMA filter",
                        2: "fp16 + fp8 MFMA filter: v_mfma_f32_32x32x16_f16 main product + v_mfma_scale_f32_32x32x64_f8f6f4 corrections",
                        3: "fp16 main-product MFMA filter: v_mfma_f32_32x32x16_f16, no correction terms; the re-rank's data-dependent "
                           "bound covers the fp16 rounding"}[kind]
                roofline = {"kernel": f"{kname} ({what} of the fused quantiser; plan {plan})",
                            "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                            "executed": round(ex * achieved, 2), "executed_frac": round(ex * achieved / PEAK_BF16_TFLOPS, 4),
                            "vs_fp32_mfma_peak": round(achieved / PEAK_F32_TFLOPS, 3),
                            "note": "achieved = algorithmic fp32-equivalent flops (SURVEY 8d: 4*dim*N per row; VQ 2*dim*N) / "
                                    "launch time against the dense bf16/fp16 MFMA peak; executed = the MFMA work the kernel issues per "
                                    "algorithmic MAC in bf16-rate MACs (fp16 main product: 1; split-bf16: three bf16 products of two-term splits; "
                                    "fp16 + fp8: one fp16 product + two fp8 correction products at twice the rate = 2) -- the exact re-rank keeps "
                                    "the indices bit-identical either way",
                            "traffic": traffic, "traffic_source": prov,
                            "traffic_note": "bytes/launch = (2*FETCH_SIZE + WRITE_SIZE) KiB, separate rocprofv3 --pmc passes of "
                                            "tools/kbench.py at this shape (not re-measured in this run: PMC needs the profiler)"}
                if fp32_us:
                    roofline["fp32_filter"] = {"kernel": "gq_filter_kernel (GQHIP_FILTER=fp32)",
                                               "avg_launch_us": round(fp32_us, 2),
                                               "achieved": round(flops / (fp32_us * 1e-6) / 1e12, 2), "peak": PEAK_F32_TFLOPS,
                                               "frac": round(flops / (fp32_us * 1e-6) / 1e12 / PEAK_F32_TFLOPS, 4)}
            else:
                kname = "gq_filter_kernel"
                traffic, prov = pmc_traffic(kname, shape_tag)
                roofline = {"kernel": f"{kname} (fp32 MFMA filter of the fused quantiser)",
                            "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(achieved / PEAK_F32_TFLOPS, 4), "traffic": traffic, "traffic_source": prov}
            peak = PEAK_BF16_TFLOPS if bf16 else PEAK_F32_TFLOPS
            roofline["whole_call"] = {
                "what": "algorithmic flops / time of the WHOLE quantiser call (three launches: prep + filter + re-rank, or -- dim 4 -- "
                        "prep + index check + search; + the module's torch ops), i.e. what a caller gets",
                "in_step_ms": stages["quantiser"], "achieved": round(whole, 2), "frac": round(whole / peak, 4),
                "back_to_back_us": round(call_us, 1), "back_to_back_achieved": round(whole_b2b, 2),
                "back_to_back_frac": round(whole_b2b / peak, 4),
                "back_to_back_us_on_survey_8d_inputs": None if call_us_8d is None else round(call_us_8d, 1),
                "inputs_note": "in_step / back_to_back: the z of this run's seeded-random encoder (logvar ~ 0: sigma ~ 1, a score that is "
                               "nearly flat in n -- the hard case for the dim-4 search, whose boxes are loose in the codebook's tails); "
                               "..._on_survey_8d_inputs: rows at SURVEY 8(d)'s trained-like operating point (mu = 0.9 randn, logvar = -1.5 + "
                               "0.3 randn), same rows / dim / codebook, gq_argmax_f32"}
            # HBM traffic of the WHOLE call: the three launches' PMC bytes summed (same committed passes as `traffic`)
            parts = {}
            for kn in launch_names or ("gq_prep_kernel", kname, "gq_rerank_kernel"):      # the call's three launches
                tb, _ = pmc_traffic(kn, shape_tag)
                parts[kn] = tb
            alg_bytes = rows * (2 * dim * 4 + 8 + dim * 4) + 4 * dim * N_CODES     # rows in / index + zhat out + codebook once
            if all(v is not None for v in parts.values()):
                tot = sum(parts.values())
                roofline["whole_call_traffic"] = {"bytes": tot, "per_kernel": parts, "algorithmic_bytes": alg_bytes,
                                                  "ratio": round(tot / alg_bytes, 2),
                                                  "note": "(2*FETCH_SIZE + WRITE_SIZE) KiB per launch, summed over the call's three "
                                                          "launches (prep, filter, re-rank); algorithmic = SURVEY 8(d)'s fused figure; the "
                                                          "x2 on FETCH_SIZE is calibrated for wide streaming reads only -- the re-rank's "
                                                          "fetches are gathers"}
            else:
                roofline["whole_call_traffic"] = None
            roofline.update({"launches": launches, "avg_launch_us": round(avg_ms * 1e3, 2),
                             "timing": "hipEvents attached to the dispatch (hipExtLaunchKernelGGL) on the launch stream, over the "
                                       "timed region; event pairs pre-created",
                             "algorithmic_flops_per_launch": flops})
        else:
            # LFQ: sign + 16-bit pack, elementwise: HBM-bound, 4*16 B in + 4*16 B q out + 8 B index per row
            nbytes = rows * (16 * 4 * 2 + 8)
            roofline = {"kernel": "lfq_pack_kernel (sign + big-endian pack; closed form of the arg-min over {+-1}^16)",
                        "bound": "hbm", "achieved": round(nbytes / (stages["quantiser"] * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS,
                        "unit": "GB/s", "frac": round(nbytes / (stages["quantiser"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                        "traffic": None,
                        "note": "whole quantiser stage (module glue + one launch) by torch events; launch-latency bound at this size",
                        "algorithmic_bytes_per_launch": nbytes}
        line = {
            "metric": f"images/sec encode+quantize+decode, {args.size}x{args.size}, codebook 2^16",
            "value": round(args.batch * world * args.steps / elapsed, 3),
            "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"sd3unet_{args.config} encode->quantize->decode, bs={args.batch}/GPU, {args.size}x{args.size}, "
                                   f"codebook 2^16 x dim {dim}, {cfg['K']} sub-codebook(s) (BASELINE {cfg['baseline']})",
                       "global_batch": args.batch * world, "rows_per_step_per_gpu": rows,
                       "weights": "seeded random init (seed 1234), no checkpoint offline",
                       "parallelism": f"dp{world} image-sharded, one packed all_gather/step"},
            "gemm_precision": "fp32 results end to end.  Every wide convolution and GEMM of the conv stack runs on the fp16 matrix "
                              "cores with fp32 accumulation of the three products of two-term fp16 splits of both operands (22-bit "
                              "significands, power-of-two scales from rigorous bounds): libgqhip's direct implicit GEMM for the 3x3 "
                              "convolutions of the 256x256 level and the encoder's 128x128 level, its 1x1 / stride-2 kernels, its own "
                              "Winograd GEMM at the 256-channel level; ONE fp16 hipBLASLt GEMM over a K axis of split products for the "
                              "other Winograd / sub-pixel GEMMs and for the two attention GEMMs.  Measured error 1.8-2.9e-7 of sum|a||b| "
                              "vs 2.6-3.5e-7 for hipBLASLt's own fp32 GEMM, which on gfx950 is itself a split-bf16 emulation "
                              "(tools/conv3_bench.py, tools/bmm_bf16x3.py, tools/wino_gemm2_bench.py, tests/test_gpu_convstack_kernels.py); "
                              "the decoder's conv_out (128 -> 3): fp32 FMAs; the encoder's conv_out (the layer that produces z) and the decoder's conv_in: "
                              "libgqhip's conv3x3_f32 on the fp32 matrix cores in a fixed summation order (bit-reproducible); the "
                              "encoder's conv_in (3 -> 128): libgqhip's conv3x3_cin_small_f32, fp32 FMAs in a fixed order (no library "
                              "convolution runs at this shape).  psnr+pack: gq_step_record_f32, ONE launch -- its per-image PSNR sums the "
                              "reference's fp32 terms in fp64 and takes log10 in fp64, so it equals the torch fp32 expression "
                              "(pit/evaluations/psnr.py:17-28) to ~2e-6 dB, not bit for bit (include/gqhip.h)",
            "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "dist_backend": args.dist_backend if world > 1 else None,
            "gather_ms": {"p50": pick(gather_ms, 0.5), "p90": pick(gather_ms, 0.9),
                          "bytes_per_rank": layout.words * 4, "per_rank_p50": rank_gather_ms},
            "rank_step_ms_p50": rank_step_ms,
            "roofline": roofline,
            "stages_ms": stages,
            "step_ms": step_ms,
            "quantiser_rows_per_s": round(rows / (stages["quantiser"] * 1e-3)),
        }
        if world == 1 and not args.no_reference_gpu:
            with torch.no_grad():
                _, info_p = vae.encode(x, return_reg_log=True)
            line["reference_gpu_path"] = reference_gpu_path(vae, x, cfg, info_p["indices"], line["value"],
                                                            steps=min(max(args.steps, 3), 20), warmup=max(args.warmup, 8),
                                                            product_step_ms_p50=step_ms.get("p50"))
            # `vs_baseline` stays null: BASELINE.md holds no published number for this metric.  The same-node, same-run measurement of
            # the reference's own GPU call sequence is reported beside it (reference_gpu_path.product_over_reference), steady state
            # against steady state.
            line["vs_baseline_note"] = ("null: BASELINE.md has no published number for this metric; the reference's own GPU call sequence, "
                                        "measured in this run with the same warm-up treatment, is in reference_gpu_path")
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], line["parity"] = cpu_baseline_and_parity(vae, x, cfg, args.channels_last)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
