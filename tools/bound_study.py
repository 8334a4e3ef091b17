"""CPU study: how many candidates / undecided rows would the re-rank see under a given filter error bound?

Rows = the bench workload's own z (seeded random-init encoder on the seeded images, CPU) and, for contrast, a
"trained-VAE-like" synthetic set (sigma ~ 0.3 .. 0.8, |mu| ~ 1).  For every row the filter expansion
f(j) = sum_i A_i n_ji^2 + B_i n_ji is evaluated in fp64 against the 65 536-code codebook, the maxima of the 64-code
candidate groups are taken, and for each bound formula we count the groups within margin = 2.5 (E_f + E_r) of the row
maximum (candidates per row) and the rows where some record set (2048 codes) has 4 or more such groups (undecided).

    python tools/bound_study.py [--rows 4096]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))

U = 2.0 ** -24
C0 = 0.91893853320467274178


def bench_rows(nimg):
    import bench
    vae = bench.build_model(torch.device("cpu"), bench.CONFIGS["gq_0.25"])
    g = torch.Generator().manual_seed(1000)
    x = (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1)[:nimg]
    with torch.no_grad():
        z = vae.encoder(x)
    b, c2, h, w = z.shape
    zf = z.reshape(b, c2, h * w).transpose(1, 2)
    mu, lv = zf.chunk(2, 2)
    sd = torch.exp(0.5 * torch.clamp(lv, -30.0, 20.0))
    return mu.reshape(-1, 16).double().numpy(), sd.reshape(-1, 16).double().numpy(), vae.regularization.prior_samples.double().numpy()


def trained_like_rows(n, seed=3):
    r = np.random.default_rng(seed)
    sd = np.exp(r.uniform(np.log(0.25), np.log(0.85), (n, 16)))
    mu = r.standard_normal((n, 16)) * np.sqrt(np.maximum(1 - sd ** 2, 0.05))
    return mu, sd


def study(name, mu, sd, cb, beta=1.0):
    rows, dim = mu.shape
    n = cb.shape[0]
    A = beta / 2 - 1 / (2 * sd ** 2)
    B = mu / sd ** 2
    N1 = np.abs(cb).max()
    R2 = (cb ** 2).sum(1).max()
    S0, S1, S2, S3 = (1 / sd ** 2).sum(1), (np.abs(mu) / sd ** 2).sum(1), (mu ** 2 / sd ** 2).sum(1), np.abs(np.log(sd)).sum(1)
    T_old = (abs(beta) * dim / 2 + S0 / 2) * N1 ** 2 + S1 * N1
    Gb = 0.5 * (N1 ** 2 * S0 + 2 * N1 * S1 + S2) + S3 + dim * (C0 + abs(beta) * (0.5 * N1 ** 2 + C0))
    Er = (dim + 16) * U * Gb
    # per-coordinate classes for the data-dependent bound: "well" (A < 0 and the vertex |mu'| <= VT) vs worst case
    VT = 6.0
    a = np.abs(A)
    with np.errstate(divide="ignore", invalid="ignore"):
        mup = np.where(a > 0, B / (2 * a), np.inf)
    well = (A < 0) & (np.abs(mup) <= VT)
    M_well = np.where(well, a * mup ** 2, 0).sum(1)
    U_wc = np.where(~well, np.maximum(A, 0) * N1 ** 2 + np.abs(B) * N1, 0).sum(1)
    T_wc = np.where(~well, a * N1 ** 2 + np.abs(B) * N1, 0).sum(1)
    Cr = 8 * M_well + 3 * U_wc + T_wc         # T_j <= Cr - 3 f(j)
    # norm-based bound: T_j <= sum|A| n^2 + |B||n| <= max|A| R2 + ||B||_2 sqrt(R2)
    T_norm = a.max(1) * R2 + np.sqrt((B ** 2).sum(1)) * np.sqrt(R2)

    X = np.concatenate([cb ** 2, cb], 1).T            # [32, n]
    Xabs = np.concatenate([cb ** 2, np.abs(cb)], 1).T
    out = {}
    fmax = np.empty(rows)
    gmax = np.empty((rows, n // 64))
    Tmax_top = np.empty(rows)
    for r0 in range(0, rows, 1024):
        sl = slice(r0, min(rows, r0 + 1024))
        W = np.concatenate([A[sl], B[sl]], 1)
        f = W @ X
        fmax[sl] = f.max(1)
        gmax[sl] = f.reshape(f.shape[0], n // 64, 64).max(2)
        # the true per-code T_j of the best code (what an oracle bound would charge)
        Tj = np.abs(W) @ Xabs
        Tmax_top[sl] = Tj[np.arange(f.shape[0]), f.argmax(1)]
    print(f"== {name}: {rows} rows, N1 {N1:.2f}, R2 {R2:.1f};  sd median {np.median(sd):.3f}, share of coords with A >= 0: {(A >= 0).mean():.3f}, "
          f"well-class share {well.mean():.3f}")
    print(f"   T_old median {np.median(T_old):.1f};  Cr - 3 fmax median {np.median(Cr - 3 * fmax):.1f};  T_norm median {np.median(T_norm):.1f};  "
          f"true T of the best code median {np.median(Tmax_top):.1f}")

    def report(label, Ef):
        margin = 2.5 * (Ef + Er)
        within = gmax >= (fmax - margin)[:, None]
        cand = within.sum(1)
        listed = np.zeros(rows, bool)
        ls = []
        for per in (32, 64, 128):                            # record sets of 2048 / 4096 / 8192 codes
            sets = within.reshape(rows, -1, per).sum(2)
            ls.append(int((sets >= 4).any(1).sum()))
        listed = (within.reshape(rows, -1, 64).sum(2) >= 4).any(1)
        print(f"   {label:58s} margin/old {np.median(margin) / np.median(2.5 * (2450 * U * T_old + Er)):6.2f}   candidates/row {cand.mean():6.3f}   "
              f"p99 {np.percentile(cand, 99):5.0f}   undecided rows (sets of 32/64/128 groups) {ls}")
        out[label] = (cand.mean(), listed.mean())

    report("today: fp16 + fp8, 2450 u T_old", 2450 * U * T_old)
    report("split-bf16, 604 u T_old", 604 * U * T_old)
    for k in (16600, 8300):
        report(f"main only, {k} u T_old", k * U * T_old)
        Edd = k * U * np.maximum(Cr - 3 * fmax, 0) / (1 - 3 * k * U)
        report(f"main only, {k} u min(T_old, Cr - 3 fmax)", np.minimum(Edd, k * U * T_old))
        report(f"main only, {k} u min(T_old, Cr - 3 fmax, T_norm)", np.minimum(np.minimum(Edd, k * U * T_old), k * U * T_norm))
        report(f"main only, {k} u (oracle: true T of the best code)", k * U * Tmax_top)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=4)
    ap.add_argument("--rows", type=int, default=4096)
    args = ap.parse_args()
    mu, sd, cb = bench_rows(args.images)
    study("bench workload (random-init encoder)", mu, sd, cb)
    mu2, sd2 = trained_like_rows(args.rows)
    study("trained-VAE-like synthetic rows", mu2, sd2, cb)


if __name__ == "__main__":
    main()
