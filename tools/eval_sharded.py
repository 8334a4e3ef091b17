#!/usr/bin/env python3
"""Image-sharded evaluation of the hot path (the part of the reference's eval.py this repo covers):
encode -> quantize -> decode per rank, ONE packed all_gather per step, rank 0 prints PSNR mean/std,
codebook usage and entropy of the gathered indices.

  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/eval_sharded.py \
      --base configs/sd3unet_gq_0.25.yaml [--ckpt model.ckpt] [--dataset DIR|list.txt] --img_size 256 --bs 16
Without --dataset a seeded synthetic image bank is used (no data offline)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
from pit_hip.eval_dist import codebook_usage, evaluate_sharded, init_from_env  # noqa: E402
from pit_hip.util import instantiate_from_config, load_config  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--base", required=True)
    ap.add_argument("--ckpt", default="")
    ap.add_argument("--dataset", default="")
    ap.add_argument("--img_size", type=int, default=256)
    ap.add_argument("--bs", type=int, default=1)
    ap.add_argument("--num", type=int, default=64, help="synthetic images when no --dataset is given")
    a = ap.parse_args()
    env = init_from_env(a.dist_backend)
    rank, world = env["rank"], env["world"]
    device = torch.device("cuda", env["local_rank"])
    cfg = load_config(a.base)
    torch.manual_seed(1234)
    model = instantiate_from_config(cfg["model"])
    if a.ckpt:
        model.init_from_ckpt(a.ckpt)   # autoencoder.py:313-329: strict=False, `loss.*` keys ignored
    model = model.eval().to(device)
    if device.type == "cuda":
        model = model.to(memory_format=torch.channels_last)   # NHWC: the fast conv stack (inputs are converted by the modules)
    if a.dataset:
        from pit_hip.data import SimpleDataset

        ds = SimpleDataset(a.dataset, a.img_size)
        n = len(ds)
        images_for = lambda ids: torch.stack([ds[i]["img"] for i in ids])  # noqa: E731
    else:
        n = a.num
        g = torch.Generator().manual_seed(1000)
        bank = torch.rand(n, 3, a.img_size, a.img_size, generator=g) * 2 - 1
        images_for = lambda ids: bank[ids]  # noqa: E731
    with torch.no_grad():  # tokens per image from a probe (depends on the config's downsampling and K)
        tokens = model.encode(images_for([0]).to(device), return_reg_log=True)[1]["indices"][0].numel()
    out = evaluate_sharded(model, images_for, n, a.bs, rank, world, device, tokens)
    if rank == 0 and out is not None:
        psnr = out["psnr"].float()
        print(f"PSNR: {psnr.mean():.4f} (±{psnr.std(unbiased=False):.4f})  over {psnr.numel()} images")
        n_codes = getattr(model.regularization, "n_samples", 65536)
        _, usage, ent = codebook_usage(out["indices"].to(device), n_codes)   # eval.py:137-141
        print(f"codebook usage: {usage:.4f}  entropy: {ent:.3f} bits")
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
