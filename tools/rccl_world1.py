"""RCCL on a one-GPU lease: world size 1 through the same code the N-rank eval uses -- init_process_group("nccl") with a device id,
StepRecord.pack_with_psnr on the device, gather_step (ONE all_gather_into_tensor of the packed record), barrier, teardown
(tests/test_gpu_modules.py drives the same collective; this prints its latency).
It cannot show scaling (that needs N > 1 GPUs: the driver's SCALE tier); it shows that the RCCL path initialises and moves the record in
this image.  Prints one JSON line."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
from pit_hip.eval_dist import StepRecord, gather_step  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="env://", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dev = torch.device("cuda:0")
    bs, tokens = 16, 1024
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(bs, 3, 256, 256, generator=g) * 2 - 1).to(dev)
    xr = (x + 0.01 * torch.randn(x.shape, generator=g).to(dev)).clamp(-1, 1)
    idx = torch.randint(0, 65536, (bs, 1, 32, 32), generator=g).to(dev)
    lay = StepRecord(bs, tokens, n_metrics=1)
    times = []
    for it in range(20):
        rec = lay.pack_with_psnr(idx, x, xr)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        allrec = gather_step(rec, 1, always_collective=True)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    got_idx, got_met = lay.unpack(allrec[0])
    ok = bool(torch.equal(got_idx.reshape(-1).to(torch.int64), idx.reshape(-1))) and tuple(allrec.shape) == (1, rec.numel())
    dist.barrier()
    out = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "record_bytes": int(rec.numel() * 4),
           "gather_ms_p50": round(sorted(times)[len(times) // 2], 4), "record_round_trip_ok": ok,
           "what": "init_process_group(nccl, world 1) + 20 x all_gather_into_tensor of the packed per-step record on the device + barrier"}
    dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
