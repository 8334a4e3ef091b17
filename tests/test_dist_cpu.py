"""CPU, gloo, world_size 2: the image-sharded eval path -- sharding, the ONE packed
all_gather per step, and the rank-0 re-interleave into dataset order."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

G = os.path.join(os.path.dirname(__file__), "golden")
META = json.load(open(os.path.join(G, "meta.json")))


def test_shard_batches_match_torch_distributed_sampler_goldens():
    from pit_hip.eval_dist import shard_batches

    for case in META["cases"]["G8"]:
        for r in range(case["world"]):
            assert shard_batches(case["n"], case["world"], r, case["bs"]) == case["per_rank"][r]
    assert shard_batches(0, 2, 0, 4) == []


def test_step_record_roundtrip():
    from pit_hip.eval_dist import StepRecord

    lay = StepRecord(bs=3, tokens_per_image=5, n_metrics=2)  # odd token count -> padded pair
    g = torch.Generator().manual_seed(0)
    idx = torch.randint(0, 65536, (3, 1, 5), generator=g)
    idx[0, 0, 0], idx[2, 0, 4] = 65535, 0
    met = torch.randn(3, 2, generator=g)
    rec = lay.pack(idx, met)
    assert rec.dtype == torch.int32 and rec.numel() == lay.words == 6 + 8
    i2, m2 = lay.unpack(rec)
    assert torch.equal(i2, idx.reshape(3, 5)) and torch.equal(m2, met)


class _FakeModel:
    """encode: 'indices' derived from the image content so order mistakes are visible."""

    def encode(self, x, return_reg_log=True):
        ids = x[:, 0, 0, 0].round().long()
        tok = (ids[:, None] * 7 + torch.arange(4)[None]) % 65536
        return x, {"indices": tok.reshape(-1, 1, 2, 2)}

    def decode(self, z):
        return z * 0.5


def _images_for(ids):
    x = torch.zeros(len(ids), 3, 4, 4)
    for k, i in enumerate(ids):
        x[k] = float(i)
    return x


def _worker(rank, world, port, n, bs, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "vq-vae-from-gaussian-vae_amd"))
    from pit_hip.eval_dist import evaluate_sharded, init_from_env

    env = init_from_env("gloo")
    out = evaluate_sharded(_FakeModel(), _images_for, n, bs, env["rank"], env["world"], torch.device("cpu"), 4)
    if rank == 0:
        q.put({k: v.numpy() for k, v in out.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,bs", [(37, 4), (16, 2)])
def test_two_rank_gloo_gather_and_reinterleave(n, bs):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, bs, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    steps = -(-n // 2) // bs
    total = steps * bs * 2
    want_ids = np.arange(total) % n  # dataset order (the sampler wraps when it pads)
    want_tok = (want_ids[:, None] * 7 + np.arange(4)[None]) % 65536
    assert np.array_equal(out["indices"], want_tok)
    assert out["psnr"].shape == (total,)
