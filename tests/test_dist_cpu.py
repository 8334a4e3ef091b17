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


def _run_world(world, n, bs):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, bs, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return out


@pytest.mark.parametrize("n,bs", [(43, 2), (61, 3), (5, 1)])
def test_eight_rank_gloo_gather_and_reinterleave(n, bs):
    """The rank count of BASELINE configs[2] (eval.py:78-107 on 8 GPUs), N not divisible by 8 bs.  (43, 2): the sampler pads 43 -> 48 by
    wrapping to the first samples and the wrapped items are INSIDE the kept batches; (61, 3): every rank gets 8 items and the loader's
    drop_last drops its ragged third batch (the wrapped items with it); (5, 1): fewer images than ranks.  Restored order
    [j % W][j // W] (eval.py:213-214) must be dataset order with the wrap."""
    from pit_hip.eval_dist import shard_batches

    world = 8
    out = _run_world(world, n, bs)
    steps = -(-n // world) // bs
    assert steps == len(shard_batches(n, world, 0, bs)) >= 1
    total = steps * bs * world
    want_ids = np.arange(total) % n
    want_tok = (want_ids[:, None] * 7 + np.arange(4)[None]) % 65536
    assert np.array_equal(out["indices"], want_tok)
    assert out["psnr"].shape == (total,)
    if (n, bs) == (43, 2):
        assert total == 48 and list(want_ids[43:]) == [0, 1, 2, 3, 4]          # the wrap is inside what is kept
    if (n, bs) == (61, 3):
        assert total == 48 < 64                                                  # drop_last bit: 2 of every rank's 8 items


@pytest.mark.parametrize("n,bs", [(37, 4), (16, 2)])
def test_two_rank_gloo_gather_and_reinterleave(n, bs):
    out = _run_world(2, n, bs)
    steps = -(-n // 2) // bs
    total = steps * bs * 2
    want_ids = np.arange(total) % n  # dataset order (the sampler wraps when it pads)
    want_tok = (want_ids[:, None] * 7 + np.arange(4)[None]) % 65536
    assert np.array_equal(out["indices"], want_tok)
    assert out["psnr"].shape == (total,)
