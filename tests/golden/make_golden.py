"""Generate the golden vectors under tests/golden/ from the REAL reference.

Run in the build container only (``python tests/golden/make_golden.py``): it
imports ``/root/reference`` (pit.quantization.*, pit.modules.unet), runs the
reference's CPU path (``backend="torch"``) and stores inputs + expected outputs
as small .npz / .json fixtures.  While doing so it checks the oracle
(oracle/gq_oracle.py + gq_oracle.c) bit-for-bit against the reference -- that
is the oracle's parity pin.  Nothing of the reference's source is stored; the
fixtures are data (seeds, tensors, hashes).

Versions used are recorded in tests/golden/meta.json.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))

from pit.modules.unet import Decoder as RefDecoder, Encoder as RefEncoder  # noqa: E402
from pit.quantization.gaussian import GaussianQuantRegularizer as RefGQ  # noqa: E402
from pit.quantization.gaussian import GaussianQuantRegularizer2 as RefGQ2  # noqa: E402
from pit.quantization.gaussian import prior_samples as ref_prior_samples  # noqa: E402
from pit.quantization.lfq import LFQQuantizer as RefLFQ  # noqa: E402
from pit.quantization.vq import VQQuantizer as RefVQ  # noqa: E402

from oracle import gq_oracle as O  # noqa: E402

torch.set_grad_enabled(False)


def sha(a) -> str:
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()[:16]


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def realistic_z(shape_c, b, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    mu = 0.9 * torch.randn(b, shape_c, h, w, generator=g)
    lv = -1.5 + 0.3 * torch.randn(b, shape_c, h, w, generator=g)
    return torch.cat([mu, lv], 1)


def kernel_boundary(ref, z):
    """Replays gaussian.py:64-81,122-123 to expose the (mu, std) rows the kernel sees."""
    b, c2, h, w = z.shape
    c = c2 // 2
    zf = z.reshape(b, c2, h * w).permute(0, 2, 1)
    mu, logvar = zf.chunk(2, 2)
    logvar = torch.clamp(logvar, ref.logvar_range[0], ref.logvar_range[1])
    std = torch.exp(0.5 * logvar)
    k = c // ref.group
    mu_r = mu.reshape(b, h * w, ref.group, k).permute(0, 1, 3, 2).reshape(-1, ref.group)
    std_r = std.reshape(b, h * w, ref.group, k).permute(0, 1, 3, 2).reshape(-1, ref.group)
    return mu_r.contiguous(), std_r.contiguous()


meta = {"torch": torch.__version__, "numpy": np.__version__, "cases": {}}
import scipy  # noqa: E402

meta["scipy"] = scipy.__version__

# ---------------------------------------------------------------- G1 codebooks
print("G1 codebooks")
g1 = {}
for dim in (4, 8, 16):
    cb = ref_prior_samples(65536, dim, 42).float().numpy()
    assert np.array_equal(cb, O.codebook(65536, dim, 42))
    nlp = torch.distributions.Normal(torch.zeros(1, dim), torch.ones(1, dim)).log_prob(torch.from_numpy(cb)).float().numpy()
    assert np.array_equal(nlp, O.nlp_table(cb)), "oracle nlp table differs from reference"
    g1[str(dim)] = {"cb_sha": sha(cb), "nlp_sha": sha(nlp), "absmax": float(np.abs(cb).max()),
                    "first_row": cb[0].tolist()}
    small = ref_prior_samples(1024, dim, 42).float().numpy()
    assert np.array_equal(small, cb[:1024]), "Sobol prefix property"
meta["cases"]["G1"] = g1

# ---------------------------------------------------------------- G2 kernel boundary
print("G2 kernel boundary (mu, std, log std) -> indices")
for dim, n, b, hw in ((16, 1024, 2, 16), (8, 1024, 2, 16), (4, 1024, 1, 16), (16, 65536, 1, 16)):
    t0 = time.time()
    ref = RefGQ("bchw", n, group=dim, backend="torch").eval()
    z = realistic_z(16, b, hw, hw, seed=100 + dim + (n > 1024))
    zh, info = ref(z)
    mu_r, std_r = kernel_boundary(ref, z)
    lsd = std_r.log()
    idx_rows = info["indices"].permute(0, 2, 3, 1).reshape(-1).numpy()
    # oracle parity pin at the kernel boundary
    oi, oz, best, second = O.argmax_rows(mu_r.numpy(), std_r.numpy(), ref.prior_samples.numpy(), 1.0,
                                         logstd=lsd.numpy(), with_gap=True)
    assert np.array_equal(oi, idx_rows), "oracle != reference at kernel boundary"
    # and at the module boundary
    ozh, oind = O.gq1_forward(z.numpy(), ref.prior_samples.numpy(), dim)
    assert np.array_equal(oind, info["indices"].numpy()) and np.array_equal(ozh, zh.numpy())
    assert np.array_equal(ref.dequant(info["indices"]).numpy(), zh.numpy())
    save(f"g2_dim{dim}_n{n}.npz", mu=mu_r.numpy(), std=std_r.numpy(), logstd=lsd.numpy(), indices=idx_rows,
         gap=(best - second).astype(np.float32))
    print(f"  dim {dim} n {n}: rows {len(idx_rows)} ok, min gap {float((best - second).min()):.3e}  ({time.time() - t0:.1f}s)")

# ---------------------------------------------------------------- G3 module boundary known answers
print("G3 module boundary (full 2^16 codebook)")
ref = RefGQ("bchw", 65536, group=16, backend="torch").eval()
g3 = {}
torch.manual_seed(0)
z_a = torch.randn(1, 32, 32, 32)
z_b = realistic_z(16, 1, 32, 32, seed=0)
for name, z in (("randn_seed0", z_a), ("realistic_seed0", z_b)):
    t0 = time.time()
    zh, info = ref(z)
    ind = info["indices"].numpy()
    ozh, oind = O.gq1_forward(z.numpy(), ref.prior_samples.numpy(), 16)
    assert np.array_equal(oind, ind) and np.array_equal(ozh, zh.numpy()), "oracle != reference (G3)"
    g3[name] = {"first8": ind.reshape(-1)[:8].tolist(), "indices_sha": sha(ind), "zhat_sha": sha(zh.numpy())}
    save(f"g3_{name}.npz", z=z.numpy(), indices=ind.astype(np.int32))
    print(f"  {name}: first8 {g3[name]['first8']}  ({time.time() - t0:.1f}s)")
meta["cases"]["G3"] = g3

# ---------------------------------------------------------------- G4 layouts (K > 1), blc, GQ2
print("G4 layouts")
z = realistic_z(16, 2, 8, 8, seed=4)
for group in (4, 8):
    r1 = RefGQ("bchw", 2048, group=group, backend="torch").eval()
    zh, info = r1(z)
    ozh, oind = O.gq1_forward(z.numpy(), r1.prior_samples.numpy(), group)
    assert np.array_equal(oind, info["indices"].numpy()) and np.array_equal(ozh, zh.numpy())
    assert np.array_equal(O.gq1_dequant(oind, r1.prior_samples.numpy(), group), r1.dequant(info["indices"]).numpy())
    save(f"g4_gq1_group{group}.npz", z=z.numpy(), indices=info["indices"].numpy().astype(np.int32), zhat=zh.numpy())
zb = z.reshape(2, 32, 64).permute(0, 2, 1).contiguous()  # blc
r1b = RefGQ("blc", 2048, group=4, backend="torch").eval()
zh, info = r1b(zb)
ozh, oind = O.gq1_forward(zb.numpy(), r1b.prior_samples.numpy(), 4, fmt="blc")
assert np.array_equal(oind, info["indices"].numpy()) and np.array_equal(ozh, zh.numpy())
save("g4_gq1_blc_group4.npz", z=zb.numpy(), indices=info["indices"].numpy().astype(np.int32), zhat=zh.numpy())
for dim_idx, zz in ((1, z), (2, zb)):
    r2 = RefGQ2(4, 2048, dim_idx=dim_idx, backend="torch").eval()
    zv, iv = r2.quant_vq(zz)
    ozv, oiv = O.gq2_quant_vq(zz.numpy(), r2.prior_samples.numpy(), 4, dim_idx)
    assert np.array_equal(oiv, iv["indices"].numpy()) and np.array_equal(ozv, zv.numpy())
    assert np.array_equal(O.gq2_dequant(oiv, r2.prior_samples.numpy(), 4, dim_idx), r2.dequant(iv["indices"]).numpy())
    save(f"g4_gq2_dimidx{dim_idx}.npz", z=zz.numpy(), indices=iv["indices"].numpy().astype(np.int32), zhat=zv.numpy())
g1k = np.load(os.path.join(HERE, "g4_gq1_group4.npz"))["indices"]
g2k = np.load(os.path.join(HERE, "g4_gq2_dimidx1.npz"))["indices"]
assert not np.array_equal(g1k, g2k), "strided and contiguous grouping must differ"

# ---------------------------------------------------------------- G5 ties / NaN / extremes
print("G5 ties, extreme sigma, far-away rows")
r5 = RefGQ("bchw", 2048, group=16, backend="torch").eval()
cb5 = r5.prior_samples.clone()
cb5[100] = cb5[7]
cb5[1500] = cb5[7]
r5.prior_samples.copy_(cb5)
r5.normal_log_prob.copy_(r5.normal_dist.log_prob(cb5).float())
z5 = realistic_z(16, 1, 8, 8, seed=5)
# (a NaN/inf mu or sigma <= 0 makes torch.distributions.Normal raise ValueError in the reference's
# torch backend -- argument validation -- so those rows are not reachable there; see tests/test_gpu_*)
z5[0, 16:, 0, 1] = -40.0               # below the logvar clamp
z5[0, 16:, 0, 2] = 30.0                # above the clamp
z5[0, :16, 0, 3] = cb5[7]              # sits exactly on the duplicated codeword
z5[0, 16:, 0, 3] = -6.0
z5[0, :16, 0, 4] = 1e4                 # far away
zh, info = r5(z5)
ozh, oind = O.gq1_forward(z5.numpy(), cb5.numpy(), 16)
assert np.array_equal(oind, info["indices"].numpy()), "oracle != reference on edge rows"
assert np.array_equal(ozh, zh.numpy(), equal_nan=True)
assert info["indices"][0, 0, 0, 3] == 7
mu_r, std_r = kernel_boundary(r5, z5)
save("g5_edges.npz", z=z5.numpy(), cb=cb5.numpy(), indices=info["indices"].numpy().astype(np.int32),
     mu=mu_r.numpy(), std=std_r.numpy(), logstd=std_r.log().numpy())

# ---------------------------------------------------------------- G6 VQ / LFQ
print("G6 VQ / LFQ")
torch.manual_seed(7)
vq = RefVQ("bchw", 4096, 16).eval()
vq.embedding.weight.data.normal_()  # non-degenerate synthetic codebook (hard parts)
g = torch.Generator().manual_seed(70)
zv = torch.randn(2, 16, 8, 8, generator=g)
zq, info = vq(zv)
ozq, oind, gap = O.vq_forward(zv.numpy(), vq.embedding.weight.numpy(), 1, with_gap=True)
clear = gap > 1e-4
assert np.array_equal(oind[clear], info["indices"].numpy()[clear]), "VQ oracle != reference on clear rows"
print(f"  VQ K=1: {int((~clear).sum())} near-tie rows of {clear.size}; all clear rows equal")
assert np.array_equal(O.vq_dequant(oind, vq.embedding.weight.numpy(), 1), vq.dequant(torch.from_numpy(oind)).numpy())
save("g6_vq_k1.npz", z=zv.numpy(), emb=vq.embedding.weight.numpy(), indices=info["indices"].numpy().astype(np.int32),
     gap=gap.astype(np.float32))
vq2 = RefVQ("bchw", 1024, 8, codebook_num=2).eval()
vq2.embedding.weight.data.normal_()
zq2, info2 = vq2(zv)
ozq2, oind2, gap2 = O.vq_forward(zv.numpy(), vq2.embedding.weight.numpy(), 2, with_gap=True)
clear2 = gap2 > 1e-4
assert np.array_equal(oind2[clear2], info2["indices"].numpy()[clear2])
assert np.array_equal(O.vq_dequant(oind2, vq2.embedding.weight.numpy(), 2), vq2.dequant(torch.from_numpy(oind2)).numpy())
save("g6_vq_k2.npz", z=zv.numpy(), emb=vq2.embedding.weight.numpy(), indices=info2["indices"].numpy().astype(np.int32),
     gap=gap2.astype(np.float32), zq=vq2.dequant(info2["indices"]).numpy())
lfq = RefLFQ("bchw", codebook_size=256, num_codebooks=2).eval()
zl = torch.randn(2, 16, 8, 8, generator=g)
zl[0, :, 0, 0] = 0.0       # x == 0 -> bit 0
zl[0, 5, 0, 1] = -0.0
ql, infol = lfq(zl)
oq, oil = O.lfq_forward(zl.numpy())
assert np.array_equal(oil, infol["indices"].numpy()) and np.array_equal(oq, ql.numpy())
assert np.array_equal(O.lfq_dequant(oil), lfq.dequant(infol["indices"]).numpy())
assert np.array_equal(lfq.dequant(infol["indices"]).numpy(), ql.numpy())
save("g6_lfq.npz", x=zl.numpy(), indices=infol["indices"].numpy().astype(np.int32), q=ql.numpy())

# ---------------------------------------------------------------- G7 encoder / decoder
print("G7 encoder / decoder (seeded init)")
from pit_hip.modules.unet import Decoder as MyDecoder, Encoder as MyEncoder  # noqa: E402

FULL = dict(attn_type="vanilla", double_z=True, z_channels=16, resolution=256, in_channels=3, out_ch=3, ch=128,
            ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[32], dropout=0.0)
SMALL = dict(attn_type="vanilla", double_z=True, z_channels=16, resolution=32, in_channels=3, out_ch=3, ch=32,
             ch_mult=[1, 2, 2], num_res_blocks=1, attn_resolutions=[8], dropout=0.0)


def sd_sha(sd):
    h = hashlib.sha256()
    for k in sd:
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k].numpy()).tobytes())
    return h.hexdigest()[:16]


g7 = {}
for tag, cfg, res in (("small", SMALL, 32), ("full", FULL, 256)):
    torch.manual_seed(1234)
    renc, rdec = RefEncoder(**cfg).eval(), RefDecoder(**cfg).eval()
    torch.manual_seed(1234)
    menc, mdec = MyEncoder(**cfg).eval(), MyDecoder(**cfg).eval()
    assert list(renc.state_dict()) == list(menc.state_dict()) and list(rdec.state_dict()) == list(mdec.state_dict())
    for a, bb in ((renc, menc), (rdec, mdec)):
        for k, v in a.state_dict().items():
            assert torch.equal(v, bb.state_dict()[k]), f"seeded init differs at {k}"
    gx = torch.Generator().manual_seed(1000)
    x = torch.rand(1, 3, res, res, generator=gx) * 2 - 1
    ze = renc(x)
    zlat = ze[:, :16] * 0.5
    xr = rdec(zlat)
    ze_m, xr_m = menc(x), mdec(zlat)
    e1, e2 = float((ze - ze_m).abs().max()), float((xr - xr_m).abs().max())
    print(f"  {tag}: enc keys {len(renc.state_dict())} dec keys {len(rdec.state_dict())}; "
          f"max|enc diff| {e1:.2e}, max|dec diff| {e2:.2e}")
    assert e1 < 1e-4 and e2 < 1e-4
    g7[tag] = {"enc_sd_sha": sd_sha(renc.state_dict()), "dec_sd_sha": sd_sha(rdec.state_dict()),
               "enc_keys": len(renc.state_dict()), "dec_keys": len(rdec.state_dict())}
    if tag == "small":
        save("g7_small.npz", x=x.numpy(), z_enc=ze.numpy(), z_lat=zlat.numpy(), x_rec=xr.numpy())
    else:
        # end-to-end with the real quantiser on the full config: x -> z_enc -> (zhat, indices) -> x_rec
        t0 = time.time()
        ref = RefGQ("bchw", 65536, group=16, backend="torch").eval()
        zh, info = ref(ze)
        xr2 = rdec(zh)
        mu_r, std_r = kernel_boundary(ref, ze)
        oi, _, best, second = O.argmax_rows(mu_r.numpy(), std_r.numpy(), ref.prior_samples.numpy(), 1.0,
                                            logstd=std_r.log().numpy(), with_gap=True)
        assert np.array_equal(oi, info["indices"].permute(0, 2, 3, 1).reshape(-1).numpy())
        save("g7_full_e2e.npz", z_enc=ze.numpy(), indices=info["indices"].numpy().astype(np.int32),
             gap=(best - second).astype(np.float32), x_rec=xr2.numpy().astype(np.float16),
             x_rec_stats=np.array([float(xr2.mean()), float(xr2.std()), float(xr2.abs().max())], np.float64))
        g7[tag]["e2e_indices_sha"] = sha(info["indices"].numpy())
        print(f"  full e2e: min gap {float((best - second).min()):.2e}, median {float(np.median(best - second)):.3f} "
              f"({time.time() - t0:.1f}s)")
meta["cases"]["G7"] = g7

# ---------------------------------------------------------------- G8 sharding index logic
print("G8 DistributedSampler / drop_last / re-interleave")
from torch.utils.data import DataLoader, Dataset  # noqa: E402
from torch.utils.data.distributed import DistributedSampler  # noqa: E402


class _Idx(Dataset):
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return i


g8 = []
for n, w, bs in ((37, 2, 4), (64, 8, 4), (100, 8, 16), (17, 4, 2), (5, 8, 1), (256, 8, 16)):
    per_rank = []
    for r in range(w):
        ds = _Idx(n)
        smp = DistributedSampler(ds, num_replicas=w, rank=r, shuffle=False)
        dl = DataLoader(ds, bs, shuffle=False, sampler=smp, drop_last=True)
        batches = [b.tolist() for b in dl]
        assert batches == O.eval_batches(n, w, r, bs), (n, w, r, bs)
        per_rank.append(batches)
    g8.append({"n": n, "world": w, "bs": bs, "per_rank": per_rank})
meta["cases"]["G8"] = g8

# ---------------------------------------------------------------- G10 BSQ / FSQ
print("G10 BSQ / FSQ")
from pit.quantization.bsq import BSQQuantizer as RefBSQ  # noqa: E402
from pit.quantization.fsq import FSQQuantizer as RefFSQ  # noqa: E402

gq10 = torch.Generator().manual_seed(110)
xb = torch.randn(2, 16, 8, 8, generator=gq10)
xb[0, :, 0, 0] = 0.0
bsq = RefBSQ("bchw", codebook_size=2, num_codebooks=16).eval()
qb, ib = bsq(xb)
oq, oi = O.bsq_forward(xb.numpy())
assert np.array_equal(oi, ib["indices"].numpy()) and np.array_equal(oq, qb.numpy())
assert np.array_equal(O.bsq_dequant(oi), bsq.dequant(ib["indices"]).numpy())
save("g10_bsq.npz", x=xb.numpy(), indices=ib["indices"].numpy().astype(np.int32), q=qb.numpy(),
     deq=bsq.dequant(ib["indices"]).numpy())
LEVELS = [8, 8, 8, 5, 5, 5]
xf = torch.randn(2, 6, 8, 8, generator=gq10) * 1.5
fsq = RefFSQ(LEVELS, "bchw").eval()
qf, inf_ = fsq(xf)
ozf, oif, marg = O.fsq_forward(xf.numpy(), LEVELS, with_margin=True)
assert np.array_equal(oif, inf_["indices"].numpy()) and np.array_equal(ozf, qf.numpy())
assert np.array_equal(O.fsq_dequant(oif, LEVELS), fsq.dequant(inf_["indices"]).numpy())
assert np.allclose(fsq.dequant(inf_["indices"]).numpy(), qf.numpy())
save("g10_fsq.npz", x=xf.numpy(), indices=inf_["indices"].numpy(), zhat=qf.numpy(), margin=marg,
     levels=np.array(LEVELS, np.int32))

# ---------------------------------------------------------------- G9 train-mode branch
print("G9 train-mode branch")
from pit_hip.quantization.gaussian import GaussianQuantRegularizer as MyGQ  # noqa: E402
from pit_hip.quantization.gaussian import GaussianQuantRegularizer2 as MyGQ2  # noqa: E402

g9 = {}
zt = realistic_z(16, 2, 8, 8, seed=9) * 1.7
for tag, ref_m, my_m in (("gq1", RefGQ("bchw", 1024, group=16, backend="torch"), MyGQ("bchw", 1024, group=16)),
                         ("gq2", RefGQ2(4, 1024, backend="torch"), MyGQ2(4, 1024))):
    steps = []
    outs = []
    for m in (ref_m, my_m):
        m.train()
        torch.manual_seed(123)
        rec = []
        for it in range(3):
            if tag == "gq1":
                zh, info = m(zt + 0.1 * it)
            else:
                zh, info = m.quant_gaussian(zt + 0.1 * it)
            rec.append({"kl_loss": float(info["kl_loss"]), "bits_mean": float(info["bits-mean"]),
                        "bits_min": float(info["bits-min"]), "bits_max": float(info["bits-max"]),
                        "lam": float(m.lam), "lam_min": float(m.lam_min), "lam_max": float(m.lam_max),
                        "zhat_sha": sha(zh.numpy())})
        outs.append(rec)
    assert outs[0] == outs[1], f"train branch of {tag} differs from the reference"
    g9[tag] = outs[0]
np.savez_compressed(os.path.join(HERE, "g9_train_z.npz"), z=zt.numpy())
meta["cases"]["G9"] = g9

with open(os.path.join(HERE, "meta.json"), "w") as f:
    json.dump(meta, f, indent=1)
print("meta.json written; all oracle-vs-reference checks passed")
