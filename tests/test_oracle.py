"""CPU: the oracle (oracle/gq_oracle.{c,py}) against the golden vectors captured from the
reference by tests/golden/make_golden.py.  Integer/index results are compared bit-exactly."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import gq_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")
META = json.load(open(os.path.join(G, "meta.json")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def load(name):
    return np.load(os.path.join(G, name))


@pytest.mark.parametrize("dim", [4, 8, 16])
def test_g1_codebook_and_nlp_hashes(dim):
    want = META["cases"]["G1"][str(dim)]
    cb = O.codebook(65536, dim, 42)
    assert sha(cb) == want["cb_sha"]
    assert sha(O.nlp_table(cb)) == want["nlp_sha"]
    assert float(np.abs(cb).max()) == want["absmax"]
    assert np.array_equal(O.codebook(1024, dim, 42), cb[:1024])  # Sobol prefix


@pytest.mark.parametrize("name,dim,n", [("g2_dim16_n1024", 16, 1024), ("g2_dim8_n1024", 8, 1024),
                                        ("g2_dim4_n1024", 4, 1024), ("g2_dim16_n65536", 16, 65536)])
def test_g2_kernel_boundary(name, dim, n):
    d = load(name + ".npz")
    cb = O.codebook(n, dim, 42)
    idx, zhat, best, second = O.argmax_rows(d["mu"], d["std"], cb, 1.0, logstd=d["logstd"], with_gap=True)
    assert np.array_equal(idx, d["indices"])
    assert np.array_equal(zhat, cb[d["indices"]])
    np.testing.assert_array_equal((best - second).astype(np.float32), d["gap"])
    # the logstd the reference derived == torch-CPU log here (same torch build) -- informational pin
    assert np.array_equal(O.torch_log(d["std"]), d["logstd"])


def test_g2_score_matrix_argmax_consistency():
    d = load("g2_dim16_n1024.npz")
    cb = O.codebook(1024, 16, 42)
    S = O.score_matrix(d["mu"][:64], d["std"][:64], cb, logstd=d["logstd"][:64])
    assert np.array_equal(S.argmax(1), d["indices"][:64])


@pytest.mark.parametrize("name", ["randn_seed0", "realistic_seed0"])
def test_g3_module_boundary_known_answers(name):
    d = load(f"g3_{name}.npz")
    want = META["cases"]["G3"][name]
    cb = O.codebook(65536, 16, 42)
    zhat, ind = O.gq1_forward(d["z"], cb, 16)
    assert ind.reshape(-1)[:8].tolist() == want["first8"]
    assert np.array_equal(ind, d["indices"])
    assert sha(ind) == want["indices_sha"] and sha(zhat) == want["zhat_sha"]
    assert np.array_equal(O.gq1_dequant(ind, cb, 16), zhat)  # dequant round trip


@pytest.mark.parametrize("name", ["g15_e2e_trained_like", "g17_e2e_nonsquare_trained_like"])
def test_trained_operating_point_goldens_pin_the_oracle(name):
    """The end-to-end goldens of round 4 (reference Encoder -> GaussianQuantRegularizer -> Decoder on CPU, checkpoint-like weights,
    z at ~17.8 bits per group; g17: three 192x320 images): the oracle's module forward on the golden z gives the reference's
    indices (pit/quantization/gaussian.py:62-81,142-150), and the stored top-2 gaps are the oracle's."""
    d = load(name + ".npz")
    cb = O.codebook(65536, 16, 42)
    zhat, ind = O.gq1_forward(d["z_enc"], cb, 16)
    assert np.array_equal(ind, d["indices"])
    assert float(d["gap"].min()) > 0.0 and d["gap"].size == d["indices"].size


def test_g4_layouts():
    cb4, cb8 = O.codebook(2048, 4, 42), O.codebook(2048, 8, 42)
    for name, cb, group in (("g4_gq1_group4", cb4, 4), ("g4_gq1_group8", cb8, 8)):
        d = load(name + ".npz")
        zhat, ind = O.gq1_forward(d["z"], cb, group)
        assert np.array_equal(ind, d["indices"]) and np.array_equal(zhat, d["zhat"])
        assert np.array_equal(O.gq1_dequant(ind, cb, group), d["zhat"])
    d = load("g4_gq1_blc_group4.npz")
    zhat, ind = O.gq1_forward(d["z"], cb4, 4, fmt="blc")
    assert np.array_equal(ind, d["indices"]) and np.array_equal(zhat, d["zhat"])
    for dim_idx in (1, 2):
        d = load(f"g4_gq2_dimidx{dim_idx}.npz")
        zhat, ind = O.gq2_quant_vq(d["z"], cb4, 4, dim_idx)
        assert np.array_equal(ind, d["indices"]) and np.array_equal(zhat, d["zhat"])
        assert np.array_equal(O.gq2_dequant(ind, cb4, 4, dim_idx), d["zhat"])
    a, b = load("g4_gq1_group4.npz")["indices"], load("g4_gq2_dimidx1.npz")["indices"]
    assert a.shape == b.shape and not np.array_equal(a, b)  # strided vs contiguous grouping


def test_g5_edges():
    d = load("g5_edges.npz")
    zhat, ind = O.gq1_forward(d["z"], d["cb"], 16)
    assert np.array_equal(ind, d["indices"])
    assert ind[0, 0, 0, 3] == 7  # duplicated codeword: first index wins
    idx, _ = O.argmax_rows(d["mu"], d["std"], d["cb"], 1.0, logstd=d["logstd"])
    assert np.array_equal(idx, d["indices"].transpose(0, 2, 3, 1).reshape(-1))


def test_argmax_semantics_nan_and_ties():
    # torch.argmax: first maximum wins; NaN counts as the maximum, first NaN wins
    cb = O.codebook(64, 4, 42).copy()
    cb[20] = cb[3]
    mu = cb[[3, 5]].copy()
    sd = np.full((2, 4), 0.1, np.float32)
    idx, _ = O.argmax_rows(mu, sd, cb)
    assert idx.tolist() == [3, 5]
    mu[1, 0] = np.nan
    idx, _ = O.argmax_rows(mu, sd, cb)
    assert idx.tolist() == [3, 0]


def test_g6_vq_lfq():
    d = load("g6_vq_k1.npz")
    zq, ind, gap = O.vq_forward(d["z"], d["emb"], 1, with_gap=True)
    clear = gap > 1e-4
    assert np.array_equal(ind[clear], d["indices"][clear])
    d = load("g6_vq_k2.npz")
    zq, ind, gap = O.vq_forward(d["z"], d["emb"], 2, with_gap=True)
    clear = gap > 1e-4
    assert np.array_equal(ind[clear], d["indices"][clear])
    assert np.array_equal(O.vq_dequant(d["indices"].astype(np.int64), d["emb"], 2), d["zq"])
    d = load("g6_lfq.npz")
    q, ind = O.lfq_forward(d["x"])
    assert np.array_equal(ind, d["indices"]) and np.array_equal(q, d["q"])
    assert np.array_equal(O.lfq_dequant(ind), d["q"])
    assert ind[0, 0, 0, 0] == 0  # x == 0 -> every bit 0


def test_g8_sharding_index_logic():
    for case in META["cases"]["G8"]:
        n, w, bs = case["n"], case["world"], case["bs"]
        per_rank = [O.eval_batches(n, w, r, bs) for r in range(w)]
        assert per_rank == case["per_rank"]
        flat = [[i for b in pr for i in b] for pr in per_rank]
        order = O.reinterleave(flat)
        steps = len(per_rank[0])
        # restored order is the dataset order (the sampler wraps where it pads)
        assert order == [j % n for j in range(steps * bs * w)]


def test_cuda_formula_equals_twice_the_torch_score_up_to_row_constant():
    d = load("g2_dim16_n1024.npz")
    cb = O.codebook(1024, 16, 42)
    mu, sd = d["mu"][:32], d["std"][:32]
    a = O.cuda_formula_scores(mu, sd, cb, 1.0).astype(np.float64)
    b = O.score_matrix(mu, sd, cb, logstd=d["logstd"][:32]).astype(np.float64)
    diff = a - 2 * b
    assert np.all(np.abs(diff - diff[:, :1]) < 2e-3)  # out = 2*s + const(r)
    assert np.array_equal(a.argmax(1), b.argmax(1))


def test_g10_bsq_fsq():
    d = load("g10_bsq.npz")
    q, ind = O.bsq_forward(d["x"])
    assert np.array_equal(ind, d["indices"]) and np.array_equal(q, d["q"])
    assert np.array_equal(O.bsq_dequant(ind), d["deq"])
    d = load("g10_fsq.npz")
    zhat, ind, margin = O.fsq_forward(d["x"], d["levels"].tolist(), with_margin=True)
    assert np.array_equal(ind, d["indices"]) and np.array_equal(zhat, d["zhat"])
    assert np.array_equal(O.fsq_dequant(ind, d["levels"].tolist()), d["zhat"])


def test_torch_restatement_matches_reference_goldens():
    """oracle/gq_torch_ref.py (the reference's backend="torch" arithmetic, timed by bench.py's cpu_baseline)
    reproduces the reference-captured indices and the C oracle on the kernel-boundary golden, and the g7 full-path
    golden's first rows at the module boundary."""
    import torch

    from oracle import gq_torch_ref as T

    d = load("g2_dim16_n1024.npz")
    cb = O.codebook(1024, 16, 42)
    idx, zhat = T.argmax_rows(torch.from_numpy(d["mu"]), torch.from_numpy(d["std"]), torch.from_numpy(cb))
    assert np.array_equal(idx.numpy(), d["indices"])
    assert np.array_equal(zhat.numpy(), cb[d["indices"]])
    assert np.array_equal(T.normal_log_prob(torch.from_numpy(cb)).numpy(), O.nlp_table(cb))
    # module boundary, strided grouping (K = 2), small codebook: torch restatement == C oracle
    g = np.random.default_rng(5)
    z = np.concatenate([0.9 * g.standard_normal((1, 16, 4, 4)), -1.5 + 0.3 * g.standard_normal((1, 16, 4, 4))],
                       axis=1).astype(np.float32)
    cb8 = O.codebook(2048, 8, 42)
    zt, it = T.gq1_forward(torch.from_numpy(z), torch.from_numpy(cb8), 8)
    zo, io = O.gq1_forward(z, cb8, 8)
    assert np.array_equal(it.numpy(), io) and np.array_equal(zt.numpy(), zo)


def test_oracle_entry_points_under_asan_and_ubsan(tmp_path):
    """`make -C oracle san` -- the C restatement + a driver over every entry point (120 shape
    combinations: dims that are not multiples of 8, one row, one code, non-finite rows, ties) built with
    -fsanitize=address,undefined and run on the CPU.  (The host half of libgqhip has the same kind of target:
    `make -C vq-vae-from-gaussian-vae_amd/csrc san`, ~2 min, run on demand.)"""
    import os
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in ("gq_oracle.c", "san_driver.c", "Makefile"):
        shutil.copy(os.path.join(root, "oracle", f), tmp_path / f)
    out = subprocess.run(["make", "-C", str(tmp_path), "san"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok: 120 shape combinations" in out.stdout, out.stdout + out.stderr
