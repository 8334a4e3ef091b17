"""CPU: bench.py's self-launcher.  `python bench.py --gpus N` without RANK in the environment must
start N fresh rank processes with the env:// rendezvous variables (the reference is launched once per node:
Readme.md:119-126, eval.py:78-91), relay rank 0's JSON line and return the children's return code -- all without the
parent touching the GPU."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import bench

    return bench


def test_launcher_plan_argv_and_env():
    b = _bench()
    argv = ["--gpus", "4", "--steps", "7", "--warmup", "2", "--dist-backend", "gloo"]
    plan = b.launcher_plan(argv, 4, 23456, base_env={"PATH": "/usr/bin", "RANK": "stale"})
    assert len(plan) == 4
    for r, (cmd, env) in enumerate(plan):
        assert cmd[0] == sys.executable and cmd[1] == os.path.join(ROOT, "bench.py") and cmd[2:] == argv
        assert env["RANK"] == str(r) and env["LOCAL_RANK"] == str(r)
        assert env["WORLD_SIZE"] == "4" and env["LOCAL_WORLD_SIZE"] == "4"
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "23456"
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/usr/bin"


def _fake_plan(tmp_path, body):
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent(body))

    def plan(argv, n, port, base_env=None):
        return [([sys.executable, str(script)], dict(os.environ, RANK=str(r), WORLD_SIZE=str(n), MASTER_PORT=str(port)))
                for r in range(n)]

    return plan


def test_self_launch_relays_rank0_line_and_returns_zero(tmp_path, monkeypatch, capfd):
    b = _bench()
    monkeypatch.setattr(b, "launcher_plan", _fake_plan(tmp_path, """
        import json, os
        if os.environ["RANK"] == "0":
            print(json.dumps({"n_gpus": int(os.environ["WORLD_SIZE"]), "port": os.environ["MASTER_PORT"]}))
        else:
            print("chatter from rank", os.environ["RANK"])
    """))
    assert b.self_launch(["--gpus", "3"], 3) == 0
    out, err = capfd.readouterr()
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 3      # ONE JSON line on stdout
    assert "chatter from rank 1" in err and "chatter from rank 2" in err   # other ranks' stdout goes to stderr


def test_self_launch_propagates_failure_and_stops_the_other_ranks(tmp_path, monkeypatch, capfd):
    b = _bench()
    monkeypatch.setattr(b, "launcher_plan", _fake_plan(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(120)   # "blocked in a collective"
    """))
    import time

    t0 = time.time()
    assert b.self_launch(["--gpus", "2"], 2) == 7
    assert time.time() - t0 < 60
    assert "rank 1 exited with 7" in capfd.readouterr().err


def test_parent_does_not_import_torch_before_launching(tmp_path):
    """The parent's launch decision happens before `import torch` (so before any HIP call): run bench.py as a script
    with --gpus 2 in an environment where importing torch raises; the children fail the same way, the parent reports it."""
    poison = tmp_path / "torch.py"
    poison.write_text("raise ImportError('torch must not be imported by the launching parent')\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["PYTHONPATH"] = str(tmp_path)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0
    assert "stopping the other ranks" in p.stderr or "exited with" in p.stderr   # the PARENT got as far as launching


def test_rank_count_mismatch_fails_fast_with_a_message():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 2 and "torch.distributed.run" in p.stderr


def test_more_rccl_ranks_than_devices_exits_2_before_any_rendezvous_or_device_context():
    """the first real 8-GPU driver run must not die in argument handling -- and a box with fewer devices than
    `--gpus` must be told so at once: a rank of an 8-rank RCCL launch (RANK / WORLD_SIZE / MASTER_* set, as `python -m
    torch.distributed.run --nproc-per-node 8` sets them) on a box with fewer than 8 visible devices exits 2 with the explanatory
    message BEFORE init_process_group (nobody listens on the port: a rendezvous would hang) and before any HIP call
    (torch.cuda.device_count() does not create a context).  eval.py:78-107 is the reference's launch."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="8", LOCAL_WORLD_SIZE="8", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29999")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dist-backend", "nccl", "--steps", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2, (p.returncode, p.stderr[-400:])
    assert "needs 8 devices" in p.stderr and "--dist-backend gloo" in p.stderr
    assert p.stdout.strip() == ""


def test_bench_reads_the_committed_pmc_passes_per_shape():
    """roofline.traffic comes from the committed rocprofv3 PMC passes of the config's own shape (profiles/rNN/pmc_*_<config>[_<size>].csv;
    gq2_0.25 shares gq_0.25's: the same rows x codes x dim and filter instantiation); a shape without a committed pass reports None."""
    import bench as b

    for kernel, cfg in (("gq_filter_bf16_kernel", "gq_0.25"), ("gq_filter_bf16_kernel", "gq_0.50"), ("gq_filter_bf16_kernel", "gq2_0.25"),
                        ("gq_grid_kernel", "gq_1.00"), ("gq_filter_bf16_kernel", "vq_16_512")):
        traffic, prov = b.pmc_traffic(kernel, cfg)
        assert traffic and traffic > 1e6, (kernel, cfg)
        assert all(os.path.exists(os.path.join(b.ROOT, f)) for f in prov["files_sha256_16"]) and prov["launches_averaged"] >= 5
    assert b.pmc_traffic("gq_filter_bf16_kernel", "gq_0.25_512") == (None, None)
    assert b.pmc_traffic("gq_filter_bf16_kernel", None) == (None, None)
