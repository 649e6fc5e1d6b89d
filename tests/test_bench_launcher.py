"""bench.py --gpus N from a bare shell: the N ranks are started as a CHILD `python -m torch.distributed.run` before any GPU
call, rank 0's JSON line is relayed and the child's failure becomes the exit code. No GPU is needed to test that plumbing:
the child is replaced by small stand-in programs; the real child (two gloo ranks sharing the one GPU) runs in the GPU suite."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def test_launcher_command_is_the_drivers_contract():
    import bench
    args = bench.parse_args(["--gpus", "4", "--steps", "2", "--warmup", "1"])
    cmd = bench.launcher_command(args, ["--gpus", "4", "--steps", "2", "--warmup", "1"], 29123)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29123"
    i = cmd.index(str(ROOT / "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]


def _run_with_child(monkeypatch, capfd, child_source, tmp_path):
    import bench
    prog = tmp_path / "child.py"; prog.write_text(child_source)
    monkeypatch.setattr(bench, "launcher_command", lambda args, argv, port: [sys.executable, str(prog)] + list(argv))
    args = bench.parse_args(["--gpus", "2"])
    rc = bench.self_launch(args, ["--gpus", "2"])
    return rc, capfd.readouterr()


def test_self_launch_relays_the_result_line(monkeypatch, capfd, tmp_path):
    line = json.dumps({"metric": "Msamples/s", "value": 1.0, "n_gpus": 2})
    rc, io = _run_with_child(monkeypatch, capfd, f"import sys\nprint('noise from a rank')\nprint({line!r})\nprint('more noise')\n", tmp_path)
    assert rc == 0 and io.out.strip() == line


def test_self_launch_fails_loudly(monkeypatch, capfd, tmp_path):
    rc, io = _run_with_child(monkeypatch, capfd, "import sys\nsys.stderr.write('rank 1 died\\n')\nsys.exit(3)\n", tmp_path)
    assert rc == 3 and io.out == "" and "rank 1 died" in io.err and "exited with code 3" in io.err
    rc, io = _run_with_child(monkeypatch, capfd, "print('no json here')\n", tmp_path)
    assert rc == 1 and "printed no result line" in io.err


def test_self_launch_kills_a_hung_child_after_the_timeout(monkeypatch, capfd, tmp_path):
    """a rank stuck in the rendezvous must not hang the parent (and its whole process group goes with it)"""
    import bench, time
    prog = tmp_path / "child.py"; prog.write_text("import time\nprint('started', flush=True)\ntime.sleep(600)\n")
    monkeypatch.setattr(bench, "launcher_command", lambda args, argv, port: [sys.executable, str(prog)] + list(argv))
    args = bench.parse_args(["--gpus", "2", "--launch-timeout", "1.5"])
    t = time.time(); rc = bench.self_launch(args, ["--gpus", "2"]); dt = time.time() - t
    io = capfd.readouterr()
    assert rc == 124 and dt < 20 and "did not finish within" in io.err and io.out == ""


def test_a_signal_aimed_at_the_parent_takes_the_rank_group_down(tmp_path):
    """the ranks live in their own session: SIGTERM (what a driver's timeout sends) to the bench.py parent must kill their whole process group, or they would
    outlive it holding the GPUs; and if the parent is SIGKILLed, PR_SET_PDEATHSIG takes the launcher child down"""
    import os, signal, time
    for sig in (signal.SIGTERM, signal.SIGKILL):
        pidfile = tmp_path / f"pid{int(sig)}"
        child = tmp_path / "child.py"
        child.write_text(f"import os, time, subprocess, sys\nsub = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(600)'])\nopen({str(pidfile)!r}, 'w').write(f'{{os.getpid()}} {{sub.pid}}')\ntime.sleep(600)\n")
        parent = tmp_path / "parent.py"
        parent.write_text(f"import sys\nsys.path.insert(0, {str(ROOT)!r})\nimport bench\nbench.launcher_command = lambda args, argv, port: [sys.executable, {str(child)!r}]\n"
                          "args = bench.parse_args(['--gpus', '2'])\nsys.exit(bench.self_launch(args, ['--gpus', '2']))\n")
        p = subprocess.Popen([sys.executable, str(parent)], stderr=subprocess.PIPE, text=True)
        for _ in range(200):
            if pidfile.exists() and len(pidfile.read_text().split()) == 2:
                break
            time.sleep(0.05)
        launcher, grandchild = map(int, pidfile.read_text().split())
        p.send_signal(sig)
        p.wait(timeout=20)
        if sig == signal.SIGTERM:
            assert p.returncode == 128 + signal.SIGTERM and "process group was killed" in p.stderr.read()

        def alive(pid):
            try:
                os.kill(pid, 0)
            except ProcessLookupError:
                return False
            try:          # (a zombie still answers signal 0)
                return open(f"/proc/{pid}/stat").read().split(")")[-1].split()[0] != "Z"
            except OSError:
                return False
        deadline = time.time() + 10
        pids = (launcher, grandchild) if sig == signal.SIGTERM else (launcher,)          # (SIGKILL of the parent: the launcher dies by PDEATHSIG; its own children are torchrun's business)
        while time.time() < deadline and any(alive(q) for q in pids):
            time.sleep(0.05)
        assert not any(alive(q) for q in pids)
        if sig == signal.SIGKILL and alive(grandchild):
            os.kill(grandchild, signal.SIGKILL)          # the exact PID this test started


def test_bare_invocation_with_gpus_gt_1_never_touches_the_gpu_in_the_parent(tmp_path):
    """the parent must not import torch / the library before it has spawned the ranks: here (no GPU) the child fails, and the
    parent reports that failure instead of dying on its own GPU check"""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--dist-backend", "gloo", "--spp", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600)
    import torch
    if torch.cuda.device_count() == 0:
        assert r.returncode != 0 and "2-rank child exited" in r.stderr and "needs an MI355X" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_two_gloo_ranks_on_one_gpu_started_by_bench_itself():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--dist-backend", "gloo", "--spp", "16", "--check", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 2 and out["sharded_equals_unsharded"] is True and out["dist_backend"] == "gloo" and out["config"]["parallelism"] == "tiles%2"


def test_roofline_block_is_recomputable_from_the_committed_pmc_summary():
    """roofline.frac = the counter the bound names / kernel time / that resource's peak, from the newest profiles/rNN_pmc.json: VALU issue (SQ_INSTS_VALU against 1024 SIMDs x
    2.4 GHz / 2), the texture addresser (TA_TA_BUSY_sum against 256 x 2.4 GHz), the L1 <- L2 gather (TCP_TCC_READ_REQ_sum x 128 B against 18.8 TB/s) or the L2's fabric side
    (bytes against 8 TB/s); `bound` is the largest -- VALU issue compared through the ceiling of the kernel's dynamic instruction mix when the class counters are there"""
    import bench
    f = bench.pmc_file()
    assert f is not None and f.parent == ROOT / "profiles"
    pmc = json.loads(f.read_text())
    assert len({r["source_digest"] for r in pmc.values()}) == 1            # one profiling session, one set of kernel sources
    for key, rec in pmc.items():
        r = bench.roofline(key, rec["counters_per_launch"], rec["kernel_ms"], "cornell" in key, 1)
        t = rec["kernel_ms"] * 1e-3
        c = rec.get("pmc", {})
        fr = {"valu": (rec.get("SQ_INSTS_VALU") or c["SQ_INSTS_VALU"]) / t / 1e9 / (1024 * 2.4 / 2), "l2_fabric": rec["hbm_bytes_per_launch"] / t / 1e9 / 8000.0}
        if "TA_TA_BUSY_sum" in c:
            fr["ta"] = c["TA_TA_BUSY_sum"] / t / 1e9 / (256 * 2.4)
        if "TCP_TCC_READ_REQ_sum" in c:
            fr["l2_gather"] = c["TCP_TCC_READ_REQ_sum"] * 128.0 / t / 1e9 / 18800.0
        assert r["pmc_record"] == key and r["bound"] in fr and 0 < r["frac"] <= 1.02
        assert abs(r["frac"] - fr[r["bound"]]) < 1e-3, (key, r["bound"], r["frac"], fr)
        assert abs(r["valu_frac"] - fr["valu"]) < 1e-3 and 0 < r["lane_util"] <= 1 and r["algorithmic_gbs"] > 0
        mix = r.get("valu_mix_ceiling")
        assert mix and 0.4 < mix["frac_of_peak"] < 1.0 and r["valu_frac_of_mix_ceiling"] < 1.1, (key, mix)
        if "fast tree" in rec["traversal"]:                               # the global-memory kernels carry the gather evidence
            assert 0 < r["ta_busy_frac"] <= 1.02 and 0 < r["wave_wait_frac"] < 1 and 0 < r["l2_gather"]["frac"] < 1.2
    hk = bench.pmc_key("cornell_1080p_512spp", "auto", "simple", 8, 0)          # the headline launch's record (bench.DEFAULT_SPLIT = 0 resolves to 8 lanes per pixel for that frame on one GPU)
    head = bench.roofline(hk, pmc[hk]["counters_per_launch"], pmc[hk]["kernel_ms"], True, 1)
    assert head["bound"] == "valu" and 0.6 < head["frac"] < 0.8 and head["pmc_file"] == "profiles/" + f.name
