"""bench.py's launch logic on CPU: `python bench.py --gpus N` (N > 1) outside a launcher starts its own ranks as a child
`python -m torch.distributed.run` and relays rank 0's ONE JSON line.  The GPU step is replaced by `--stub-step` (gloo, the bank
all-gather alone) so that the whole N > 1 flow -- self-launch, rendezvous on 127.0.0.1, barriers, rank reductions, fd-1
hygiene, exit-code propagation -- runs here; the measured variant is tests/test_gpu_bench_contract.py."""
import io
import json
import os
import subprocess
import sys

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_who_self_launches():
    assert bench.needs_self_launch(2, {}) is True
    assert bench.needs_self_launch(8, {"PATH": "x"}) is True
    assert bench.needs_self_launch(1, {}) is False
    assert bench.needs_self_launch(1, {"SEAM_BENCH_SELF_LAUNCH": "1"}) is True
    # already under a launcher (the driver's own torch.distributed.run command): never nest
    assert bench.needs_self_launch(8, {"WORLD_SIZE": "8", "RANK": "3"}) is False
    assert bench.needs_self_launch(2, {"RANK": "0"}) is False


def test_launch_command_is_the_drivers():
    cmd = bench.launch_command(["--gpus", "4", "--steps", "3"], 4, 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5] == os.path.join(ROOT, "bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "3"]


class _FakeChild:
    def __init__(self, text, rc):
        self.stdout, self.rc = io.StringIO(text), rc

    def wait(self):
        return self.rc


def _relay(text, rc, capsys):
    class A:
        gpus, stub_step = 2, False
    seen = {}

    def popen(cmd, **kw):
        seen["cmd"], seen["env"] = cmd, kw["env"]
        return _FakeChild(text, rc)
    got = bench.self_launch(A, ["--gpus", "2"], popen=popen)
    cap = capsys.readouterr()
    return got, cap.out, cap.err, seen


def test_relay_keeps_one_line_on_stdout(capsys):
    noise = 'RCCL version 2.22\n{"not": "the line"}\n'
    line = json.dumps({"metric": "video-clips/sec", "value": 1.0, "n_gpus": 2})
    rc, out, err, seen = _relay(noise + line + "\ntrailing\n", 0, capsys)
    assert rc == 0 and out == line + "\n"
    assert "RCCL version" in err and "trailing" in err and '{"not": "the line"}' in err
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "SEAM_BENCH_SELF_LAUNCH" not in seen["env"]


def test_relay_propagates_failure(capsys):
    rc, out, _, _ = _relay("boom\n", 7, capsys)
    assert rc == 7 and out == ""
    rc, out, err, _ = _relay("no line at all\n", 0, capsys)          # rc 0 but nothing to relay is a failure too
    assert rc == 1 and out == "" and "no result line" in err


def _run(args, extra_env=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, "bench.py"] + args, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=env)


def test_two_ranks_end_to_end_with_stubbed_step():
    """The real thing minus the GPU: `python bench.py --gpus 2` -> child torchrun -> 2 gloo ranks -> one line."""
    r = _run(["--gpus", "2", "--stub-step", "--workload", "c4", "--steps", "3", "--warmup", "1", "--clips", "2"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["data"].startswith("stub") and d["config"]["gallery"] == 50000
    assert d["bank_identical_on_all_ranks"] is True
    per = d["ms_per_step_per_rank"]
    assert len(per) == 2 and abs(max(per) - d["ms_per_step"]) < 1e-3          # MAX over ranks
    # whole-job aggregate: both ranks' clips over the slowest rank's time
    assert abs(d["value"] - 2 * 2 * 3 / (d["ms_per_step"] * 3 / 1e3)) / d["value"] < 1e-2
    assert "dp2" in d["config"]["parallelism"]


def test_self_launch_branch_with_one_rank():
    r = _run(["--gpus", "1", "--stub-step", "--steps", "2", "--warmup", "1"], {"SEAM_BENCH_SELF_LAUNCH": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 1
    assert "self-launch" in r.stderr


def test_rank_count_mismatch_is_an_error():
    r = _run(["--gpus", "3", "--stub-step", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "--gpus 3" in r.stderr


def test_two_ranks_refuse_to_time_a_diverging_bank():
    """A rank whose all-gathered bank differs from its peers' makes EVERY rank exit before the timed region, with a clear message
    and no result line (test hook SEAM_BENCH_TEST_CORRUPT_RANK, stub runs only)."""
    r = _run(["--gpus", "2", "--stub-step", "--workload", "c4", "--steps", "3", "--warmup", "1", "--clips", "2"],
             {"SEAM_BENCH_TEST_CORRUPT_RANK": "1"})
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert "differs between ranks after the warm-up" in r.stderr and "refusing to time" in r.stderr
