"""`bench.py --gpus N` starts N ranks itself when no launcher did (VERDICT r4 "missing" #1: the flag used to be parsed and
ignored, so `python bench.py --gpus 8` ran one rank and said n_gpus 1).  The argument plumbing with a stub launcher, on
the CPU: command line, forwarded JSON line, exit code, refusals.  The same code path with the real launcher on one GPU:
tests/test_train_infer_gpu.py::test_bench_spawns_its_ranks_through_the_launcher.
Reference: the ranks come from the launcher there too (config/config.yaml:45-46 `devices`, train.sh:6 `devices=[0,1]`)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STUB = r'''
import json, os, sys
print("RCCL version banner on stdout")                      # noise a real run may print before the line
code = int(os.environ.get("STUB_EXIT", "0"))
if os.environ.get("STUB_SILENT") != "1":
    print(json.dumps({"argv": sys.argv[1:], "spawned": os.environ.get("MRMT3_BENCH_SPAWNED"),
                      "ipc": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}))
sys.exit(code)
'''


@pytest.fixture
def bench():
    import importlib
    return importlib.import_module("bench")


def _run(bench, tmp_path, argv, n_visible, monkeypatch, **env):
    stub = tmp_path / "stub_launcher.py"
    stub.write_text(STUB)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    out = tmp_path / "line.json"
    fd = os.open(str(out), os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
    try:
        rc = bench.launch_ranks(bench.parse(argv), argv, launcher=[sys.executable, str(stub)], n_visible=n_visible, out_fd=fd)
    finally:
        os.close(fd)
    return rc, out.read_text()


def test_ranks_are_started_with_the_same_arguments_and_the_line_is_forwarded(bench, tmp_path, monkeypatch):
    argv = ["--gpus", "4", "--steps", "5", "--warmup", "2", "--spawn", "--no-inference"]
    rc, text = _run(bench, tmp_path, argv, 8, monkeypatch)
    assert rc == 0
    lines = text.splitlines()
    assert len(lines) == 1                                   # exactly one line reaches stdout, the JSON record
    d = json.loads(lines[0])
    assert d["argv"][0] == os.path.join(ROOT, "bench.py")
    assert d["argv"][1:] == ["--gpus", "4", "--steps", "5", "--warmup", "2", "--no-inference"]      # --spawn is the parent's
    assert d["spawned"] == "1" and d["ipc"] == "0"


def test_default_launcher_command_is_torch_distributed_run(bench, monkeypatch):
    seen = {}

    class R:
        returncode, stdout, stderr = 0, b'{"ok": 1}\n', b""

    def fake_run(cmd, **kw):
        seen["cmd"], seen["env"] = cmd, kw["env"]
        return R()
    monkeypatch.setattr(subprocess, "run", fake_run)
    argv = ["--gpus", "8"]
    r, w = os.pipe()
    try:
        assert bench.launch_ranks(bench.parse(argv), argv, n_visible=8, out_fd=w) == 0
        assert os.read(r, 100) == b'{"ok": 1}\n'
    finally:
        os.close(r), os.close(w)
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-3:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8"]


def test_more_ranks_than_visible_gpus_is_refused(bench, tmp_path, monkeypatch):
    rc, text = _run(bench, tmp_path, ["--gpus", "8"], 1, monkeypatch)
    assert rc == 2 and text == ""


def test_child_exit_code_and_a_missing_line_are_reported(bench, tmp_path, monkeypatch):
    rc, text = _run(bench, tmp_path, ["--gpus", "2"], 2, monkeypatch, STUB_EXIT="7")
    assert rc == 7 and json.loads(text)["argv"][1:] == ["--gpus", "2"]
    rc, text = _run(bench, tmp_path, ["--gpus", "2"], 2, monkeypatch, STUB_EXIT="0", STUB_SILENT="1")
    assert rc == 3 and text == ""


def test_ranks_that_die_without_a_line_still_leave_one_json_record(bench, tmp_path, monkeypatch):
    """the first multi-GPU run is one shot.  When the ranks exit non-zero without a result line (a rank other
    than 0 failed and the launcher took the rest down), the parent writes ONE JSON line with "error" and "stage", built from
    the per-rank `rank r: stage NAME` / `rank r: FAILED at stage NAME: why` lines the ranks put on stderr."""
    stub = tmp_path / "dying_launcher.py"
    stub.write_text(
        "import sys\n"
        "sys.stderr.write('rank 0: stage rccl_init done in 1.20 s\\nrank 1: stage rccl_init done in 1.21 s\\n')\n"
        "sys.stderr.write('rank 0: stage bucket_allreduce 5 buckets ok\\n')\n"
        "sys.stderr.write('rank 1: FAILED at stage bucket_allreduce: RuntimeError: bucket 2 (9961472 elements): all-reduce checksum\\n')\n"
        "sys.exit(1)\n")
    out = tmp_path / "line.json"
    fd = os.open(str(out), os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
    try:
        argv = ["--gpus", "2"]
        rc = bench.launch_ranks(bench.parse(argv), argv, launcher=[sys.executable, str(stub)], n_visible=2, out_fd=fd)
    finally:
        os.close(fd)
    assert rc == 1
    lines = out.read_text().splitlines()
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 2 and d["stage"] == "bucket_allreduce" and d["failed_rank"] == 1
    assert "checksum" in d["error"] and d["last_stage_per_rank"] == {"0": "bucket_allreduce", "1": "bucket_allreduce"}
    assert d["metric"] == bench.METRIC and d["returncode"] == 1
    # no stage line at all (killed before anything ran): the record says so
    d = bench.failure_record(8, "Killed\n", 137)
    assert d["stage"] == "launch" and d["failed_rank"] is None and "137" in d["error"]


def test_a_rank_that_fails_reports_its_stage_and_rank_zero_prints_the_error_line(monkeypatch, capsys):
    """Inside a rank: whatever stops it, its stage goes to stderr (`rank r: FAILED at stage S: why`) and rank 0 still prints a JSON
    line with "error" and "stage" — here the very first thing fails (no GPU in this container)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU (with one, the rank would go on to a rendezvous nobody answers)")
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env)
    assert r.returncode == 1
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] is None and d["stage"] == "start" and d["failed_rank"] == 0 and "no CPU fallback" in d["error"]
    assert "rank 0: FAILED at stage start" in r.stderr


def test_flag_and_launcher_must_agree_and_no_gpu_is_refused():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env)
    assert r.returncode == 2 and "must agree" in r.stderr and r.stdout == ""
    env = {k: v for k, v in os.environ.items() if k != "WORLD_SIZE"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, env=env)
    if r.returncode == 2:                                    # (this container: no GPU at all)
        assert "visible" in r.stderr and r.stdout == ""
