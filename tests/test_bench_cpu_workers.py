"""bench.py's CPU-baseline worker processes (round 5): the protocol the parent relies on -- a worker pins itself, builds the CPU restatement, prints
"ready", does NOTHING until a line arrives on stdin, then runs whole images for --cpu-seconds and prints one JSON line -- and the aggregation.  The
workers import the oracle (they ARE the checker timed as a baseline: bench.cpu_baseline's leg); no GPU is involved."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_worker_waits_for_go_then_reports_images():
    env = dict(os.environ, CASAPOSE_CPU_WORKER_THREADS="2", OMP_NUM_THREADS="2", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-worker", "0/1", "--cpu-seconds", "1.5", "--height", "64", "--width", "96"],
                         stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    try:
        assert p.stdout.readline().strip() == "ready"
        time.sleep(0.5)
        assert p.poll() is None          # still waiting: nothing runs before the parent says so
        p.stdin.write("go\n")
        p.stdin.flush()
        out, err = p.communicate(timeout=300)
    finally:
        if p.poll() is None:
            p.kill()
    res = json.loads(out.strip().splitlines()[-1])
    assert res["images"] >= 1 and res["seconds"] >= 1.5 and res["worker"] == 0 and res["of"] == 1, (res, err[-500:])


def test_spawn_wait_collect_adds_up_two_workers(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench

    monkeypatch.setattr(bench, "CPU_WORKERS", 2)
    monkeypatch.setattr(bench, "CPU_WORKER_THREADS", 2)
    monkeypatch.setenv("CASAPOSE_CPU_WORKER_THREADS", "2")

    class A:
        cpu_seconds, height, width = 1.0, 64, 96

    ready = bench.wait_cpu_workers(bench.spawn_cpu_workers(A))
    assert len(ready) == 2
    res = bench.collect_cpu_workers(ready, 1.0)
    assert res["processes"] == 2 and res["images"] == sum(res["per_process_images"]) >= 2
    assert abs(res["images_per_s"] - res["images"] / res["seconds"]) < 0.01 * res["images_per_s"] + 1e-3


def test_worker_shape_follows_the_cgroup_quota(monkeypatch):
    """256 visible logical CPUs with a 16-CPU cgroup quota (the GPU box) -> 8 x 4; no quota -> one worker per 8 physical cores, at most 16; the
    environment overrides either."""
    import bench
    monkeypatch.delenv("CASAPOSE_CPU_WORKERS", raising=False)
    monkeypatch.delenv("CASAPOSE_CPU_WORKER_THREADS", raising=False)
    monkeypatch.setattr(bench.os, "cpu_count", lambda: 256)
    monkeypatch.setattr(bench, "cpu_quota", lambda: 16.0)
    assert bench.cpu_worker_shape() == (8, 4)
    monkeypatch.setattr(bench, "cpu_quota", lambda: None)
    assert bench.cpu_worker_shape() == (16, 8)
    monkeypatch.setattr(bench, "cpu_quota", lambda: 200.0)   # a quota above the physical cores does not bind
    assert bench.cpu_worker_shape() == (16, 8)
    monkeypatch.setattr(bench.os, "cpu_count", lambda: 8)
    monkeypatch.setattr(bench, "cpu_quota", lambda: None)
    assert bench.cpu_worker_shape() == (1, 8)
    monkeypatch.setenv("CASAPOSE_CPU_WORKERS", "3")
    monkeypatch.setenv("CASAPOSE_CPU_WORKER_THREADS", "2")
    assert bench.cpu_worker_shape() == (3, 2)


def test_cpu_quota_reads_cgroup_v2(monkeypatch, tmp_path):
    import builtins
    import bench
    real_open = builtins.open
    files = {"/sys/fs/cgroup/cpu.max": "1600000 100000\n"}

    def fake_open(path, *a, **k):
        if path in files:
            f = tmp_path / "f"
            f.write_text(files[path])
            return real_open(f, *a, **k)
        if isinstance(path, str) and path.startswith("/sys/fs/cgroup/"):
            raise FileNotFoundError(path)
        return real_open(path, *a, **k)

    monkeypatch.setattr(builtins, "open", fake_open)
    assert bench.cpu_quota() == 16.0
    files["/sys/fs/cgroup/cpu.max"] = "max 100000\n"
    assert bench.cpu_quota() is None
    del files["/sys/fs/cgroup/cpu.max"]
    assert bench.cpu_quota() is None
