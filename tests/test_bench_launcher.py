"""`python3 bench.py --gpus N` is ONE command (the reference's harness is one command per run,
benchmark/throughput/run_throughput_benchmarks.sh:56-62): without a launcher in front of it bench.py starts the N ranks
itself.  The launcher is plain process plumbing, pinned here on the CPU with a stand-in script; the GPU twin
(tests/test_gpu_bench_harness.py) runs the real file through it."""
import json
import os
import subprocess
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import json, os, sys, time
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
    assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    mode = sys.argv[1]
    if mode == "rendezvous":   # a real world-N gloo rendezvous on the address / port the launcher handed out
        import torch, torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([rank + 1]); dist.all_reduce(t); dist.barrier(); dist.destroy_process_group()
        print(json.dumps({"rank": rank, "sum": int(t.item()), "argv": sys.argv[2:]}), flush=True)
    elif mode == "fail" and rank == int(sys.argv[2]):
        sys.exit(7)
    elif mode in ("fail", "hang"):
        time.sleep(600)   # (a rank stuck in a collective whose peer died)
""")


def _launch(tmp_path, nranks, argv, **kw):
    child = tmp_path / "child.py"
    child.write_text(CHILD)
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import bench; "
            f"sys.exit(bench.launch_ranks({nranks}, {argv!r}, script={str(child)!r}, **{kw!r}))")
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=300, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    return r, time.monotonic() - t0


def test_launcher_starts_n_ranks_and_only_rank_0_owns_stdout(tmp_path):
    r, _ = _launch(tmp_path, 3, ["rendezvous", "--gpus", "3", "--steps", "2"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = [json.loads(ln) for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert out == [{"rank": 0, "sum": 6, "argv": ["--gpus", "3", "--steps", "2"]}]   # ONE line, rank 0's; the arguments travel verbatim
    err = [json.loads(ln) for ln in r.stderr.decode().splitlines() if ln.startswith("{")]
    assert sorted(e["rank"] for e in err) == [1, 2] and all(e["sum"] == 6 for e in err)


def test_launcher_names_the_failing_rank_and_stops_the_others(tmp_path):
    r, dt = _launch(tmp_path, 3, ["fail", "1"])
    assert r.returncode == 7
    assert "rank 1 of 3" in r.stderr.decode() and "exited with code 7" in r.stderr.decode()
    assert dt < 60   # the sleeping ranks were stopped, not waited for


def test_launcher_deadline(tmp_path):
    r, dt = _launch(tmp_path, 2, ["hang"], deadline_s=2.0)
    assert r.returncode == 124 and "deadline" in r.stderr.decode() and "[0, 1]" in r.stderr.decode()
    assert dt < 60


def test_bench_refuses_a_world_that_is_not_gpus():
    """Under a launcher that set WORLD_SIZE, --gpus must agree with it (no silent mismatch of `n_gpus` in the line)."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and b"WORLD_SIZE is 2" in r.stderr


def test_a_stopped_launcher_takes_its_ranks_along(tmp_path):
    """A driver that times out kills the launcher: the ranks must not stay behind on their GPUs."""
    import signal
    child = tmp_path / "child.py"
    child.write_text("import os, sys, time\nopen(sys.argv[1] + os.environ['RANK'], 'w').write(str(os.getpid()))\ntime.sleep(600)\n")
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import bench; "
            f"sys.exit(bench.launch_ranks(2, [{str(tmp_path / 'pid')!r}], script={str(child)!r}))")
    p = subprocess.Popen([sys.executable, "-c", code], cwd=ROOT, stderr=subprocess.PIPE,
                         env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    t0 = time.monotonic()
    while not all((tmp_path / f"pid{r}").exists() and (tmp_path / f"pid{r}").read_text() for r in (0, 1)):
        assert time.monotonic() - t0 < 60 and p.poll() is None
        time.sleep(0.05)
    pids = [int((tmp_path / f"pid{r}").read_text()) for r in (0, 1)]
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=30) == 128 + signal.SIGTERM
    t0 = time.monotonic()
    alive = pids
    while alive and time.monotonic() - t0 < 20:
        alive = [q for q in alive if os.path.exists(f"/proc/{q}") and "Z" not in open(f"/proc/{q}/stat").read().split()[2]]
        time.sleep(0.1)
    assert not alive, alive
