"""bench.py's N > 1 harness, executed on a one-GPU box: torch.distributed.run starts N ranks, all of them
use device 0 (--ranks-on-one-gpu N), the library's shared-memory transport stands in for RCCL (which refuses two ranks on
one GPU) and gloo carries torch's own collectives.  Every `world > 1` / `rank > 0` branch of bench.py -- shard bounds with the
144-byte shift, communicator set-up and path agreement, warm-up agreement, config 5's workload text -- and bzq_shard_stitch
with real neighbours run here; what stays RCCL-only is the transport itself (ncclAllGather / grouped ncclSend / ncclRecv,
bzq_comm.hpp), covered at world 1 by tests/test_c_driver.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(nranks, extra, reads):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--ranks-on-one-gpu", str(nranks),
           "--reads", str(reads), "--steps", "3", "--warmup", "1", "--min-seconds", "0.05"] + extra
    r = subprocess.run(cmd, capture_output=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr.decode()[-4000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]   # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [2, 8])
def test_bench_sharded_harness_runs_with_real_neighbours(nranks):
    reads = 200_000
    out = _run(nranks, [], reads)
    assert out["n_gpus"] == nranks and out["steps"] == 3 and out["scaling"] == "weak" and out["unit"] == "GB/s"
    assert out["metric"].startswith("FASTQ GB/s + Mrecords/s (150 bp synthetic)")
    cfg = out["config"]
    assert "BASELINE config 5 shard shape" in cfg["workload"] and f"{reads * nranks} reads of one synthetic stream" in cfg["workload"]
    assert "cut 144 B into a record" in cfg["workload"]
    assert cfg["exchange"] == "native" and cfg["parallelism"] == f"byte-range shards x{nranks}"
    assert cfg["ranks_on_one_gpu"].startswith(f"{nranks} ranks share device 0")
    # value = the WHOLE stream's bytes per step: records per rank x ranks x record bytes / ms_per_step
    total_bytes = reads * nranks * cfg["record_bytes"]
    assert abs(out["value"] - total_bytes / (out["ms_per_step"] * 1e-3) / 1e9) <= 0.01 * out["value"] + 0.01
    assert abs(out["mrecords_per_s"] - reads * nranks / (out["ms_per_step"] * 1e-3) / 1e6) <= 0.01 * out["mrecords_per_s"] + 0.01
    assert "cpu_baseline" not in out and "views_mode" not in out   # N = 1 only
    # the library's own summary all-gather delivered every rank's row, and every rank said where it runs
    assert out["ranks_seen"] == nranks and len(out["ranks"]) == nranks
    assert sorted(r["rank"] for r in out["ranks"]) == list(range(nranks)) and all(r["device"] == 0 and r["ranks_seen"] == nranks for r in out["ranks"])
    assert out["roofline"]["frac"] > 0 and out["roofline"]["launches_per_step"] >= 1


@pytest.mark.gpu
def test_bench_sharded_harness_torch_cross_check_and_fasta():
    """The torch.distributed cross-check protocol (blazeseq_amd/sharded.py over gloo) and the FASTA byte-range shards through
    the same harness."""
    out = _run(2, ["--exchange", "torch"], 100_000)
    assert out["n_gpus"] == 2 and out["config"]["exchange"] == "torch"
    out = _run(3, ["--fasta"], 20_000)
    assert out["n_gpus"] == 3 and "byte-range shards over 3 rank(s)" in out["config"]["parallelism"]


@pytest.mark.gpu
def test_bench_from_file_mode_shards_one_file_across_the_ranks():
    """bench.py --from-file: the stream is ONE file on /dev/shm, every rank reads its own byte range of it per
    step (bzq_shard_read_range) and the ranks stitch; the line counts the whole stream's bytes per step."""
    reads = 150_000
    out = _run(3, ["--from-file", "--reader-threads", "3"], reads)
    assert out["n_gpus"] == 3 and out["config"]["exchange"] == "native"
    ff = out["from_file"]
    assert ff and abs(ff["file_gb"] - 3 * reads * out["config"]["record_bytes"] / 1e9) < 1e-3 and ff["reader_threads"] == 3
    assert not os.path.exists(ff["path"])   # removed again
    assert out["ranks_seen"] == 3 and all(r["reader_threads"] == 3 and r["reader_cpus_bound"] >= 0 for r in out["ranks"])
    total_bytes = reads * 3 * out["config"]["record_bytes"]
    assert abs(out["value"] - total_bytes / (out["ms_per_step"] * 1e-3) / 1e9) <= 0.01 * out["value"] + 0.01


@pytest.mark.gpu
def test_plain_command_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` WITHOUT torch.distributed.run in front -- bench.py starts the ranks itself
    (bench.launch_ranks), rank 0 prints the ONE line, every rank's row reached the library's summary all-gather."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--ranks-on-one-gpu", "2", "--reads", "200000",
           "--steps", "3", "--warmup", "1", "--min-seconds", "0.05"]
    r = subprocess.run(cmd, capture_output=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr.decode()[-4000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and sorted(x["rank"] for x in out["ranks"]) == [0, 1]
    assert out["config"]["exchange"] == "native" and out["steps"] == 3
    # a rank that dies takes the run down with its name (no JSON line): rank 1 is told to use a device that does not exist
    bad = subprocess.run(cmd + ["--fail-rank", "1"], capture_output=True, timeout=600, cwd=ROOT, env=env)
    assert bad.returncode != 0 and "rank 1 of 2" in bad.stderr.decode()
    assert not [ln for ln in bad.stdout.decode().splitlines() if ln.startswith("{")]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--views"], ["--synchronous"]])
def test_single_gpu_line_at_a_reduced_size(extra):
    """The default (N = 1) path of bench.py -- the pipelined timed loop (result k -> submit k+1 -> batches k), the synchronous loop
    beside it, the roofline object -- at a size that runs in seconds: the driver's bench run must not meet this code for the first time."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--reads", "300000", "--steps", "4", "--warmup", "1", "--min-seconds", "0.05",
           "--no-extra-modes", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, capture_output=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr.decode()[-4000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 4 and out["unit"] == "GB/s" and out["dtype"] == "u8" and out["value"] > 0
    assert abs(out["value"] - 300000 * out["config"]["record_bytes"] / (out["ms_per_step"] * 1e-3) / 1e9) <= 0.01 * out["value"] + 0.01
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1 and rf["avg_launch_ms"] > 0
    if "--synchronous" in extra:
        assert out["synchronous"] is None and out["step"].startswith("submit -> result")
    else:
        assert out["synchronous"]["ms_per_step"] > 0 and out["step"].startswith("result(k) -> submit(k+1)")
