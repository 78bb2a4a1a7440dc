"""A fresh process per run over the reference's 3 GiB / 100 bp workload (bench.py process_mode's plain file) at several ingest chunk sizes:
what a one-file process pays for its device arenas (hipMalloc of fresh memory: 30-80 ms per GiB) against what small chunks cost in
steady state.   python scripts/process_chunk_sweep.py [runs]"""
import os, re, statistics, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
exe = os.path.join(ROOT, "tests", "c_driver", "bzq_throughput")
path = "/dev/shm/bzq_chunk_sweep.fastq"
data = O.generate_synthetic(14_700_000, 100, 100, 33, 73, "generic")
data.tofile(path)
n = data.size
del data
env = dict(os.environ, BZQ_THROUGHPUT_TIMES="1")
try:
    for chunk in (256, 128, 96, 64, 32):
        walls, parts = [], []
        for it in range(3 + runs):
            t0 = time.perf_counter()
            r = subprocess.run([exe, path, "batches", str(chunk), "8"], capture_output=True, text=True, env=env, timeout=120)
            dt = time.perf_counter() - t0
            assert r.returncode == 0, r.stderr[-300:]
            if it >= 3:
                walls.append(dt)
                m = re.search(r"create ([\d.]+) ms, open ([\d.]+) ms, first chunk ([\d.]+) ms, remaining \d+ chunks ([\d.]+) ms, close\+destroy ([\d.]+) ms, main total ([\d.]+) ms", r.stderr)
                if m:
                    parts.append([float(x) for x in m.groups()])
        mean = statistics.fmean(walls)
        pm = [round(statistics.fmean(p[i] for p in parts), 1) for i in range(6)]
        print(f"chunk {chunk:3d} MiB: {n / mean / 1e9:5.2f} GB/s whole process (mean {mean * 1e3:.0f} ms, min {min(walls) * 1e3:.0f}, stdev {statistics.pstdev(walls) * 1e3:.0f}); "
              f"create {pm[0]} open {pm[1]} first {pm[2]} rest {pm[3]} close {pm[4]} main {pm[5]} outside {round(mean * 1e3 - pm[5], 1)}", flush=True)
finally:
    os.remove(path)
