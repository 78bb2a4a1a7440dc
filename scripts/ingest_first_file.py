"""Where the first file of a process spends its time (bench.py ingest_mode: first_file_of_the_process): per-call wall clock of
bzq_ingest_open / every bzq_ingest_next / close for the first and the following opens of one process.
usage: python scripts/ingest_first_file.py [GB]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import blazeseq_amd as B
from blazeseq_amd import _lib as L

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 3.2
dev = torch.device("cuda:0")
ctx0 = B.Context(B.ParserConfig(), "generic", 4096, 0, min_record_bytes=256)
n = int(gb * 1e9) // 318
nb = ctx0.generate_synthetic_device(n, 150, 33, 73, "generic")
buf = torch.empty(nb + 64, dtype=torch.uint8, device=dev)
ctx0.generate_synthetic_device(n, 150, 33, 73, "generic", d_out=buf.data_ptr(), cap=buf.numel())
torch.cuda.synchronize()
host = buf[:nb].cpu().numpy()
path = "/dev/shm/bzq_first_%d.fastq" % os.getpid()
host.tofile(path)
try:
    for rnd in range(4):
        if rnd == 2:   # the file read once more by this process before the run: is it the file's pages or the buffers?
            np.fromfile(path, dtype=np.uint8).sum()
            for key in ('pin_cache_bytes', 'dev_cache_bytes'):
                ctx0.set_option(key, 0); ctx0.set_option(key, 8 << 30)
        ctx = B.Context(B.ParserConfig(), "generic", 4096, 0, min_record_bytes=256)
        t0 = time.perf_counter()
        ing = B.Ingest(ctx, path, chunk_bytes=256 << 20, n_threads=8)
        t1 = time.perf_counter()
        laps, taken = [], 0
        while True:
            ta = time.perf_counter()
            r = ing.next(taken)
            taken = int(r.n_records)
            laps.append((time.perf_counter() - ta) * 1e3)
            if int(r.status) != L.OK:
                break
        t2 = time.perf_counter()
        st = ing.stats()
        ing.close()
        t3 = time.perf_counter()
        ctx.close()
        print(f"run {rnd} (fresh ctx): open {1e3*(t1-t0):.1f} ms, next calls {' '.join('%.1f' % x for x in laps)} = {1e3*(t2-t1):.1f} ms, close {1e3*(t3-t2):.1f} ms, total {1e3*(t3-t0):.1f} ms = {host.size/(t3-t0)/1e9:.1f} GB/s; reader threads inside pread {st.read_s*1e3:.0f} ms, consumer waiting {st.wait_s*1e3:.0f} ms")
finally:
    os.remove(path)
