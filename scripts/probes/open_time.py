import sys, time, os, gzip
sys.path.insert(0, "/root/repo")
import numpy as np
import blazeseq_amd as B
from blazeseq_amd import _lib as L
from tests.gzip_util import sequencer_like
data, n = sequencer_like(8 << 20)
p = "/dev/shm/open_t.fastq.gz"
open(p, "wb").write(gzip.compress(data, 6))
c = B.Context(B.ParserConfig(), "generic", 4096, 0)
c.set_option("ingest_gpu_inflate", 1)
for rep in range(3):
    t0 = time.perf_counter(); ing = B.Ingest(c, p, chunk_bytes=256 << 20, n_threads=8); t1 = time.perf_counter()
    res = ing.next(0); t2 = time.perf_counter()
    ing.close(); t3 = time.perf_counter()
    print(f"open {1e3*(t1-t0):.1f} ms, first chunk {1e3*(t2-t1):.1f} ms, close {1e3*(t3-t2):.1f} ms")
os.remove(p)
