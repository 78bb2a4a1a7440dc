import sys, time, zlib
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blazeseq_amd.parser import Context
from tests.gzip_util import DeviceGunzip, gzip_member
from oracle import oracle as O
data = O.generate_synthetic(100_000, 150, 150, 33, 73, "generic").tobytes()   # 31.8 MB
ctx = Context()
for name, level, strat in (("stored", 0, zlib.Z_DEFAULT_STRATEGY), ("fixed", 6, zlib.Z_FIXED), ("huffman_only", 6, zlib.Z_HUFFMAN_ONLY), ("default", 6, zlib.Z_DEFAULT_STRATEGY)):
    comp = gzip_member(data, level, strat)
    for rep in range(2):
        g = DeviceGunzip(ctx, len(data) + (1 << 20))
        t0 = time.perf_counter()
        out = g.decode(comp)
        dt = time.perf_counter() - t0
        if rep == 0:
            g.close()
    assert out == data
    st = g.dec.stats()
    hc = g.dec.set_option("host_calls", 0)
    print(f"{name}: {len(comp)/1e6:.1f} MB -> {len(data)/1e6:.1f} MB in {dt*1e3:.0f} ms = {len(data)/dt/1e6:.0f} MB/s; fallback_jobs {st.fallback_jobs}, host calls {hc}", flush=True)
    g.close()
