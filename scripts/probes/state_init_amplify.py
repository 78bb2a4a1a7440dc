"""Amplifier for the in-kernel state init mismatch: ONE ctx, the two streams of the campaign that shared the failing ctx (seeds
4101 and 2723 of tests/fuzz_campaign.py's kind "tiny": clean chunks of ~11-byte records whose record arrays overflow and are re-made),
parsed alternately as fast as the host can, no oracle in the loop (expected results computed once).  On a wrong result the state
snapshots of that very result are printed (query dump_state).
   python scripts/probes/state_init_amplify.py <mode> [seconds] [min_record_bytes]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from fastq_fuzz import rand_stream, rand_record
src = open(os.path.join(ROOT, "tests", "fuzz_campaign.py")).read()
g = {"np": np, "rand_stream": rand_stream, "rand_record": rand_record}
exec(compile(src[src.index("def make_stream(rng):"):src.index("ap = argparse.ArgumentParser()")], "fc", "exec"), g)
from gpu_util import make_pair
from oracle import oracle as O
from blazeseq_amd import _lib as L
mode = int(sys.argv[1]); seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 60
mrb = int(sys.argv[3]) if len(sys.argv) > 3 else 32
kw = dict(check_ascii=True, check_quality=True, quality_schema="sanger", views_only=True, buffer_capacity=64)
ctx, ocfg = make_pair(batch_size=100, single_pass=False, min_record_bytes=mrb, **kw)
ctx.set_option("state_init_in_kernel", mode)
streams = []
for seed in (4101, 2723):
    rng = np.random.default_rng(seed)
    data, kind = g["make_stream"](rng)
    data = np.frombuffer(bytes(data), dtype=np.uint8).copy()
    f = O.flat_parse(data, ocfg, is_eof=True)
    streams.append((seed, data, (f.n_records, f.term_code)))
t0 = time.time(); done = bad = 0
while time.time() - t0 < seconds:
    for seed, data, want in streams:
        r = ctx.parse(data, 0, True)
        got = (int(r.n_records), int(r.status))
        done += 1
        if got != want:
            bad += 1
            if bad <= 12:
                print(f"WRONG mode {mode} mrb {mrb} iteration {done} seed {seed}: got {got} want {want} error_record {int(r.error_record)} consumed {int(r.bytes_consumed)} newlines {int(r.total_newlines)}", flush=True)
                sys.stderr.flush(); L.lib().bzq_set_option(ctx.h, b"dump_state", 0)
print(f"state_init_amplify mode {mode} min_record_bytes {mrb}: {done} parses, {bad} wrong in {time.time() - t0:.0f} s", flush=True)
