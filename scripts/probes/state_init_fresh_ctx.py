"""Second amplifier of the round-5 mismatch: what the campaign does and the one-ctx amplifier does not -- a FRESH ctx for
(nearly) every stream.  Loop: create a ctx (views mode, the failing stream's configuration), parse the campaign's failing stream once
or twice, compare with the expected result, close.  Environment BZQ_POOL_ZERO / BZQ_POOL_POISON select how bzq_create zeroes the views
pool's ticket (see bzq_create).   python scripts/probes/state_init_fresh_ctx.py <mode> [seconds] [parses_per_ctx]"""
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
per_ctx = int(sys.argv[3]) if len(sys.argv) > 3 else 1
kw = dict(check_ascii=True, check_quality=True, quality_schema="sanger", views_only=True, buffer_capacity=64)
ctx, ocfg = make_pair(batch_size=100, single_pass=False, **kw)
rng = np.random.default_rng(2723)
data, kind = g["make_stream"](rng)
data = np.frombuffer(bytes(data), dtype=np.uint8).copy()
f = O.flat_parse(data, ocfg, is_eof=True)
want = (f.n_records, f.term_code)
ctx.close()
# garbage in freshly freed device memory: small allocations of a new ctx then come back non-zero
junk = [torch.full((64,), 0x7F7F7F7F, dtype=torch.int32, device="cuda") for _ in range(256)]
torch.cuda.synchronize(); del junk; torch.cuda.empty_cache()
t0 = time.time(); made = done = bad = 0
while time.time() - t0 < seconds:
    ctx, _ = make_pair(batch_size=100, single_pass=False, **kw)
    ctx.set_option("state_init_in_kernel", mode)
    made += 1
    for k in range(per_ctx):
        r = ctx.parse(data, 0, True)
        got = (int(r.n_records), int(r.status))
        done += 1
        if got != want:
            bad += 1
            if bad <= 10:
                print(f"WRONG mode {mode} ctx {made} parse {k}: got {got} want {want} error_record {int(r.error_record)} consumed {int(r.bytes_consumed)} newlines {int(r.total_newlines)}", flush=True)
                sys.stderr.flush(); L.lib().bzq_set_option(ctx.h, b"dump_state", 0)
    ctx.close()
print(f"state_init_fresh_ctx mode {mode} POOL_ZERO={os.environ.get('BZQ_POOL_ZERO', 'stream')} POISON={os.environ.get('BZQ_POOL_POISON', '-')}: {made} ctxs, {done} parses, {bad} wrong in {time.time() - t0:.0f} s", flush=True)
