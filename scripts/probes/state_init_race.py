"""Hunt for round 5's timing-dependent mismatch: views mode on chunks of tiny records with the chunk state's initial values
(a) copied in on the side stream (the product), (b) written by workgroup 0 of k_scan_reduce (option state_init_in_kernel = 1 / 2).
Same seeds and draws as tests/fuzz_campaign.py's kind "tiny".  On a mismatch the same stream is replayed 20 times on the same ctx
(persistent = a wrong value that stays; transient = timing) and the result's fields are printed.
    python scripts/probes/state_init_race.py <mode> [seconds] [host_delay_us]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from fastq_fuzz import rand_stream, rand_record
src = open(os.path.join(ROOT, "tests", "fuzz_campaign.py")).read()
g = {"np": np, "rand_stream": rand_stream, "rand_record": rand_record}
exec(compile(src[src.index("def make_stream(rng):"):src.index("ap = argparse.ArgumentParser()")], "fc", "exec"), g)
from gpu_util import make_pair, check_views_against_oracle
from oracle import oracle as O
mode = int(sys.argv[1])
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 120
t0, done, bad, pairs, remakes = time.time(), 0, 0, {}, 0
cache = {}
for rnd in range(1000):
    for seed in range(0, 100000):
        if time.time() - t0 > seconds:
            break
        if seed in cache:
            item = cache[seed]
            if item is None:
                continue
        else:
            rng = np.random.default_rng(seed)
            if int(np.random.default_rng(seed).integers(0, 7)) != 2:
                cache[seed] = None
                continue
            data, kind = g["make_stream"](rng)
            kw = {}
            if rng.random() < 0.4:
                kw.update(check_ascii=True, check_quality=bool(rng.random() < 0.7))
                if rng.random() < 0.5:
                    kw["quality_schema"] = str(rng.choice(["sanger", "solexa", "illumina_1.3", "illumina_1.5", "illumina_1.8"]))
            rng.random()
            kw["views_only"] = True
            if rng.random() < 0.2:
                kw["buffer_capacity"] = int(rng.choice([64, 256, 4096, 65536]))
            if rng.random() < 0.15:
                kw["compat_simd_width"] = int(rng.choice([16, 32, 64]))
            bs = int(rng.choice([1, 7, 100, 256, 300, 4096]))
            is_eof = bool(rng.random() < 0.85)
            item = cache[seed] = (np.frombuffer(bytes(data), dtype=np.uint8).copy(), kind, kw, bs, is_eof) if seed < 6000 else None
            if item is None:
                continue
        data, kind, kw, bs, is_eof = item
        key = (bs, tuple(sorted(kw.items())))
        if key not in pairs:
            if len(pairs) > 40:
                for c, _ in pairs.values(): c.close()
                pairs.clear()
            pairs[key] = make_pair(batch_size=bs, single_pass=False, **kw)
            pairs[key][0].set_option("state_init_in_kernel", mode)
        ctx, ocfg = pairs[key]
        try:
            check_views_against_oracle(ctx, ocfg, data, is_eof=is_eof, what=f"seed {seed} {kind}")
        except AssertionError as e:
            bad += 1
            print(f"MISMATCH mode {mode} round {rnd} seed={seed} n={len(data)} bs={bs} is_eof={is_eof} kw={kw} {str(e)[:300]}", flush=True)
            f = O.flat_parse(data, ocfg, is_eof=is_eof)
            again = []
            for _ in range(20):
                r = ctx.parse(data, 0, is_eof)
                again.append((int(r.n_records), int(r.status)))
            print(f"   oracle ({f.n_records}, {f.term_code}); 20 replays on the same ctx: {sorted(set(again))} wrong {sum(1 for a in again if a != (f.n_records, f.term_code))}", flush=True)
        done += 1
    if time.time() - t0 > seconds:
        break
print(f"state_init_race mode {mode}: {done} streams, {bad} mismatches in {time.time() - t0:.0f} s", flush=True)
