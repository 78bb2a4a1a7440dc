"""Stress of views mode on chunks of tiny records (tests/fuzz_campaign.py's kind "tiny": ~1000 newlines per tile, pool slots, record arrays
that overflow and are re-made), the same seeds and draws as the campaign, other kinds skipped: python scripts/probes/views_tiny_stress.py [seconds]"""
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
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120
t0, done, bad, pairs = time.time(), 0, 0, {}
for rnd in range(1000):
    for seed in range(0, 100000):
        if time.time() - t0 > seconds:
            break
        rng = np.random.default_rng(seed)
        if int(np.random.default_rng(seed).integers(0, 7)) != 2:
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
        key = (bs, tuple(sorted(kw.items())))
        if key not in pairs:
            if len(pairs) > 40:
                for c, _ in pairs.values(): c.close()
                pairs.clear()
            pairs[key] = make_pair(batch_size=bs, single_pass=False, **kw)
        ctx, ocfg = pairs[key]
        is_eof = bool(rng.random() < 0.85)
        try:
            check_views_against_oracle(ctx, ocfg, data, is_eof=is_eof, what=f"seed {seed} {kind}")
        except AssertionError as e:
            bad += 1
            print(f"MISMATCH round {rnd} seed={seed} n={len(data)} bs={bs} is_eof={is_eof} kw={kw} {str(e)[:200]}", flush=True)
        done += 1
    if time.time() - t0 > seconds:
        break
print(f"views tiny stress: {done} streams, {bad} mismatches in {time.time() - t0:.0f} s")
