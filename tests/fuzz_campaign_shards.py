"""Shard-protocol randomized campaign (by hand on a GPU box): a stream cut into 2..8 byte ranges ANYWHERE (inside
headers, '+' lines, right after '@', at newlines), every range parsed as a shard (bzq_shard_scan / plan_shards /
bzq_submit_shard; the RCCL halo exchange is a device-to-device copy on the one GPU), concatenated outputs == the
one-shot parse.   python tests/fuzz_campaign_shards.py [--seconds 180]"""
import argparse, os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
from fastq_fuzz import rand_stream
from oracle import oracle as O
from test_gpu_parity import _run_shards_on_one_gpu
from gpu_util import EXPERIMENTS

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=180)
args = ap.parse_args()
t0, done, seed = time.time(), 0, 50_000
while time.time() - t0 < args.seconds:
    seed += 1
    rng = np.random.default_rng(seed)
    max_len = int(rng.choice([10, 40, 150, 1000, 20000]))
    nrec = int(rng.integers(100, 3000)) if max_len <= 150 else (int(rng.integers(40, 300)) if max_len == 1000 else int(rng.integers(8, 40)))
    data = np.frombuffer(rand_stream(rng, n_records=nrec, max_len=max_len, dirty=0.0, tail=int(rng.choice([0, 0, 3])),
                                     crlf=bool(rng.random() < 0.15)), dtype=np.uint8)
    validate = bool(rng.random() < 0.5)
    kw = dict(check_ascii=True, check_quality=True) if validate else {}
    oc = O.make_config(**kw)
    whole = O.flat_parse(data, oc)
    if whole.term_code != O.EOF:
        continue
    P = int(rng.integers(2, 9))
    lo = 2 * (2 * max_len + 40) + 64     # every shard must hold at least one whole record
    if data.size < P * lo * 2:
        continue
    cuts = sorted(set(int(x) for x in rng.integers(lo, data.size - lo, P - 1)))
    cuts = [c for i, c in enumerate(cuts) if i == 0 or c - cuts[i - 1] > lo]
    mode = [True, False, 2, 3][int(rng.integers(0, 4))] if (EXPERIMENTS and rng.random() < 0.3) else False
    try:
        total, ids, seqs, quals, ends = _run_shards_on_one_gpu(data, cuts, oc, single_pass=mode, **kw)
        ok = (total == whole.n_records and np.array_equal(ids, whole.id_bytes) and np.array_equal(seqs, whole.seq_bytes)
              and np.array_equal(quals, whole.qual_bytes) and np.array_equal(ends, whole.ends))
        why = "" if ok else f"total {total} vs {whole.n_records}"
    except Exception as e:   # noqa: BLE001
        ok, why = False, repr(e)[:500]
    if not ok:
        print(f"MISMATCH seed={seed} n={data.size} max_len={max_len} cuts={cuts} mode={mode} validate={validate}: {why}")
        sys.exit(1)
    done += 1
print(f"shard campaign: {done} streams x 2..8 shards identical to the one-shot parse in {time.time()-t0:.0f} s")
