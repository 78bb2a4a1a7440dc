"""The CPU twins of the device-side consumers (oracle/bzq_oracle.c: orc_nw_score, orc_pipeline_run) against a line-for-line Python
restatement of the reference's example kernel (examples/nw_gpu/kernels.mojo:21-89) and a plain per-record count of the quality
bytes (the v0.1 quality_distribution example, CHANGELOG.md:73).  They are the checker of bzq_batch_nw_scores_dev /
bzq_batch_quality_by_position_acc (tests/test_gpu_consumers.py) and the host figure beside bench.py's pipeline_mode."""
import numpy as np
import pytest

from oracle import oracle as O

REF_40BP = b"ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT"   # examples/nw_gpu/execution.mojo:36


def nw_kernel_restated(ref: bytes, query: bytes) -> int:
    ref_len, query_len = len(ref), len(query)
    if query_len > 256 or ref_len > 256:          # kernels.mojo:47-50
        return 0
    prev = [-i for i in range(ref_len + 1)]       # 60-62
    for j in range(1, query_len + 1):             # 65-86
        curr = [-j] + [0] * ref_len
        for i in range(1, ref_len + 1):
            diag = prev[i - 1] + (1 if ref[i - 1] == query[j - 1] else -1)
            curr[i] = max(diag, prev[i] - 1, curr[i - 1] - 1)
        prev = curr
    return prev[ref_len]                          # 88


def test_known_scores():
    assert O.nw_score(b"ACGT", b"ACGT") == 4                  # four matches
    assert O.nw_score(b"ACGT", b"") == -4                     # the first row: gap * ref_len
    assert O.nw_score(b"", b"ACG") == -3
    assert O.nw_score(b"ACGT", b"AGT") == 2                   # three matches, one gap
    assert O.nw_score(b"A" * 257, b"A") == 0 and O.nw_score(b"A", b"A" * 257) == 0
    assert O.nw_score(b"A" * 256, b"A" * 256) == 256


@pytest.mark.parametrize("seed", range(4))
def test_nw_score_equals_the_restated_kernel(seed):
    rng = np.random.default_rng(seed)
    for _ in range(60):
        ref = bytes(rng.choice(list(b"ACGTN"), int(rng.integers(0, 70))).astype(np.uint8))
        q = bytes(rng.choice(list(b"ACGT"), int(rng.integers(0, 200))).astype(np.uint8))
        assert O.nw_score(ref, q) == nw_kernel_restated(ref, q), (ref, q)


@pytest.mark.parametrize("bs", [7, 4096])
def test_pipeline_run_equals_per_record_work(bs):
    data = O.generate_synthetic(700, 5, 190, 0, 40, "sanger")
    cfg = O.make_config(buffer_capacity=64 * 1024, batch_size=bs)
    n, counts, score_sum = O.pipeline_run(data, cfg, REF_40BP, 150)
    f = O.flat_parse(data, O.make_config())
    assert n == f.n_records == 700
    want = np.zeros((150, 128), dtype=np.uint64)
    ss = 0
    e0 = 0
    for r in range(n):
        e1 = int(f.ends[r])
        ss += nw_kernel_restated(REF_40BP, f.seq_bytes[e0:e1].tobytes())
        q = f.qual_bytes[e0:e1][:150]
        want[np.arange(q.size), np.minimum(q, 127)] += 1
        e0 = e1
    assert score_sum == ss
    np.testing.assert_array_equal(counts, want)
    assert int(counts.sum()) == int(np.minimum(np.diff(np.concatenate([[0], f.ends])), 150).sum())
