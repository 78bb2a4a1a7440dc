"""bench.py's `roofline.traffic` is read from the committed rocprofv3 summary it cites: the parse is pinned here, and so is
the file's presence -- a number in the bench line that no profile backs would be a pasted constant."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_traffic_comes_out_of_the_cited_profile():
    import bench
    path = os.path.join(ROOT, bench.TRAFFIC_PROFILE)
    assert os.path.exists(path), "the profile bench.py cites must be committed"
    got = bench.profile_traffic()
    assert got is not None
    text = open(path).read()
    fetch = re.search(r"== pmc_fetch.*?" + re.escape(bench.TRAFFIC_KERNEL) + r"\s+\(dispatches \d+\)\s+FETCH_SIZE\s+([\d.]+)", text, re.S)
    write = re.search(r"== pmc_write.*?" + re.escape(bench.TRAFFIC_KERNEL) + r"\s+\(dispatches \d+\)\s+WRITE_SIZE\s+([\d.]+)", text, re.S)
    assert (float(fetch.group(1)), float(write.group(1))) == got
    per_record = bench.measured_traffic_bytes_per_record()
    assert 646 <= per_record <= 1.25 * 646, per_record   # at least the algorithmic bytes, no re-reads


def test_bench_defaults_name_the_baseline_configs():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "78_125_000" in src and "BASELINE config 5" in src and "10_000_000" in src
    assert "--min-seconds" in src   # time-based warm-up
