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
    per_record, why = bench.measured_traffic_bytes_per_record()
    assert why is None and 646 <= per_record <= 1.25 * 646, per_record   # at least the algorithmic bytes, no re-reads


def test_a_stale_profile_is_refused():
    """the traffic figure belongs to the kernel that was profiled.  A run whose dominant kernel takes more
    than 10 % longer or shorter than the profile's kernel-trace duration gets no traffic figure and a reason instead."""
    import bench
    prof_ms = bench.profile_kernel_avg_ms()
    assert prof_ms is not None and 0.3 < prof_ms < 5.0, prof_ms   # (the cited summary holds the kernel-trace section too)
    ok, why = bench.measured_traffic_bytes_per_record(prof_ms * 1.05)
    assert ok is not None and why is None
    for f in (0.85, 1.2):
        stale, why = bench.measured_traffic_bytes_per_record(prof_ms * f)
        assert stale is None and "stale profile" in why


def test_bench_defaults_name_the_baseline_configs():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "78_125_000" in src and "BASELINE config 5" in src and "10_000_000" in src
    assert "--min-seconds" in src   # time-based warm-up
    # configs 3 and 4 and the CPU views() baseline ride in the default line; the knob is named
    for key in ('"validated_mode"', '"long_reads_mode"', '"cpu_baseline_views"', '"min_record_bytes"', "--ranks-on-one-gpu"):
        assert key in src, key
