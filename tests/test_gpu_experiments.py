"""The single-launch variants (decoupled look-back, prefix-service workgroup, two-level look-back) and the
first-generation kernels are NOT in the product library: they live in libblazeseq_hip_exp.so (make -C blazeseq_amd/csrc
exp, -DBZQ_EXPERIMENTS=1) as independent implementations.  This test re-runs the batch-mode parity file against that
library with every variant switched on, in a process of its own; and checks that the product refuses the switches."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EXP = os.path.join(ROOT, "blazeseq_amd", "libblazeseq_hip_exp.so")


def test_product_library_has_no_experimental_variants():
    import blazeseq_amd as B
    from blazeseq_amd import _lib as L
    if os.environ.get("BLAZESEQ_HIP_LIB"):
        pytest.skip("a library override is active")
    ctx = B.Context(B.ParserConfig())
    for key, val in (("single_pass", 1), ("single_pass", 2), ("kernels_v2", 0)):
        assert L.lib().bzq_set_option(ctx.h, key.encode(), val) == L.ERR_ARG
    assert L.lib().bzq_set_option(ctx.h, b"experiments", 0) == L.ERR_ARG
    syms = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True).stdout
    raw = open(L.LIB_PATH, "rb").read()
    assert L.lib().bzq_set_option(ctx.h, b"inflate_ms", 1) == L.ERR_ARG
    for name in (b"k_single", b"k_tile_emit", b"lookback_", b"k_bgzf_inflate_ms"):   # (k_gz_shift: option early_find is part of the file pipeline since round 5)
        assert name not in raw, name
    assert "bzq_submit_chunk_device" in syms
    ctx.close()


def test_parity_of_every_variant_in_the_experiments_build():
    if not os.path.exists(EXP):
        pytest.skip("libblazeseq_hip_exp.so not built (make -C blazeseq_amd/csrc exp)")
    env = dict(os.environ, BLAZESEQ_HIP_LIB=EXP, BZQ_TEST_EXPERIMENTS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-4000:], r.stderr[-2000:])
    print(r.stdout.strip().splitlines()[-1])


def test_experimental_inflate_and_gzip_variants_in_the_experiments_build():
    """Eight BGZF blocks per wave (option inflate_ms): a measured loser that lives in the EXPERIMENTS library only; its parity tests
    run against it here (and the gzip staging matrix once more, against that build)."""
    if not os.path.exists(EXP):
        pytest.skip("libblazeseq_hip_exp.so not built (make -C blazeseq_amd/csrc exp)")
    env = dict(os.environ, BLAZESEQ_HIP_LIB=EXP, BZQ_TEST_EXPERIMENTS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_bgzf_inflate.py", "tests/test_gpu_gzip.py::test_the_next_piece_under_this_one",
                        "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", "-k", "eight_blocks_per_wave or next_piece"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-4000:], r.stderr[-2000:])
    print(r.stdout.strip().splitlines()[-1])
