"""The randomized parity campaigns (tests/fuzz_campaign*.py: every stream compared with the oracle, bit for bit) under
the driver: each generator runs from its fixed first seed for a fixed time budget in a process of its own.  By hand the
same scripts run for minutes (DESIGN.md quotes those totals); here they are the regression gate."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BUDGET = os.environ.get("BZQ_CAMPAIGN_SECONDS", "7")   # (8 campaigns: the GPU suite stays under six minutes; by hand they run for minutes each)

CAMPAIGNS = {
    "chunk_level": (["tests/fuzz_campaign.py", "--seconds", BUDGET], "all bit-identical to the oracle"),
    "chunk_level_views": (["tests/fuzz_campaign.py", "--views", "--seconds", BUDGET], "all bit-identical to the oracle"),
    "parser_level": (["tests/fuzz_campaign_parser.py", "--seconds", BUDGET], "identical"),
    "shards": (["tests/fuzz_campaign_shards.py", "--seconds", BUDGET], "identical to the one-shot parse"),
    "fasta": (["tests/fuzz_campaign_fasta.py", BUDGET, "1"], "identical"),
    "inflate": (["tests/fuzz_campaign_inflate.py", "--seconds", BUDGET], "identical to the bytes zlib compressed"),
    "gzip": (["tests/fuzz_campaign_gzip.py", "--seconds", BUDGET], "identical to the bytes zlib compressed"),
    "fasta_shards": (["tests/fuzz_campaign_fasta_shards.py", "--seconds", BUDGET], "identical to the sequential parse"),
}


@pytest.mark.parametrize("name", sorted(CAMPAIGNS))
def test_campaign(name):
    args, marker = CAMPAIGNS[name]
    r = subprocess.run([sys.executable, *args], cwd=ROOT, capture_output=True, text=True, timeout=float(BUDGET) * 6 + 240)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert marker in r.stdout, r.stdout[-2000:]
    print(r.stdout.strip().splitlines()[-1])
