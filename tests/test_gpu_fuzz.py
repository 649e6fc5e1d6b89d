"""Short runs of the randomised consistency tools (tools/fuzz_*.py) so that every `-m gpu` pass covers a fresh slice of
random scenes / rectangles / shards / splits; the tools were run with 700 + 400 cases when they were written."""
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tool,args", [("fuzz_vs_oracle.py", ["40", "101"]), ("fuzz_split_shard.py", ["40", "102"])])
def test_randomised_consistency(H, amd_lib, orc_lib, tool, args):
    r = subprocess.run([sys.executable, str(H.ROOT / "tools" / tool)] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout
