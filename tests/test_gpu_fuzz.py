"""Short runs of the randomised consistency tools (tools/fuzz_*.py) so that every `-m gpu` pass covers a fresh slice of
random scenes / rectangles / shards / splits; the tools were run with 700 + 400 cases when they were written."""
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tool,args,scale", [("fuzz_vs_oracle.py", ["40", "101"], "1"), ("fuzz_split_shard.py", ["40", "102"], "1"),
                                             ("fuzz_vs_oracle.py", ["20", "103"], "2"),      # vertices up to +-5, camera at 12: just inside the containment check's range
                                             ("fuzz_vs_oracle.py", ["12", "104"], "4"),      # camera at 24: the automatic mode falls back per call
                                             ("fuzz_vs_oracle.py", ["12", "105"], "100")])   # everything outside: reachability mode (fast tree) or, for LDS-resident scenes, the cull on rebuilt leaf boxes
def test_randomised_consistency(H, amd_lib, orc_lib, tool, args, scale):
    import os
    r = subprocess.run([sys.executable, str(H.ROOT / "tools" / tool)] + args, capture_output=True, text=True, timeout=600, env={**os.environ, "FUZZ_SCALE": scale})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout
    if scale == "100" :
        # outside the verified range nothing falls back to the replica any more: the fast tree with the reference's reachability replayed, or (LDS-resident
        # scenes) the reference tree with the leaf-box cull on leaf boxes rebuilt at the scene's rounding bound
        assert "'replica': 0" in r.stdout and "'fast': 0" not in r.stdout and "'cull': 0" not in r.stdout
