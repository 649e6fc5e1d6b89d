"""A short slice of tools/fuzz_oracle_vs_reference.py (random soups, diffuse/Phong mixes, all integrators, tonemaps,
accumulating passes): the oracle reproduces the compiled reference bit for bit. 400 cases were run when it was written."""
import subprocess
import sys


def test_random_scenes_oracle_equals_reference(H, ref_lib, orc_lib):
    r = subprocess.run([sys.executable, str(H.ROOT / "tools" / "fuzz_oracle_vs_reference.py"), "60", "77"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "60 cases, 0 mismatches" in r.stdout
