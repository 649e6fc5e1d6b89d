import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def H():
    import harness
    return harness


@pytest.fixture(scope="session")
def orc_lib(H):
    H.build_oracle()
    return H.lib("orc")


@pytest.fixture(scope="session")
def amd_lib(H):
    from terra_amd import build
    build.build()
    return H.lib("amd")


@pytest.fixture(scope="session")
def ref_lib(H):
    if not H.have_reference():
        pytest.skip("/root/reference not present (only in the build container)")
    H.build_reference()
    return H.lib("ref")


@pytest.fixture()
def libm_mode(H, orc_lib):
    H.set_oracle_math(0)
    yield
    H.set_oracle_math(0)


@pytest.fixture()
def devmath_mode(H, orc_lib):
    H.set_oracle_math(1)
    yield
    H.set_oracle_math(0)
