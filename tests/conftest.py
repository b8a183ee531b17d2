import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "timeout: per-test limit (pytest-timeout; ignored when the plugin is absent)")
    # Build the test-only checkers if they are missing (seconds).
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=False)


@pytest.fixture(scope="session")
def ref():
    from refharness import Ref, have_ref
    if not have_ref():
        pytest.skip("oracle/_ref/libbrotli_ref.so not built")
    return Ref()


@pytest.fixture(scope="session")
def oracle():
    from refharness import Oracle
    return Oracle()
