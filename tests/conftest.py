import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (-m "not gpu") is simulator work, several hundred independent cases: when
    pytest-xdist is there and the caller did not choose (-n, -p no:xdist, BROTLI_AMD_TESTS_SERIAL=1)
    it is spread over the cores.  The GPU suite stays in one process (one device, large buffers).

    xdist WORKERS run this hook too (xdist/remote.py calls pytest_cmdline_main in every worker): a
    worker that asked for workers of its own would multiply without end, so the hook stands down
    in anything that is, or descends from, a distributed run — three independent signs of that."""
    if os.environ.get("PYTEST_XDIST_WORKER") or hasattr(config, "workerinput"):
        return None
    if os.environ.get("BROTLI_AMD_PYTEST_PARENT") or os.environ.get("BROTLI_AMD_TESTS_SERIAL"):
        return None
    if "not gpu" not in (config.option.markexpr or ""):
        return None
    if not config.pluginmanager.hasplugin("xdist") or getattr(config.option, "numprocesses", None) is not None:
        return None
    if getattr(config.option, "collectonly", False) or getattr(config.option, "usepdb", False):
        return None
    os.environ["BROTLI_AMD_PYTEST_PARENT"] = str(os.getpid())     # inherited by every worker
    # the checkers are built once, here, not by eight workers at the same time
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=False)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "simt")], check=False)
    config.option.numprocesses = max(1, min(8, os.cpu_count() or 1))
    config.option.dist = "load"
    return None


_CONFIG = []


def pytest_runtest_logstart(nodeid, location):
    """An abort inside a test (a GPU memory fault is a SIGABRT from the HSA runtime, not an exception) must still name
    its test: the id goes to the terminal reporter's stream, flushed, and — on a GPU box — to gpurun_out/last_test.txt, before the test runs."""
    try:
        tr = _CONFIG[0].pluginmanager.get_plugin("terminalreporter") if _CONFIG else None
        expr = (_CONFIG[0].option.markexpr or "") if _CONFIG else ""
        if tr is not None and "gpu" in expr and "not gpu" not in expr:
            # the terminal reporter writes to the real stdout (pytest's capture does not swallow it): the driver's
            # pytest.log ends with the id of the test that was running
            tr.ensure_newline()
            tr.write_line("[test start] %s" % nodeid)
            tr._tw.flush()
        if os.environ.get("GRAFT_REPO_ROOT") or os.path.exists("/dev/kfd"):
            d = os.path.join(ROOT, "gpurun_out")
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "last_test.txt"), "a") as f:
                f.write(nodeid + "\n")
    except OSError:
        pass


def pytest_configure(config):
    # no core files: a faulting GPU process of this suite maps tens of GiB, and a core of that size once filled the box's
    # disk and took the rest of the round's evidence with it (GPUTEST_r03)
    try:
        import resource
        resource.setrlimit(resource.RLIMIT_CORE, (0, 0))
    except (ImportError, ValueError, OSError):
        pass
    _CONFIG[:] = [config]
    # gpurun_out/last_test.txt names the test that was running when a GPU process died: one run's record, not a growing log
    if (os.environ.get("GRAFT_REPO_ROOT") or os.path.exists("/dev/kfd")) and not os.environ.get("PYTEST_XDIST_WORKER"):
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            open(os.path.join(ROOT, "gpurun_out", "last_test.txt"), "w").close()
        except OSError:
            pass
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
