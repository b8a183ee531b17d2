"""bench.py's line: printed as soon as the timed region is over, printed again as the optional parts arrive, never
lost to what runs behind it (the driver's round-4 run ended without a line).  bench.GpuJob is replaced
by tests/bench_stub.StubJob (the oracle stands in for the device); no GPU needed."""
import json
import os
import signal
import subprocess
import sys
import time

import pytest

from refharness import have_oracle, have_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import bench; from bench_stub import StubJob; "
          "bench.GpuJob = StubJob; bench.main(sys.argv[1:])" % (ROOT, os.path.join(ROOT, "tests")))
ARGS = ["--size-mb", "2", "--shard-kb", "128", "--steps", "3", "--warmup", "1"]
HEADLINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline")

pytestmark = pytest.mark.skipif(not have_oracle(), reason="oracle/liboracle.so not built")


def check_headline(line):
    for k in HEADLINE_KEYS:
        assert k in line, k
    assert line["value"] > 0 and line["steps"] == 3 and line["warmup"] == 1 and line["n_gpus"] == 1
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert "workload" in line["config"] and "model" not in line["config"]


def test_headline_survives_sigkill_behind_the_timed_region():
    env = dict(os.environ, BENCH_STUB_HANG_S="120")
    p = subprocess.Popen([sys.executable, "-c", DRIVER] + ARGS, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    try:
        first = p.stdout.readline()                 # arrives as soon as the timed region is over
        time.sleep(5.0)
        os.kill(p.pid, signal.SIGKILL)
        rest = p.stdout.read()
    finally:
        p.kill()
        p.wait()
    assert p.returncode == -signal.SIGKILL
    lines = [ln for ln in (first + rest).decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    check_headline(json.loads(lines[-1]))


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
def test_later_lines_are_supersets_and_the_last_one_is_complete():
    r = subprocess.run([sys.executable, "-c", DRIVER] + ARGS + ["--no-legs"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 2
    check_headline(lines[0])
    check_headline(lines[1])
    for k in HEADLINE_KEYS:
        if k != "config":
            assert lines[0][k] == lines[1][k]
    last = lines[-1]
    cb = last["cpu_baseline"]
    assert cb["kind"] == "reference" and cb["value"] > 0 and cb["cores"] >= 1 and "sample" in cb
    assert last["config"]["parity_full_sha256_equal"] is True          # the stub's bytes are the oracle's = the reference's
    assert last["config"]["spot_check_first_shards_bit_exact"] is True
    plans = last["config"]["plans"]
    assert [p["shard_KiB"] for p in plans] == [128, 1024] and plans[0]["headline"]
    for p in plans:
        assert p["sha256_equal_reference"] is True and p["x_reference_same_plan"] > 0


def test_a_leg_that_dies_leaves_an_error_not_an_exception():
    import bench
    res = bench.run_leg("stock", "/nonexistent/input.bin", 5, 22, timeout_s=60)
    assert res is None or "error" in res


def test_other_configs_leg_failures_are_reported_not_raised(monkeypatch, tmp_path):
    """`other_configs` of the last line: every configuration is a child run of bench.py with a timeout; one that cannot
    run (no GPU here) comes back as an item with `error`, the others still run."""
    import bench
    monkeypatch.setattr(bench, "OTHER_CONFIGS", [("configs[2] stand-in", ["--quality", "1", "--data", "random", "--steps", "1", "--warmup", "0"], False, 120)])
    f = tmp_path / "in.bin"
    f.write_bytes(b"x" * (1 << 20))
    res = bench.other_configs(str(f), 1, {"env": {}, "threads": 1, "cpus": [0]})
    assert len(res) == 1 and res[0]["config"] == "configs[2] stand-in"
    assert "error" in res[0] or res[0].get("value", 0) > 0
