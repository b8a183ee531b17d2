"""Test-only stand-in for bench.GpuJob (tests/test_bench_line.py): the "device" is the oracle, so that bench.py's
control flow — what it prints, and when — can be checked on a box without a GPU.  Never part of a measured run."""
import hashlib
import os
import time

from refharness import Oracle


class StubJob:
    def __init__(self, args, rank, local_rank, world):
        self.rank, self.world, self.o = rank, world, Oracle()

    def load(self, data, quality, lgwin, shard, size_hint):
        self.data, self.q, self.lgwin, self.shard, self.hint = data, quality, lgwin, shard, size_hint
        self.out = self._plan(shard)

    def _plan(self, shard):
        n = len(self.data)
        return b"".join(self.o.encode_shard(self.data[i:i + shard], self.q, self.lgwin, self.hint, min(i, 1 << 30), i + shard >= n)
                        for i in range(0, n, shard))

    def barrier(self):
        pass

    def step(self):
        time.sleep(0.01)
        info = {"nshards": -(-len(self.data) // self.shard), "ms_total": 10.0, "ms_init": 0.1, "ms_index": 5.0,
                "ms_ix_bucket": 4.0, "ms_parse": 3.0, "ms_build": 1.0, "ms_store": 0.8, "ms_gather": 0.1}
        return len(self.out), info

    def reduce(self, dt, nbytes):
        return dt, nbytes, None

    def output_bytes(self, nbytes):
        hang = float(os.environ.get("BENCH_STUB_HANG_S", "0"))
        if hang:
            time.sleep(hang)          # "everything behind the timed region hangs"
        return self.out[:nbytes]

    def time_plan(self, quality, lgwin, kb, size_hint, reps=3):
        out = self._plan(kb << 10)
        return {"shard_KiB": kb, "shards": -(-len(self.data) // (kb << 10)), "MBps": 100.0, "ms_per_step": 1.0,
                "ratio": round(len(self.data) / len(out), 4), "compressed_bytes": len(out),
                "sha256": hashlib.sha256(out).hexdigest(), "headline": False, "steps": reps}

    def close(self):
        pass

    def finish(self):
        pass
