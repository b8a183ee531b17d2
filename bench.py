#!/usr/bin/env python3
"""bench.py — encode MB/s at quality 5, lgwin 22 (BASELINE.json metric).

One "step" = one pass of the encoder hot path (table init, LZ77 parse, meta-block
modelling, prefix codes + bit emission, shard concatenation) over the whole
synthetic input, which is resident in HBM when the timed region starts.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--size-mb M] [--shard-kb S]

N = 1: BASELINE configs[1] (1 GiB synthetic enwik-style text, q5, lgwin 22, one
MI355X).  N > 1 (launched by torch.distributed.run): weak scaling, every rank
encodes its own M MiB piece of one stream (BROTLI_PARAM_STREAM_OFFSET shards,
encode.h:231-246) and the ranks concatenate the compressed pieces with one
RCCL all-gather of sizes + one of (padded) payloads inside the timed region.

Rank 0 prints the driver contract's JSON line — up to three times, each a superset of the one before (the timed
region alone; + cpu_baseline / plans[] / parity; + the legs behind the headline, each run in a child process with a
timeout): the LAST line is the complete one, and a run that ends early for any reason still leaves a parseable
headline on stdout.  Two extra objects:
  roofline     — the dominant kernel (quality 5: k_ix_bucket with k_ix_big behind it —
                 the sort inside the index buckets + the window search of every
                 position; the second kernel takes the blocks of the buckets too
                 big for one wave, and the HIP events bracket the pair): its
                 algorithmic HBM bytes per launch (17 B per input
                 byte) / its HIP-event time, measured live on the library's
                 stream; `parse_path` beside it prices the whole LZ77 parse
                 (index kernels + chain) at SURVEY.md §8(d)'s 48 B per input byte;
  cpu_baseline — the reference encoder (oracle/_ref, built from /root/reference
                 by oracle/Makefile) with the SAME partition plan on the host
                 cores of this box, on a bounded sample (N = 1 only).
The oracle / reference are only used as the baseline and for a spot check of
the bytes; the measured path is the HIP library behind the C ABI.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# SURVEY.md §8(d) models: q5 / H68 48 B per input byte; q9 / H6 0.47 KiB.
# (qualities 2 - 4: one 4-byte slot written per stored position, 1 << sweep_bits slots and as many 32-byte
#  candidate reads per searched position, about every third position searched)
ALGO_BYTES_PER_INPUT_BYTE = {2: 20.0, 3: 32.0, 4: 56.0, 5: 48.0, 6: 48.0, 7: 481.0, 8: 481.0, 9: 481.0}
# k_ix_bucket per position (= per input byte): entry 4 B in, the input byte itself 1 B in (the 16-byte
# gathers hit the L2), srt 4 B + res 8 B out
IX_BUCKET_BYTES_PER_INPUT_BYTE = 17.0
HBM_ACHIEVABLE_GBS = 6300.0        # what a copy kernel reaches
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def cgroup_cpu():
    """The CPU bandwidth limit of this process's cgroup (v2 cpu.max, v1 cpu.cfs_quota_us / cfs_period_us) as a number of
    CPUs (None: unlimited or unreadable), and the throttling counters (periods in which the group was stopped)."""
    quota, throttled, periods = None, None, None
    try:
        rel = "/"
        for ln in open("/proc/self/cgroup"):
            parts = ln.strip().split(":", 2)
            if len(parts) == 3 and (parts[1] == "" or "cpu" in parts[1].split(",")):
                rel = parts[2]
                if parts[1] != "":
                    break
        for base in ("/sys/fs/cgroup" + rel, "/sys/fs/cgroup/cpu" + rel, "/sys/fs/cgroup", "/sys/fs/cgroup/cpu"):
            if os.path.exists(base + "/cpu.max"):
                q, per = open(base + "/cpu.max").read().split()[:2]
                quota = None if q == "max" else float(q) / float(per)
            elif os.path.exists(base + "/cpu.cfs_quota_us"):
                q = float(open(base + "/cpu.cfs_quota_us").read())
                per = float(open(base + "/cpu.cfs_period_us").read())
                quota = None if q <= 0 else q / per
            else:
                continue
            for ln in open(base + "/cpu.stat"):
                k, v = ln.split()[:2]
                if k == "nr_throttled":
                    throttled = int(v)
                if k == "nr_periods":
                    periods = int(v)
            break
    except (OSError, ValueError):
        pass
    return {"quota_cpus": quota, "nr_throttled": throttled, "nr_periods": periods}


def host_cpu_info():
    """CPU model, socket count and one logical CPU per physical core of socket 0 (BASELINE.md
    section 3.3: the baseline runs pinned to the physical cores of ONE socket); info["smt_siblings"] = the other
    hardware threads of those cores (same order), [] where the box shows none."""
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    allowed = sorted(os.sched_getaffinity(0))
    pk, seen, first_socket = {}, set(), []
    for c in allowed:
        base = "/sys/devices/system/cpu/cpu%d/topology/" % c
        try:
            p = int(open(base + "physical_package_id").read())
            core = int(open(base + "core_id").read())
        except (OSError, ValueError):
            p, core = 0, c
        pk.setdefault(p, 0)
        pk[p] += 1
        if (p, core) not in seen:
            seen.add((p, core))
            if p == min(pk):
                first_socket.append(c)
    s0 = min(pk) if pk else 0
    siblings = []
    for c in first_socket:
        try:
            txt = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip()
        except OSError:
            txt = ""
        sib = []
        for part in txt.split(","):
            if "-" in part:
                a, b = part.split("-")
                sib += list(range(int(a), int(b) + 1))
            elif part:
                sib.append(int(part))
        siblings += [x for x in sib if x != c and x in allowed]
    numa = None
    try:
        numa = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
    except OSError:
        pass
    return {"model": model, "sockets": max(1, len(pk)), "logical_cpus": len(allowed),
            "socket0_physical_cores": len(first_socket), "socket": s0, "smt_siblings": siblings,
            "numa_nodes": numa, "cgroup_cpu_quota": cgroup_cpu()["quota_cpus"],
            "loadavg": (open("/proc/loadavg").read().split()[:3] if os.path.exists("/proc/loadavg") else None)}, first_socket


def data_file(data):
    """The input as a file in memory-backed storage: what the CPU baseline's driver and the legs' child processes read."""
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    path = os.path.join(tmpdir, "brotli_amd_bench_%d.bin" % os.getpid())
    with open(path, "wb") as f:
        f.write(data)
    return path


def cpu_scaling(run, path, nbytes, shard_size, cpus, smt, sample_mb=256):
    """How the reference scales over this box's cores at the headline plan, and what it takes to make it fast: the
    first `sample_mb` MiB of the input, oracle/plan_bench.c driven (a) at all physical cores of socket 0 with each
    allocator / worker variant (glibc malloc — mmap + munmap of ~2.7 MiB per instance —, glibc arenas, a per-worker
    pool behind the reference's alloc hooks, forked processes with and without the pool), (b) with the best of those at
    1, 8, 16, 32 ... threads up to the SMT siblings.  Returns the table and the fastest configuration."""
    cores = len(cpus)
    n = min(nbytes, sample_mb << 20)
    sample = path + ".scal"
    with open(path, "rb") as f, open(sample, "wb") as g:
        g.write(f.read(n))
    try:
        variants = [("malloc, threads", {}),
                    ("glibc arenas (no mmap per block), threads", {"PLAN_BENCH_ALLOC": "arena"}),
                    ("per-worker pool behind alloc_func, threads", {"PLAN_BENCH_ALLOC": "pool"}),
                    ("malloc, forked processes", {"PLAN_BENCH_PROCS": "1"}),
                    ("per-worker pool, forked processes", {"PLAN_BENCH_ALLOC": "pool", "PLAN_BENCH_PROCS": "1"})]
        at_full = []

        def throttled():
            return cgroup_cpu()["nr_throttled"]
        for label, env in variants:
            th0 = throttled()
            r = run(sample, cores, shard_size, 3, cpus, env)
            at_full.append({"variant": label, "env": env, "threads": cores, "MBps": round(r["MBps"], 1),
                            "seconds_all": r["seconds_all"],
                            "cgroup_throttled_periods": None if th0 is None else throttled() - th0})
        best_v = max(at_full, key=lambda v: v["MBps"])
        quota = cgroup_cpu()["quota_cpus"]
        counts = sorted(set([1] + [t for t in (8, 16, 32, 64, 128) if t < cores] + [cores] +
                            ([int(quota)] if quota and 1 <= int(quota) < cores else [])))
        curve = []
        for t in counts:
            th0 = throttled()
            r = run(sample, t, shard_size, 1 if t == 1 else 3, cpus[:t], best_v["env"])
            curve.append({"threads": t, "cpus": "one per physical core", "MBps": round(r["MBps"], 1), "seconds_all": r["seconds_all"],
                          "cgroup_throttled_periods": None if th0 is None else throttled() - th0})
        if smt:
            both = list(cpus) + list(smt)
            r = run(sample, len(both), shard_size, 3, both, best_v["env"])
            curve.append({"threads": len(both), "cpus": "physical cores + their SMT siblings", "MBps": round(r["MBps"], 1)})
        one = curve[0]["MBps"]
        for c in curve:
            c["x_one_thread"] = round(c["MBps"] / one, 2)
            c["parallel_efficiency"] = round(c["MBps"] / one / min(c["threads"], cores), 3)
        top = max(curve, key=lambda c: c["MBps"])
        # (of the points within 3 % of the fastest, the one with the fewest threads: past a CPU quota the curve is noise)
        top = min([c for c in curve if c["MBps"] >= 0.97 * top["MBps"]], key=lambda c: c["threads"])
        use_smt = top["cpus"].startswith("physical cores +")
        best = {"env": best_v["env"], "variant": best_v["variant"], "threads": top["threads"],
                "cpus": (list(cpus) + list(smt)) if use_smt else list(cpus[:top["threads"]]), "MBps_on_sample": top["MBps"]}
        # the largest thread count that ran unthrottled (inside the quota): what the extrapolation to a socket rests on
        free = max([c for c in curve if c["cpus"] == "one per physical core" and (not quota or c["threads"] <= quota)] or [curve[0]],
                   key=lambda c: c["threads"])
        if quota and top["threads"] <= 2 * quota and quota < cores:
            bound = ("the CPU bandwidth limit of this box's cgroup: cpu.max allows %.0f CPUs' worth of run time per period, so the "
                     "curve is ~linear up to %d threads and flat or falling beyond (the kernel stops the group for the rest of "
                     "each period: cgroup_throttled_periods).  It is not the reference, the allocator (variants above) or the "
                     "partition plan that stops scaling here; a whole %d-core socket could not be measured on this box"
                     % (quota, int(quota), cores))
        else:
            bound = "no cgroup CPU limit in the way: see the curve for where the reference itself stops scaling"
        return {"sample": "first %d MiB of the input, the headline plan (%d KiB shards), oracle/plan_bench.c; the main thread "
                          "pins itself to the CPU list before it reads the input (first touch on the workers' socket)" % (n >> 20, shard_size >> 10),
                "cgroup_cpu_quota": quota, "bounded_by": bound,
                "one_thread_MBps": one,
                "whole_socket_if_scaling_held_MBps": round(one * cores * free["parallel_efficiency"], 1),
                "whole_socket_note": "one thread's rate x %d physical cores x the parallel efficiency measured at %d threads, the largest "
                                     "count the cgroup did not throttle — an extrapolation, NOT a measurement (printed so that the "
                                     "multiple against this box's throttled baseline is not mistaken for one against a full socket)"
                                     % (cores, free["threads"]),
                "variants_at_all_physical_cores": [{k: v for k, v in a.items() if k != "env"} for a in at_full],
                "curve_with_best_variant": curve, "best": best,
                "malloc_threads_MBps": at_full[0]["MBps"],
                "gain_over_round5_setting": round(top["MBps"] / at_full[0]["MBps"], 3)}
    finally:
        try:
            os.unlink(sample)
        except OSError:
            pass


def cpu_baseline(path, nbytes, quality, lgwin, shard_size, size_hint, reps=5, other_plans=(), scaling_study=True,
                 bounded=False):
    """The reference encoder (oracle/_ref/libbrotli_ref.so, built from /root/reference by
    oracle/Makefile) driven by oracle/_ref/plan_bench (C, one pinned POSIX thread per physical
    core of socket 0, one encoder instance per shard) over the input file `path`: (B2) the SAME partition plan as the GPU
    run, median of `reps`, with the sha256 of the concatenated output; (B2') the reference at a
    plan that suits the CPU (8 MiB shards); (B1) one instance on one core (what c/enc does by
    itself).  Falls back to the Python thread pool over the oracle when the prebuilt reference
    is absent."""
    info, cpus = host_cpu_info()
    cores = max(1, len(cpus))
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libbrotli_ref.so")
    drv = os.path.join(ROOT, "oracle", "_ref", "plan_bench")
    if os.path.exists(ref_so) and os.path.exists(drv):
        def run(src, threads, shard, nreps, pin, mode=None):
            cmd = [drv, ref_so, src, str(quality), str(lgwin), str(shard), str(threads),
                   str(size_hint), str(nreps)]
            if pin:
                cmd.append(",".join(str(c) for c in pin))
            env = dict(os.environ)
            for k, v in (mode or {}).items():
                env[k] = v
            r = subprocess.run(cmd, capture_output=True, text=True, check=True, env=env)
            return json.loads(r.stdout.strip().splitlines()[-1])
        # The denominator is the FASTEST way found to drive the reference on this box: allocator,
        # threads or processes, thread count up to the SMT siblings — measured on a bounded sample first
        scaling = None
        mode, threads, pin = None, cores, cpus
        if os.environ.get("BENCH_CPU_MODE"):
            # (a leg of `other_configs`: the configuration the headline's scaling study found fastest)
            try:
                m = json.loads(os.environ["BENCH_CPU_MODE"])
                mode, threads, pin = m.get("env"), int(m["threads"]), list(m["cpus"])
            except (ValueError, KeyError):
                pass
        if bounded:
            reps, scaling_study, other_plans = 1, False, ()
        if scaling_study and nbytes >= (64 << 20):
            try:
                scaling = cpu_scaling(run, path, nbytes, shard_size, cpus, info.get("smt_siblings") or [])
                mode, threads, pin = scaling["best"]["env"], scaling["best"]["threads"], scaling["best"]["cpus"]
            except Exception as e:
                scaling = {"error": repr(e)[:300]}
        if not bounded:
            run(path, threads, shard_size, 1, pin, mode)                  # warm-up, discarded
        same = run(path, threads, shard_size, reps, pin, mode)
        big = 8 << 20
        best = run(path, threads, big, 3 if reps > 1 else 1, pin, mode) if nbytes >= 4 * big and reps > 1 else None
        others = {sh: run(path, threads, sh, 3, pin, mode) for sh in other_plans}
        one_n = min(nbytes, (16 << 20) if bounded else (64 << 20))
        one_path = path + ".one"
        try:
            with open(path, "rb") as f, open(one_path, "wb") as g:
                g.write(f.read(one_n))
            single = run(one_path, 1, 0, 1, cpus[:1])                      # one instance, one core
        finally:
            try:
                os.unlink(one_path)
            except OSError:
                pass
        out = {
            "value": round(same["MBps"], 1), "unit": "MB/s", "cores": threads, "kind": "reference",
            "threads": threads, "physical_cores_of_the_socket": cores, "workers": same.get("workers"), "alloc": same.get("alloc"),
            "sample": "the whole %d MiB input, same plan (%d shards of %d KiB), %d %s (%s) pinned to the "
                      "physical cores%s of socket %d (oracle/plan_bench.c) — the fastest configuration of `scaling` —, "
                      "median of %d run(s)%s, %.3f s, ratio %.3f" % (
                          nbytes >> 20, same["shards"], shard_size >> 10, threads, same.get("workers", "threads"),
                          same.get("alloc", "malloc"), " and their SMT siblings" if threads > cores else "", info["socket"], reps,
                          "" if bounded else " after one warm-up", same["seconds"], same["bytes"] / max(1, same["out_bytes"])),
            "cpu": info, "seconds_all": same["seconds_all"], "sha256": same["sha256"],
            "out_bytes": same["out_bytes"],
            "single_stream_1core_MBps": round(single["MBps"], 1),
            "single_stream_ratio": round(single["bytes"] / max(1, single["out_bytes"]), 4),
            "single_stream_sample": "first %d MiB, one encoder instance, 1 thread" % (one_n >> 20),
        }
        if scaling is not None:
            out["scaling"] = scaling
            if "curve_with_best_variant" in scaling:
                one = scaling["curve_with_best_variant"][0]["MBps"]
                out["parallel_efficiency"] = round(same["MBps"] / (one * cores), 3)
                out["parallel_efficiency_note"] = ("whole-input rate / (one thread's rate on the sample x %d physical cores); "
                                                   "see scaling.curve_with_best_variant for where it flattens" % cores)
        if others:
            # the reference driven with the other partition plans of config.plans[] (same cores, median of 3)
            out["other_plans"] = {str(sh >> 10): {"MBps": round(r["MBps"], 1), "sha256": r["sha256"], "out_bytes": r["out_bytes"],
                                                  "shards": r["shards"], "seconds_all": r["seconds_all"]}
                                  for sh, r in others.items()}
        if best:
            out["reference_own_plan"] = {
                "MBps": round(best["MBps"], 1), "shard_KiB": big >> 10, "shards": best["shards"],
                "ratio": round(best["bytes"] / max(1, best["out_bytes"]), 4),
                "note": "the reference at a plan that suits the CPU (tables cleared once per 8 MiB), same cores"}
        return out
    from concurrent.futures import ThreadPoolExecutor
    from refharness import Oracle
    enc = Oracle()
    data = open(path, "rb").read()
    nsh = min(-(-len(data) // shard_size), max(cores, 64))
    sample = data[:nsh * shard_size]

    def one_shard(k):
        off = k * shard_size
        piece = sample[off:off + shard_size]
        return len(enc.encode_shard(piece, quality, lgwin, size_hint, min(off, 1 << 30),
                                    off + len(piece) == len(data)))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        out_bytes = sum(ex.map(one_shard, range(nsh)))
    dt = time.perf_counter() - t0
    return {"value": round(len(sample) / 1e6 / dt, 1), "unit": "MB/s", "cores": cores, "kind": "port",
            "sample": "first %d MiB, same plan, %d Python threads over oracle/liboracle.so, %.1f s; "
                      "ratio %.3f" % (len(sample) >> 20, cores, dt, len(sample) / max(1, out_bytes))}


# ---- the legs behind the headline: each one a child process with a timeout ------------------------------------
# Whatever happens in one of them — a GPU fault, a lost context, a timeout — the headline line is on stdout already
# and the parent prints the line again, enriched with what did come back (round 4's driver run
# ended without a line because the only print came after all of these).
LEG_TIMEOUT_S = {"round_trip": 150, "abi": 150, "stock": 90, "stock_whole": 200, "process_fed": 150, "concurrent": 90}


def _bind_encoder(lib_path):
    L = C.CDLL(lib_path)
    L.BrotliEncoderMaxCompressedSize.restype = C.c_size_t
    L.BrotliEncoderMaxCompressedSize.argtypes = [C.c_size_t]
    L.BrotliEncoderCompress.restype = C.c_int
    L.BrotliEncoderCompress.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_char_p,
                                        C.POINTER(C.c_size_t), C.c_char_p]
    return L


DROPIN = os.path.join(ROOT, "brotli_amd", "lib", "libbrotlienc_amd.so")


def leg_abi(path, quality, lgwin, shard_kb, want_bytes, want_sha):
    """host buffer -> BrotliEncoderCompress of the drop-in library
    (libbrotlienc_amd.so, partition plan from BROTLI_AMD_SHARD_KB) -> host buffer, PCIe included.
    Never the reported `value`."""
    import hashlib
    if not os.path.exists(DROPIN):
        return {"error": "libbrotlienc_amd.so missing"}
    data = open(path, "rb").read()
    os.environ["BROTLI_AMD_SHARD_KB"] = str(shard_kb)
    L = _bind_encoder(DROPIN)
    cap = L.BrotliEncoderMaxCompressedSize(len(data))
    out = C.create_string_buffer(cap)
    times, n_out = [], 0
    for _ in range(3):
        sz = C.c_size_t(cap)
        t0 = time.perf_counter()
        ok = L.BrotliEncoderCompress(quality, lgwin, 0, len(data), data, C.byref(sz), out)
        times.append(time.perf_counter() - t0)
        if not ok:
            return {"error": "BrotliEncoderCompress returned BROTLI_FALSE"}
        n_out = sz.value
    dt = sorted(times)[len(times) // 2]
    return {"MBps": round(len(data) / 1e6 / dt, 1), "seconds_all": [round(t, 4) for t in times],
            "out_bytes": n_out, "bytes_equal_device_path": n_out == want_bytes and
            hashlib.sha256(out.raw[:n_out]).hexdigest() == want_sha,
            "note": "one BrotliEncoderCompress call on a pageable host buffer, BROTLI_AMD_SHARD_KB=%d, "
                    "median of 3 (the first includes context creation)" % shard_kb}


def leg_stock(path, quality, lgwin):
    """What a caller that knows nothing of this library gets: BrotliEncoderCompress(quality, lgwin, ...) of the drop-in
    library with NO partition plan (BROTLI_AMD_SHARD_KB unset), on the first (1 << lgwin) - 16 bytes of the input — the
    longest input a one-shot quality-5 call runs on the index + tiled chain of ONE shard."""
    if not os.path.exists(DROPIN) or quality != 5 or lgwin < 17:
        return None
    os.environ.pop("BROTLI_AMD_SHARD_KB", None)
    L = _bind_encoder(DROPIN)
    n = (1 << min(lgwin, 22)) - 16
    with open(path, "rb") as f:
        piece = f.read(n)
    n = len(piece)
    cap = L.BrotliEncoderMaxCompressedSize(n)
    out = C.create_string_buffer(cap)
    times = []
    for _ in range(4):
        sz = C.c_size_t(cap)
        t0 = time.perf_counter()
        ok = L.BrotliEncoderCompress(quality, lgwin, 0, n, piece, C.byref(sz), out)
        times.append(time.perf_counter() - t0)
        if not ok:
            return {"error": "BrotliEncoderCompress returned BROTLI_FALSE"}
    dt = sorted(times[1:])[1]
    res = {"bytes": n, "MBps": round(n / 1e6 / dt, 1), "seconds_all": [round(t, 4) for t in times], "out_bytes": sz.value,
           "note": "one BrotliEncoderCompress call, no partition plan, pageable host buffers; median of calls 2-4"}
    print(json.dumps(res), flush=True)
    try:
        from refharness import Ref, have_ref
        if have_ref():
            t0 = time.perf_counter()
            want = Ref().compress(piece, quality, lgwin)
            res["reference_1core_MBps"] = round(n / 1e6 / (time.perf_counter() - t0), 1)
            res["bytes_equal_reference"] = want == out.raw[:sz.value]
    except Exception as e:
        res["reference"] = repr(e)[:200]
    return res


def leg_stock_whole(path, quality, lgwin):
    """The same stock call on the WHOLE input (1 GiB by default): longer than the window, it takes the tiled stream
    path of the library (k_tile.h, JOB_FLAG_STREAMT) — or the serial device stream where the data does not suit it."""
    import hashlib
    data = open(path, "rb").read()
    n = len(data)
    if n <= (1 << lgwin) or n >= (1 << 31):
        return None
    os.environ.pop("BROTLI_AMD_SHARD_KB", None)
    L = _bind_encoder(DROPIN)
    cap = L.BrotliEncoderMaxCompressedSize(n)
    out = C.create_string_buffer(cap)
    times = []
    for _ in range(3):
        sz = C.c_size_t(cap)
        t0 = time.perf_counter()
        ok = L.BrotliEncoderCompress(quality, lgwin, 0, n, data, C.byref(sz), out)
        times.append(time.perf_counter() - t0)
        if not ok:
            return {"error": "BrotliEncoderCompress returned BROTLI_FALSE"}
    sha = hashlib.sha256(out.raw[:sz.value]).hexdigest()
    res = {"bytes": n, "out_bytes": sz.value, "seconds_all": [round(t, 3) for t in times],
           "MBps": round(n / 1e6 / min(times[1:]), 1), "sha256": sha,
           "note": "one BrotliEncoderCompress call on the whole input, no partition plan, pageable host buffers, PCIe both "
                   "ways included; the best of calls 2-3 (the first one creates the context and its workspace)"}
    print(json.dumps(res), flush=True)          # (kept even if the reference leg below does not finish)
    try:
        from refharness import Ref, have_ref
        if have_ref():
            t0 = time.perf_counter()
            want = Ref().compress(data, quality, lgwin)
            res["reference_1core_MBps"] = round(n / 1e6 / (time.perf_counter() - t0), 1)
            res["bytes_equal_reference"] = hashlib.sha256(want).hexdigest() == sha and len(want) == sz.value
    except Exception as e:
        res["reference"] = repr(e)[:200]
    return res


def leg_round_trip(path, quality, lgwin, shard_kb, size_hint):
    """Round trip on the device: the shards of the plan decode as independent pieces (k_decode.h, one wave per shard)
    and must give back the input."""
    import torch
    from brotli_amd import hip
    data = open(path, "rb").read()
    n, shard = len(data), shard_kb << 10
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    d_in = hip.to_device(data, 0)
    ctx = hip.Context(0)
    params = hip.make_params(quality, lgwin, shard, size_hint, stream_base=0, is_last=True)
    d_out = torch.empty(ctx.max_output(n, params), dtype=torch.uint8, device=dev)
    nsh = -(-n // shard)
    d_sizes = torch.zeros(nsh, dtype=torch.int64, device=dev)
    nb2, _ = ctx.encode_device(d_in, n, params, d_out, d_sizes)
    sizes = d_sizes.cpu().tolist()
    if nb2 + hip.DECODE_SLACK > d_out.numel() or sum(sizes) != nb2:
        return {"error": "shard sizes do not add up"}
    d_back = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    pieces = hip.plan_pieces(sizes, n, shard, lgwin)
    res, dec_ms = ctx.decode_device(d_out, nb2, d_back, n, pieces, check=False)
    errs = [r[1] for r in res if r[1]]
    same = bool(torch.equal(d_back[:n], d_in[:n]))
    return {"equal_to_input": same and not errs and res[-1][2] == 1, "pieces": len(pieces),
            "piece_errors": len(errs), "decode_ms": round(dec_ms, 3),
            "decode_GBps_of_output": round(n / 1e9 / (dec_ms / 1e3), 2) if dec_ms > 0 else None}


def leg_process_fed(path, quality, lgwin, piece_kb, max_mb):
    """A stream handed over in PROCESS calls of piece_kb KiB with no BROTLI_PARAM_SIZE_HINT, then FINISH (what the Python
    module's Compressor.process or the CLI on a pipe does): the first max_mb MiB of the input through
    BrotliEncoderCompressStream of the drop-in library, next to the reference library driven the same way."""
    import hashlib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_abi import _bind, drive
    with open(path, "rb") as f:
        data = f.read(max_mb << 20)
    os.environ.pop("BROTLI_AMD_SHARD_KB", None)
    piece = piece_kb << 10
    ops = [(piece, 0)] * (len(data) // piece) + [(len(data) % piece, 2)]
    params = ((1, quality), (2, lgwin))
    L = _bind(DROPIN)
    times = []
    got = b""
    # (the calls themselves are timed: instance creation, every BrotliEncoderCompressStream call with the output
    #  taken as it comes, destruction — not the harness's copies of the input and the output, which tests' drive() makes)
    buf = C.create_string_buffer(data, len(data))
    cap = len(data) + (len(data) >> 3) + 4096
    out = C.create_string_buffer(cap)
    for _ in range(2):
        t0 = time.perf_counter()
        st = L.BrotliEncoderCreateInstance(None, None, None)
        for k, v in params:
            L.BrotliEncoderSetParameter(st, k, v)
        off = pos = 0
        for nb, op in ops:
            avail_in = C.c_size_t(nb)
            next_in = C.c_void_p(C.addressof(buf) + off)
            off += nb
            while True:
                avail_out = C.c_size_t(cap - pos)
                next_out = C.c_void_p(C.addressof(out) + pos)
                if not L.BrotliEncoderCompressStream(st, op, C.byref(avail_in), C.byref(next_in), C.byref(avail_out), C.byref(next_out), None):
                    return {"error": "BrotliEncoderCompressStream returned BROTLI_FALSE"}
                pos = cap - avail_out.value
                if avail_in.value == 0 and not L.BrotliEncoderHasMoreOutput(st):
                    break
        fin = bool(L.BrotliEncoderIsFinished(st))
        L.BrotliEncoderDestroyInstance(st)
        times.append(time.perf_counter() - t0)
        got = out.raw[:pos]
        if not fin:
            return {"error": "stream not finished"}
    res = {"bytes": len(data), "piece_KiB": piece_kb, "MBps": round(len(data) / 1e6 / times[-1], 1), "seconds_all": [round(t, 3) for t in times],
           "out_bytes": len(got), "sha256": hashlib.sha256(bytes(got)).hexdigest(),
           "note": "PROCESS calls of %d KiB without a size hint, then FINISH; pageable host buffers, the second of two runs" % piece_kb}
    print(json.dumps(res), flush=True)
    try:
        from refharness import have_ref
        if have_ref():
            R = _bind(os.path.join(ROOT, "oracle", "_ref", "libbrotli_ref.so"))
            t0 = time.perf_counter()
            want, _ = drive(R, data, ops, params=params, out_chunk=1 << 24)
            res["reference_1core_MBps"] = round(len(data) / 1e6 / (time.perf_counter() - t0), 1)
            res["bytes_equal_reference"] = bytes(want) == bytes(got)
    except Exception as e:
        res["reference"] = repr(e)[:200]
    return res


def leg_concurrent(path, quality, lgwin, mib, threads, calls):
    """T caller threads, each making stock one-shot calls of `mib` MiB on an encoder instance of its own (the library
    lends each a device context with its own HIP stream): what a server compressing many files at once does.  One call of
    a few MiB is a string of small latency-bound kernels; the device is filled by having many in flight (the library
    asks the HIP runtime for 32 hardware queues, hip_layer.hip env_read).  Aggregate rate, sha256 of every thread's last
    output against the reference library's."""
    import hashlib
    import threading
    n = (mib << 20) - 16
    with open(path, "rb") as f:
        datas = [f.read(mib << 20)[:n] for _ in range(min(threads, 8))]
    os.environ.pop("BROTLI_AMD_SHARD_KB", None)
    L = _bind_encoder(DROPIN)
    cap = L.BrotliEncoderMaxCompressedSize(n)
    want = None
    try:
        from refharness import Ref, have_ref
        if have_ref():
            r = Ref()
            want = [hashlib.sha256(r.compress(d, quality, lgwin)).hexdigest() for d in datas]
    except Exception:
        want = None
    outs = [C.create_string_buffer(cap) for _ in range(threads)]
    bad = []

    def work(k, reps):
        for rep in range(reps):
            sz = C.c_size_t(cap)
            ok = L.BrotliEncoderCompress(quality, lgwin, 0, n, datas[k % len(datas)], C.byref(sz), outs[k])
            if not ok or (want is not None and rep == reps - 1 and
                          hashlib.sha256(outs[k].raw[:sz.value]).hexdigest() != want[k % len(datas)]):
                bad.append(k)

    def wave(nthreads, reps):
        ws = [threading.Thread(target=work, args=(k, reps)) for k in range(nthreads)]
        t0 = time.perf_counter()
        [t.start() for t in ws]
        [t.join() for t in ws]
        return time.perf_counter() - t0

    wave(threads, 1)                           # every thread's context and workspace exist
    one = wave(1, calls)
    dt = wave(threads, calls)
    return {"threads": threads, "MiB_per_call": mib, "calls": threads * calls, "seconds": round(dt, 4),
            "aggregate_MBps": round(threads * calls * n / 1e6 / dt, 1), "one_thread_MBps": round(calls * n / 1e6 / one, 1),
            "ms_per_call_in_flight": round(dt / calls * 1e3, 2), "bytes_equal_reference": (not bad) if want is not None else None,
            "GPU_MAX_HW_QUEUES": (lambda g: (g(b"GPU_MAX_HW_QUEUES") or b"").decode() or None)(
                (lambda lc: (setattr(lc.getenv, "restype", C.c_char_p), lc.getenv)[1])(C.CDLL(None))),
            "note": "stock BrotliEncoderCompress calls from %d threads at once, no partition plan, pageable host buffers" % threads}


LEGS = {"abi": leg_abi, "stock": leg_stock, "stock_whole": leg_stock_whole, "round_trip": leg_round_trip, "process_fed": leg_process_fed,
        "concurrent": leg_concurrent}


def leg_main(argv):
    """bench.py --leg <name> <path> <int args ...> [str]: runs one leg, prints its result as the last JSON line."""
    name, path = argv[0], argv[1]
    rest = [int(a) if a.lstrip("-").isdigit() else a for a in argv[2:]]
    try:
        res = LEGS[name](path, *rest)
    except Exception as e:
        res = {"error": repr(e)[:300]}
    print(json.dumps(res), flush=True)


def run_leg(name, path, *leg_args, timeout_s=None):
    """Runs a leg in a child process; the last JSON line of its stdout, or {"error": ...}."""
    timeout_s = timeout_s or LEG_TIMEOUT_S[name]
    cmd = [sys.executable, os.path.abspath(__file__), "--leg", name, path] + [str(a) for a in leg_args]

    def last_json(text):
        lines = [ln for ln in (text or "").splitlines() if ln.startswith("{") or ln == "null"]
        return json.loads(lines[-1]) if lines else None
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
        got = last_json(r.stdout)
        if r.returncode != 0 or (got is None and "null" not in (r.stdout or "")):
            res = got if isinstance(got, dict) else {}
            res["error"] = "child rc %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:])
            return res
        return got
    except subprocess.TimeoutExpired as e:
        out = e.stdout.decode("utf-8", "replace") if isinstance(e.stdout, bytes) else (e.stdout or "")
        res = last_json(out)
        res = res if isinstance(res, dict) else {}
        res["error"] = "child not through within %d s" % timeout_s
        return res
    except Exception as e:
        return {"error": repr(e)[:300]}


def main_q1(args):
    """BASELINE configs[2]: quality 1 (two-pass fragment compressor) on one GPU.  The stream is
    the reference's own unpartitioned stream: one BrotliEncoderCompress-style FINISH call
    (fragments of 1 << lgwin) or, with --feed-kb, the CLI's call pattern."""
    import torch
    import gen_inputs as G
    from brotli_amd import hip
    if int(os.environ.get("WORLD_SIZE", "1")) != 1:
        sys.exit("quality 1 is a single-GPU configuration (BASELINE configs[2])")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    n = args.size_mb << 20
    if args.data == "random":
        g = torch.Generator(device="cuda").manual_seed(G.SEED)
        d_in = torch.zeros(n + hip.INPUT_SLACK, dtype=torch.uint8, device=dev)
        d_in[:n] = torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev, generator=g)
        data = None
    else:
        data = G.enwik_text(n, seed=G.SEED)
        d_in = hip.to_device(data, 0)
    ctx = hip.Context(0)
    feed = args.feed_kb << 10
    calls = None
    if feed:
        calls = [min(feed, n - o) for o in range(0, n, feed)]
        if n % feed == 0:
            calls.append(0)
    d_out = torch.empty(ctx.fast_max_output(n, len(calls) if calls else 1, args.lgwin), dtype=torch.uint8, device=dev)

    def step():
        return ctx.encode_fast_device(d_in, n, d_out, args.lgwin, calls)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    infos = []
    nbits = 0
    for _ in range(args.steps):
        nbits, info = step()
        infos.append(info)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    nbytes = nbits // 8
    stage = {k: round(sum(i[k] for i in infos) / len(infos), 3) for k in ("ms_total", "ms_parse", "ms_store", "ms_gather")}
    stage["ms_place"] = stage.pop("ms_gather")
    dom = max(("ms_parse", "ms_store", "ms_place"), key=lambda k: stage[k])
    kernel = {"ms_parse": "k_fast_parse", "ms_store": "k_fast_store", "ms_place": "k_fast_sizes+scan+emit"}[dom]
    # algorithmic bytes per launch: parse reads the input once and writes the literal
    # bytes + command words it keeps; store reads those and writes the bit string; place moves
    # the output once (read + write).
    out_b = float(nbytes)
    algo = {"ms_parse": 2.0 * n, "ms_store": 2.0 * n + out_b, "ms_place": 2.0 * out_b}[dom]
    achieved = algo / (stage[dom] / 1e3) / 1e9
    traffic, traffic_source = None, "not measured (no PMC pass on record for this configuration)"
    prof = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(prof):
        t = json.load(open(prof)).get("q1/%d/%s" % (args.size_mb, args.data))
        if t and t.get("kernel") == kernel:
            traffic, traffic_source = t["hbm_bytes_per_launch"], "model input, not this run: " + t["source"]
    line = {
        "metric": "encode MB/s at quality 1, lgwin %d, %d MiB %s input; bit-exact vs c/enc" % (
            args.lgwin, args.size_mb, args.data),
        "value": round(n / 1e6 / (dt / args.steps), 1), "unit": "MB/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {
            "workload": "%d MiB %s, quality 1, lgwin %d, 1 x MI355X" % (
                args.size_mb, "uniform random bytes (torch.randint on the device, seed %d)" % G.SEED
                if args.data == "random" else "synthetic enwik-style text (tests/gen_inputs.enwik_text)", args.lgwin),
            "call_pattern": "one FINISH call (fragments of %d KiB)" % (1 << (args.lgwin - 10)) if not feed else
                            "%d KiB per CompressStream call (c/tools/brotli.c pattern)" % args.feed_kb,
            "fragments": infos[-1]["nshards"], "compressed_bytes": nbytes, "ratio": round(n / max(1, nbytes), 4),
            "stage_ms": stage,
        },
        "roofline": {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                     "note": "dominant stage %s: %.3f ms per launch (HIP events on the library's stream), "
                             "algorithmic bytes %.0f per launch" % (dom, stage[dom], algo)},
    }
    emit(line)          # the headline exists: printed before anything else can go wrong (the last line is the complete one)
    if not args.no_cpu_baseline:
        import hashlib
        from refharness import Ref, have_ref, Oracle
        m = n if n <= (1 << 30) else (256 << 20)       # (the reference does ~0.75 GB/s at quality 1: the whole GiB is 1.4 s)
        sample = d_in[:m].cpu().numpy().tobytes() if data is None else data[:m]
        # spot check: the first 16 MiB against the oracle
        k = min(n, 16 << 20)
        nb2, _ = ctx.encode_fast_device(d_in, k, d_out, args.lgwin)
        line["config"]["spot_check_first_16MiB_bit_exact"] = \
            d_out[:nb2 // 8].cpu().numpy().tobytes() == Oracle().encode_fast(sample[:k], args.lgwin)
        ref_so = os.path.join(ROOT, "oracle", "_ref", "libbrotli_ref.so")
        drv = os.path.join(ROOT, "oracle", "_ref", "plan_bench")
        if have_ref() and not calls and os.path.exists(drv):
            # one reference instance, one FINISH call, timed inside oracle/plan_bench.c (no Python buffers in the timing)
            path = data_file(sample)
            try:
                _, cpus = host_cpu_info()
                cmd = [drv, ref_so, path, "1", str(args.lgwin), "0", "1", str(min(len(sample), 1 << 30)), "1", str(cpus[0] if cpus else 0)]
                rr = json.loads(subprocess.run(cmd, capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
            finally:
                os.unlink(path)
            line["cpu_baseline"] = {"value": round(rr["MBps"], 1), "unit": "MB/s", "cores": 1, "kind": "reference",
                                    "sample": "%s %d MiB, one reference encoder instance, one FINISH call (the reference has no "
                                              "threads; quality 1 needs no plan: its fragments are the stream's own), %.2f s, ratio %.3f" % (
                                                  "the whole" if m == n else "first", m >> 20, rr["seconds"], rr["bytes"] / max(1, rr["out_bytes"])),
                                    "sha256": rr["sha256"], "out_bytes": rr["out_bytes"]}
            if m == n:
                nb3, _ = step()
                got = d_out[:(nb3 + 7) // 8].cpu().numpy().tobytes()
                line["config"]["gpu_output_sha256"] = hashlib.sha256(got).hexdigest()
                line["config"]["parity_full_sha256_equal"] = (line["config"]["gpu_output_sha256"] == rr["sha256"] and len(got) == rr["out_bytes"])
        elif have_ref():
            r = Ref()
            sample = sample[:256 << 20]
            t0 = time.perf_counter()
            out = r.encode_calls(sample, 1, args.lgwin, [(len(sample), 2)])
            dtc = time.perf_counter() - t0
            line["cpu_baseline"] = {"value": round(len(sample) / 1e6 / dtc, 1), "unit": "MB/s", "cores": 1,
                                    "kind": "reference",
                                    "sample": "first %d MiB, one reference encoder instance (the reference has no "
                                              "threads), %.2f s, ratio %.3f" % (len(sample) >> 20, dtc, len(sample) / len(out))}
        emit(line)


ALL_CONFIGS = [
    ("configs[1]: 1 GiB enwik-style text, quality 5, lgwin 22", []),
    ("configs[2]: 1 GiB random bytes, quality 1", ["--quality", "1", "--data", "random"]),
    ("configs[3] workload on one GPU: 1 GiB Silesia-style mix, quality 5, lgwin 22", ["--workload", "silesia"]),
    # (384 KiB shards: 2731 waves of k_parse_deep in flight instead of 2048 — 713 against 836 ms, ratio 3.169 against
    #  3.210, profiles/r04_g3_summary.txt; 256 KiB shards would need 128 GiB of bucket tables)
    ("configs[4]: 1 GiB text, quality 9, lgwin 24", ["--quality", "9", "--lgwin", "24", "--shard-kb", "384"]),
]


def main_all_configs(args):
    """Runs this script once per single-GPU BASELINE configuration (a process each: every one builds its own input and
    context) and prints their JSON lines in order; `cpu_baseline` is part of every line."""
    import subprocess
    rc = 0
    for label, extra in ALL_CONFIGS:
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--size-mb", str(args.size_mb)] + extra
        r = subprocess.run(cmd, capture_output=True, text=True)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            rc = 1
            print(json.dumps({"config_label": label, "error": (r.stderr or r.stdout)[-600:]}), flush=True)
            continue
        line = json.loads(lines[-1])
        line["config_label"] = label
        print(json.dumps(line), flush=True)
    sys.exit(rc)


OTHER_CONFIGS = [
    # (label, extra argv, reuses the parent's text file, timeout s)
    ("configs[2]: random bytes, quality 1, lgwin 22, one call", ["--quality", "1", "--data", "random", "--steps", "10", "--warmup", "2"], False, 120),
    ("configs[4]: text, quality 9, lgwin 24, 512 KiB plan", ["--quality", "9", "--lgwin", "24", "--shard-kb", "512", "--steps", "3", "--warmup", "1"], True, 200),
    ("configs[3] workload on one GPU: Silesia-style mix, quality 5, lgwin 22, 128 KiB plan", ["--workload", "silesia", "--steps", "3", "--warmup", "1"], False, 200),
]


def other_configs(path, size_mb, cpu_mode):
    """BASELINE configs[2], [4] and the workload of [3] on this GPU, each a bounded child run of this script (its own
    process, context and timeout): value, roofline, cpu_baseline (ONE run of the reference with the same plan / call on
    the whole input, in the CPU configuration the headline's scaling study found fastest) and the whole-output sha256
    parity.  What comes back is that run's last JSON line, cut down to the fields named."""
    out = []
    for label, extra, reuse, timeout_s in OTHER_CONFIGS:
        cmd = [sys.executable, os.path.abspath(__file__), "--size-mb", str(size_mb), "--no-legs", "--bounded-baseline",
               "--no-other-configs"] + extra + (["--input-file", path] if reuse else [])
        env = dict(os.environ)
        if cpu_mode:
            env["BENCH_CPU_MODE"] = json.dumps({"env": cpu_mode.get("env"), "threads": cpu_mode.get("threads"), "cpus": cpu_mode.get("cpus")})
        t0 = time.perf_counter()
        text = ""
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
            text, err = r.stdout, (None if r.returncode == 0 else "child rc %d: %s" % (r.returncode, (r.stderr or "")[-300:]))
        except subprocess.TimeoutExpired as e:
            text = e.stdout.decode("utf-8", "replace") if isinstance(e.stdout, bytes) else (e.stdout or "")
            err = "child not through within %d s" % timeout_s
        except Exception as e:
            err = repr(e)[:300]
        lines = [ln for ln in (text or "").splitlines() if ln.startswith("{")]
        item = {"config": label, "wall_s": round(time.perf_counter() - t0, 1)}
        if lines:
            d = json.loads(lines[-1])
            c = d.get("config", {})
            item.update({"metric": d.get("metric"), "value": d.get("value"), "unit": d.get("unit"), "steps": d.get("steps"),
                         "ms_per_step": d.get("ms_per_step"), "dtype": d.get("dtype"), "workload": c.get("workload"),
                         "partition_plan": c.get("partition_plan") or c.get("call_pattern"),
                         "ratio": c.get("ratio"), "stage_ms": c.get("stage_ms"),
                         "roofline": {k: v for k, v in (d.get("roofline") or {}).items() if k != "parse_path"},
                         "parity_full_sha256_equal": c.get("parity_full_sha256_equal"),
                         "gpu_output_sha256": c.get("gpu_output_sha256")})
            cb = d.get("cpu_baseline")
            if cb:
                item["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "threads", "workers", "alloc", "kind", "sample",
                                                                "sha256", "single_stream_1core_MBps") if k in cb}
                if cb.get("value"):
                    item["x_cpu_baseline"] = round(d["value"] / cb["value"], 2)
        if err:
            item["error"] = err
        out.append(item)
    return out


class GpuJob:
    """Everything of the measured path that touches the device: the input in HBM, the library's context, the step
    (brotli_amd_encode_device through brotli_amd.hip, plus the RCCL concatenation at N > 1).  tests/test_bench_line.py
    swaps this class for a stub to check what bench.py prints, and when, without a GPU."""

    def __init__(self, args, rank, local_rank, world):
        import torch
        from brotli_amd import hip
        self.torch, self.hip, self.args, self.rank, self.world = torch, hip, args, rank, world
        self.dist = None
        if world > 1 or os.environ.get("BENCH_FORCE_DIST"):   # the latter: exercise the RCCL path on one GPU
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            self.dist = dist
        self.local_rank = local_rank
        self.dev = torch.device("cuda", local_rank)
        torch.cuda.set_device(self.dev)
        self.gathered, self.stream, self.pad_hint = None, None, 0

    def load(self, data, quality, lgwin, shard, size_hint):
        hip, torch = self.hip, self.torch
        self.n = len(data)
        self.d_in = hip.to_device(data, self.local_rank)
        self.ctx = hip.Context(self.local_rank)
        self.params = hip.make_params(quality, lgwin, shard, size_hint, stream_base=self.rank * self.n,
                                      is_last=(self.rank == self.world - 1))
        self.d_out = torch.empty(self.ctx.max_output(self.n, self.params), dtype=torch.uint8, device=self.dev)

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def step(self):
        got = {}

        def encode_local():
            got["nbytes"], got["info"] = self.ctx.encode_device(self.d_in, self.n, self.params, self.d_out)
            return self.d_out, got["nbytes"]
        if self.dist is not None:
            # C1: one all-gather of the sizes, one of the padded payloads, padding stripped —
            # all of it inside the timed region (brotli_amd/dist.py, shared with the gloo test)
            from brotli_amd.dist import sharded_step
            self.stream, _, self.gathered, self.pad_hint = sharded_step(encode_local, scratch=self.gathered,
                                                                        pad_hint=self.pad_hint)
        else:
            encode_local()
        return got["nbytes"], got["info"]

    def reduce(self, dt, nbytes):
        """(max of dt over the ranks, sum of the output bytes, (equal on all ranks, sha256, bytes) of the gathered stream)"""
        if self.dist is None:
            return dt, nbytes, None
        torch, dist = self.torch, self.dist
        t = torch.tensor([dt], dtype=torch.float64, device=self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        from brotli_amd.dist import same_stream_on_all_ranks
        ok, sha = same_stream_on_all_ranks(self.stream)
        tot = torch.tensor([nbytes], dtype=torch.int64, device=self.dev)
        dist.all_reduce(tot)
        return float(t.item()), int(tot.item()), (ok, sha, int(self.stream.numel()))

    def output_bytes(self, nbytes):
        return self.d_out[:nbytes].cpu().numpy().tobytes()

    def time_plan(self, quality, lgwin, kb, size_hint, reps=3):
        """Another partition plan of the same input, timed the same way as the headline (device-resident input,
        synchronize on both sides), outside the headline's timed region."""
        import hashlib
        hip, torch, n = self.hip, self.torch, self.n
        p2 = hip.make_params(quality, lgwin, kb << 10, size_hint, stream_base=0, is_last=True)
        cap2 = self.ctx.max_output(n, p2)
        d_out2 = self.d_out if cap2 <= self.d_out.numel() else torch.empty(cap2, dtype=torch.uint8, device=self.dev)
        self.ctx.encode_device(self.d_in, n, p2, d_out2)
        torch.cuda.synchronize(self.dev)
        t1 = time.perf_counter()
        inf2 = []
        nb2 = 0
        for _ in range(reps):
            nb2, i2 = self.ctx.encode_device(self.d_in, n, p2, d_out2)
            inf2.append(i2)
        torch.cuda.synchronize(self.dev)
        dt2 = (time.perf_counter() - t1) / reps
        comp2 = d_out2[:nb2].cpu().numpy().tobytes()
        return {"shard_KiB": kb, "shards": inf2[-1]["nshards"], "MBps": round(n / 1e6 / dt2, 1),
                "ms_per_step": round(dt2 * 1e3, 3), "ratio": round(n / nb2, 4), "compressed_bytes": nb2,
                "sha256": hashlib.sha256(comp2).hexdigest(), "headline": False, "steps": reps,
                "stage_ms": {k: round(sum(i.get(k, 0.0) for i in inf2) / len(inf2), 3) for k in
                             ("ms_total", "ms_index", "ms_ix_bucket", "ms_parse", "ms_build", "ms_store")},
                "tile_sweeps": inf2[-1].get("tile_sweeps"),
                "tile_fallback_shards": inf2[-1].get("tile_fallback_shards")}

    def close(self):
        """Gives the device back: the legs behind the headline run in processes of their own (quality 9 at 384 KiB
        shards holds 85 GiB of bucket tables, the stock call wants ~85 B per input byte)."""
        try:
            self.ctx.close()
            self.d_out = self.d_in = self.gathered = self.stream = None
            self.torch.cuda.empty_cache()
        except Exception:
            pass

    def finish(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def emit(line):
    """One JSON line on stdout, flushed: the driver parses the LAST line; every earlier one is the same line with
    fewer optional fields, so that whatever ends the process early leaves a complete headline behind."""
    try:
        # (RCCL prints a version banner through C stdio when its communicator is created; left in that buffer it would
        #  come out at exit, BEHIND the last JSON line — seen in the world-size-1 run of the RCCL path)
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.write(json.dumps(line) + "\n")
    sys.stdout.flush()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size-mb", type=int, default=1024, help="input MiB per GPU")
    ap.add_argument("--shard-kb", type=int, default=128, help="partition plan: KiB per encoder shard")
    ap.add_argument("--quality", type=int, default=5)
    ap.add_argument("--lgwin", type=int, default=22)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the legs behind the headline (device round trip, "
                    "BrotliEncoderCompress end to end, the stock calls)")
    ap.add_argument("--workload", choices=["text", "silesia"], default="text",
                    help="text: BASELINE configs[1] (enwik-style); silesia: configs[3] (Silesia-style mix, "
                         "size-mb MiB per GPU: 8 GiB on 8 GPUs)")
    ap.add_argument("--data", choices=["text", "random"], default="text", help="quality 1 only")
    ap.add_argument("--feed-kb", type=int, default=0, help="quality 1 only: KiB per CompressStream call (0 = one call)")
    ap.add_argument("--input-file", default=None, help="read the input from this file instead of generating it (the legs of "
                    "`other_configs` reuse the parent's text)")
    ap.add_argument("--bounded-baseline", action="store_true", help="cpu_baseline with one run of the same plan and no "
                    "scaling study (the legs of `other_configs`)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the bounded legs for BASELINE configs[2], [4] and "
                    "[3]'s workload (`other_configs` of the last line)")
    ap.add_argument("--all-configs", action="store_true",
                    help="one line per single-GPU BASELINE configuration (configs[1], [2], [4], and [3]'s workload on one "
                         "GPU), each with the reference timed beside it on this box's host cores")
    args = ap.parse_args(argv)
    if args.all_configs:
        return main_all_configs(args)
    if args.quality == 1:
        return main_q1(args)

    import gen_inputs as G

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                 % (args.gpus, args.gpus))
    job = GpuJob(args, rank, local_rank, world)

    n = args.size_mb << 20
    shard = args.shard_kb << 10
    total = n * world
    size_hint = min(total, 1 << 30)
    if args.input_file and os.path.exists(args.input_file) and os.path.getsize(args.input_file) == n and world == 1:
        data = open(args.input_file, "rb").read()
    else:
        data = G.enwik_text(n, seed=G.SEED + rank) if args.workload == "text" else G.mixed_corpus(n, seed=G.SEED + rank)
    job.load(data, args.quality, args.lgwin, shard, size_hint)

    for _ in range(args.warmup):
        job.step()
    job.barrier()
    t0 = time.perf_counter()
    infos = []
    nbytes = 0
    for _ in range(args.steps):
        nbytes, info = job.step()
        infos.append(info)
    job.barrier()
    dt = time.perf_counter() - t0
    dt, out_total, stream_check = job.reduce(dt, nbytes)

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = total / 1e6 / (dt / args.steps)
        def avg(k):
            return sum(i.get(k, 0.0) for i in infos) / len(infos)
        ms_parse, ms_index, ms_ixb = avg("ms_parse"), avg("ms_index"), avg("ms_ix_bucket")
        algo = ALGO_BYTES_PER_INPUT_BYTE[args.quality]
        indexed = args.quality == 5 and ms_ixb > 0
        if indexed and ms_ixb >= ms_parse:
            # (the HIP events bracket k_ix_bucket and k_ix_big, which searches the blocks of the buckets too big for one wave
            #  right behind it — a tenth of the positions of the bench text: rocprofv3 shows the two separately, 25.5 + 3.3 ms)
            kernel, k_ms, k_bytes = "k_ix_bucket+k_ix_big", ms_ixb, IX_BUCKET_BYTES_PER_INPUT_BYTE
        else:
            kernel = "k_chain" if indexed else ("k_parse4" if args.quality == 5 else "k_parse_quick" if args.quality < 5 else "k_parse_deep")
            k_ms, k_bytes = ms_parse, (algo if not indexed else 9.0 + 16.0 * 0.4)
        achieved = k_bytes * n / (k_ms / 1e3) / 1e9
        # HBM bytes of the dominant kernel: not measured by this process (the counters need rocprofv3 around it) — from
        # the PMC passes of the same command in this round's GPU sessions, calibrated on known byte counts
        # (profiles/traffic.json names its source), null for configurations without such a pass
        traffic, traffic_source = None, "not measured (no PMC pass on record for this configuration)"
        prof = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(prof):
            t = json.load(open(prof)).get(("" if args.quality == 5 else "q%d/" % args.quality) + "%d/%d" % (args.size_mb, args.shard_kb) +
                                          ("" if args.workload == "text" else "/silesia"))
            if t and t.get("kernel") == kernel:
                traffic, traffic_source = t["hbm_bytes_per_launch"], "model input, not this run: " + t["source"]
        path_ms = ms_index + ms_parse
        path = algo * n / (path_ms / 1e3) / 1e9
        line = {
            "metric": "encode MB/s at quality %d, lgwin %d, 1 GiB input; bit-exact vs c/enc" % (
                args.quality, args.lgwin),
            "value": round(value, 1), "unit": "MB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {
                "workload": "%d MiB synthetic %s per GPU (tests/gen_inputs.%s, "
                            "seed %d+rank), quality %d, lgwin %d, %d x MI355X" % (
                                args.size_mb, "enwik-style text" if args.workload == "text" else "Silesia-style mix",
                                "enwik_text" if args.workload == "text" else "mixed_corpus",
                                G.SEED, args.quality, args.lgwin, world),
                "partition_plan": "%d shards of %d KiB per GPU (STREAM_OFFSET contract); "
                                  "bytes identical to the reference driven with the same plan" % (
                                      infos[-1]["nshards"], args.shard_kb),
                "compressed_bytes": out_total, "ratio": round(total / out_total, 4),
                "gathered_stream": None if stream_check is None else {
                    "sha256": stream_check[1], "equal_on_all_ranks": stream_check[0],
                    "bytes": stream_check[2], "concatenation_inside_timed_region": True},
                "parse_ms_per_step": [round(i["ms_index"] + i["ms_parse"], 1) for i in infos],
                "stage_ms": {k: round(avg(k), 3) for k in
                             ("ms_total", "ms_init", "ms_index", "ms_ix_bucket", "ms_parse", "ms_build", "ms_store", "ms_gather")},
            },
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "frac_of_achievable_6300": round(achieved / HBM_ACHIEVABLE_GBS, 5),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "parse_path": {"kernels": "k_ix_count + k_ix_scan + k_ix_scatter + k_ix_bucket + k_chain + k_cmd_encode"
                                        if indexed else kernel,
                                        "ms": round(path_ms, 3), "achieved": round(path, 1),
                                        "frac": round(path / HBM_PEAK_GBS, 5),
                                        "model": "SURVEY.md 8(d): %.0f B per input byte for the whole LZ77 parse" % algo},
                         "note": "%s: algorithmic bytes = %.1f B per input byte x %d bytes per launch; kernel "
                                 "time %.3f ms (HIP events on the library's stream).  The kernel is bound by its "
                                 "vector instructions (SQ counters, DESIGN.md 5), not by bandwidth" % (
                                     kernel, k_bytes, n, k_ms)},
            "line": "1 of up to 3: the timed region (cpu_baseline, plans[] and the legs behind the headline follow)",
        }
        # (1) the headline exists: print it.  Everything below only adds fields to it.
        emit(line)
        if world == 1 and not args.no_cpu_baseline:
            import hashlib
            cfg = line["config"]
            comp = job.output_bytes(nbytes)
            cfg["gpu_output_sha256"] = hashlib.sha256(comp).hexdigest()
            # spot check of the bytes against the oracle on the first shards
            try:
                from refharness import Oracle
                o = Oracle()
                k = min(4, -(-n // shard))
                want = b"".join(o.encode_shard(data[i * shard:(i + 1) * shard], args.quality, args.lgwin,
                                               size_hint, min(i * shard, 1 << 30), (i + 1) * shard >= n)
                                for i in range(k))
                cfg["spot_check_first_shards_bit_exact"] = comp[:len(want)] == want
            except Exception as e:
                cfg["spot_check_first_shards_bit_exact"] = repr(e)[:200]
            del comp
            # config.plans[]: the headline plan and the other one of {128 KiB, 1 MiB} (the plan whose ratio is close
            # to the single stream's), each with ratio, MB/s, whole-output sha256 and the reference driven with the
            # same plan on this box's cores
            plans = [{"shard_KiB": args.shard_kb, "shards": infos[-1]["nshards"], "MBps": round(value, 1),
                      "ms_per_step": round(ms_step, 3), "ratio": round(total / out_total, 4), "compressed_bytes": out_total,
                      "sha256": cfg["gpu_output_sha256"], "headline": True}]
            other_kb = []
            if args.bounded_baseline:
                pass
            elif args.quality == 5 and args.workload == "text" and args.shard_kb in (128, 1024):
                other_kb = [1024 if args.shard_kb == 128 else 128]
            elif args.quality == 9 and args.shard_kb in (384, 512):
                other_kb = [512 if args.shard_kb == 384 else 384]       # (both plans, not the kinder one)
            for kb in other_kb:
                try:
                    plans.append(job.time_plan(args.quality, args.lgwin, kb, size_hint))
                except Exception as e:
                    plans.append({"shard_KiB": kb, "error": repr(e)[:300]})
            cfg["plans"] = plans
            # the device is not needed by the baseline: give it back before the host cores are timed
            job.close()
            path = data_file(data)
            try:
                cb = cpu_baseline(path, len(data), args.quality, args.lgwin, shard, size_hint,
                                  other_plans=[p["shard_KiB"] << 10 for p in plans[1:] if "error" not in p],
                                  bounded=args.bounded_baseline)
                line["cpu_baseline"] = cb
                for pl in plans:
                    ref = ({"MBps": cb["value"], "sha256": cb.get("sha256"), "out_bytes": cb.get("out_bytes"),
                            "seconds_all": cb.get("seconds_all")} if pl.get("headline")
                           else cb.get("other_plans", {}).get(str(pl["shard_KiB"])))
                    if ref and "MBps" in pl:
                        pl["reference_same_plan_MBps"] = ref["MBps"]
                        pl["reference_seconds_all"] = ref.get("seconds_all")
                        pl["x_reference_same_plan"] = round(pl["MBps"] / ref["MBps"], 2)     # (GPU mean of K steps / CPU median)
                        pl["sha256_equal_reference"] = ref.get("sha256") == pl["sha256"] and ref.get("out_bytes") == pl["compressed_bytes"]
                        ws = (cb.get("scaling") or {}).get("whole_socket_if_scaling_held_MBps")
                        if ws and pl.get("headline"):
                            # (an extrapolated denominator — cpu_baseline.scaling.whole_socket_note — beside the measured one)
                            pl["x_whole_socket_extrapolated"] = round(pl["MBps"] / ws, 2)
                # the reference encoded the WHOLE input with the same plan in this
                # run; its concatenated output must be the GPU's, byte for byte
                if "sha256" in cb:
                    cfg["parity_full_sha256_equal"] = (cb["sha256"] == cfg["gpu_output_sha256"] and cb["out_bytes"] == nbytes)
                cfg["single_stream_ratio"] = cb.get("single_stream_ratio")
                # (2) headline + baseline + plans + parity
                line["line"] = "2 of up to 3: with cpu_baseline, plans[] and the whole-output parity (the legs follow)"
                emit(line)
                if not args.no_legs:
                    legs = [("device_round_trip", "round_trip", (args.quality, args.lgwin, args.shard_kb, size_hint)),
                            ("end_to_end_abi", "abi", (args.quality, args.lgwin, args.shard_kb, nbytes, cfg["gpu_output_sha256"]))]
                    for key, name, largs in legs:
                        cfg[key] = run_leg(name, path, *largs)
                    sc = run_leg("stock", path, args.quality, args.lgwin)
                    if isinstance(sc, dict) and "error" not in sc:
                        sc["whole_input"] = run_leg("stock_whole", path, args.quality, args.lgwin)
                        sc["process_fed_256MiB"] = run_leg("process_fed", path, args.quality, args.lgwin, 1024, 256)
                        # what the reference's CLI does to a big file when no -w is given: lgwin 24 (c/tools/brotli.c:1434-1447)
                        sc["whole_input_cli_default_lgwin24"] = run_leg("stock_whole", path, args.quality, 24)
                        sc["concurrent_4MiB_calls"] = run_leg("concurrent", path, args.quality, args.lgwin, 4, 16, 4)
                    cfg["stock_call_no_plan"] = sc
                    line["line"] = "3 of 4: the legs of the headline configuration (other_configs follows)" if not args.no_other_configs else "3 of 3: complete"
                    emit(line)
                    if not args.no_other_configs and args.quality == 5 and args.workload == "text":
                        mode = (cb.get("scaling") or {}).get("best")
                        line["other_configs"] = other_configs(path, args.size_mb, mode)
                        line["line"] = "4 of 4: complete"
                        emit(line)
            finally:
                try:
                    os.unlink(path)
                except OSError:
                    pass
    job.finish()


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "--leg":
        leg_main(sys.argv[2:])
    else:
        main()
