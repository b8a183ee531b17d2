"""profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE lines of a round's pmc summaries (tools/gpu_r06.sh pmc):
    python tools/make_traffic_json.py ROUND_TAG            (reads profiles/<ROUND_TAG>_pmc_{text,mix,q9,q1}_summary.txt)
FETCH_SIZE is doubled (gfx950 reports half the bytes of wide reads: calibrated in round 3, tools/write_calib.hip — a
coalesced 4-byte read of 1 GiB counts 0.5 GiB), WRITE_SIZE taken as reported (coalesced 16-byte stores of 1 GiB count
1 GiB); both are KB per dispatch."""
import json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_dispatch(path, kernel, counter):
    if not os.path.exists(path):
        return None
    for ln in open(path):
        if "PMC" in ln and kernel in ln and counter in ln:
            m = re.search(r"per_dispatch=(\S+)", ln)
            if m:
                return float(m.group(1))
    return None


def entry(path, kernel, note):
    f, w = per_dispatch(path, kernel + "(", "FETCH_SIZE"), per_dispatch(path, kernel + "(", "WRITE_SIZE")
    if f is None:
        f, w = per_dispatch(path, kernel + "<", "FETCH_SIZE"), per_dispatch(path, kernel + "<", "WRITE_SIZE")
    if f is None or w is None:
        return None
    return {"kernel": kernel, "hbm_bytes_per_launch": int((2 * f + w) * 1024), "fetch_size_kb": f, "write_size_kb": w,
            "source": note % os.path.relpath(path, ROOT)}


if __name__ == "__main__":
    tag = sys.argv[1]
    note = ("%s (tools/gpu_r06.sh pmc: bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-legs with this configuration's "
            "flags under separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes, per dispatch, at the round's final "
            "kernels).  FETCH_SIZE doubled (gfx950 reports half the bytes: calibrated in round 3 with tools/write_calib.hip); "
            "WRITE_SIZE as reported (same calibration)")
    out = {}
    for key, name, kernel in (("1024/128", "text", "k_ix_bucket"), ("1024/128/silesia", "mix", "k_chain"),
                              ("q9/1024/512", "q9", "k_parse_deep"), ("q1/1024/random", "q1", "k_fast_parse")):
        path = os.path.join(ROOT, "profiles", "%s_pmc_%s_summary.txt" % (tag, name))
        e = entry(path, kernel, note)
        if e and kernel == "k_ix_bucket":
            # the window search of the buckets too big for one wave runs in k_ix_big right behind it: bench.py's HIP events
            # bracket the pair (ms_ix_bucket), so does this entry
            b = entry(path, "k_ix_big", note)
            if b:
                e["kernel"] = "k_ix_bucket+k_ix_big"
                for k in ("hbm_bytes_per_launch", "fetch_size_kb", "write_size_kb"):
                    e[k] += b[k]
        if e:
            out[key] = e
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps({k: (v["kernel"], v["hbm_bytes_per_launch"]) for k, v in out.items()}, indent=1))
