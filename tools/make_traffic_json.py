"""profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE lines of a session's pmc summaries:
    python tools/make_traffic_json.py SUMMARY_TEXT SUMMARY_MIX SOURCE_NOTE
FETCH_SIZE is doubled (gfx950 reports half the bytes of wide reads: calibrated in round 3, tools/write_calib.hip),
WRITE_SIZE taken as reported; both are KB per dispatch."""
import json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_dispatch(path, kernel, counter):
    for ln in open(path):
        m = re.search(r"PMC %s\(JobArgs\)\s+%s\s+dispatches=\d+ total=\S+ per_dispatch=(\S+)" % (kernel, counter), ln)
        if m:
            return float(m.group(1))
    return None


def entry(path, kernel, note):
    f, w = per_dispatch(path, kernel, "FETCH_SIZE"), per_dispatch(path, kernel, "WRITE_SIZE")
    if f is None or w is None:
        return None
    return {"kernel": kernel, "hbm_bytes_per_launch": int((2 * f + w) * 1024), "fetch_size_kb": f, "write_size_kb": w, "source": note}


if __name__ == "__main__":
    text, mix, note = sys.argv[1], sys.argv[2], sys.argv[3]
    out = {}
    e = entry(text, "k_ix_bucket", note)
    if e:
        out["1024/128"] = e
    if os.path.exists(mix):
        e = entry(mix, "k_chain", note + " (the Silesia-style mix: its dominant kernel is the chain)")
        if e:
            out["1024/128/silesia"] = e
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
