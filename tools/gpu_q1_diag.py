"""Diagnostic: quality-1 job on 1 GiB of random bytes, alone and after other jobs on the same context."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gen_inputs as G
from brotli_amd import hip
n = int(os.environ.get("PROBE_MB", "1024")) << 20
ctx = hip.Context(0)
g = torch.Generator(device="cuda").manual_seed(G.SEED)
d_in = torch.zeros(n + hip.INPUT_SLACK, dtype=torch.uint8, device="cuda:0")
d_in[:n] = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda:0", generator=g)
d_out = torch.empty(ctx.fast_max_output(n, 1, 22), dtype=torch.uint8, device="cuda:0")
def q1(tag):
    nbits, info = ctx.encode_fast_device(d_in, n, d_out, 22)
    print("Q1DIAG", tag, nbits // 8, info["nshards"], flush=True)
q1("fresh"); q1("again")
text = G.enwik_text(256 << 20)
d_t = hip.to_device(text)
for q, w, sh in ((5, 22, 1 << 17), (9, 24, 1 << 19)):
    p = hip.make_params(q, w, sh)
    o = torch.empty(ctx.max_output(len(text), p), dtype=torch.uint8, device="cuda:0")
    nb, info = ctx.encode_device(d_t, len(text), p, o)
    print("Q1DIAG job q%d out %d ws %d" % (q, nb, info["ws_bytes"]), flush=True)
    del o
    q1("after_q%d" % q)
