"""Offline fuzz of the device decoder (k_decode.h) on the simulator: the reference ENCODER driven over its parameter
space (qualities 0-11, windows, modes, NPOSTFIX / NDIRECT, lgblock, literal context modelling off, flush patterns) makes the
streams, the decoder must give back the input and consume exactly the stream.  python tools/fuzz_decode_sim.py SEED COUNT"""
import sys, random, ctypes as C; import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from simharness import Sim
from refharness import Ref
import gen_inputs as G
sim=Sim(); R=Ref(); L=R.L
rng=random.Random(int(sys.argv[1]))
alice=open(os.path.join(ROOT, 'tests', 'golden', 'alice29.txt'), 'rb').read()
srcs=[alice, bytes(G.mixed_corpus(400000, seed=3)), bytes(G.enwik_text(200000, seed=9, vocab=4000)), bytes(G.random_bytes(30000))]
bad=0
for it in range(int(sys.argv[2])):
    src=rng.choice(srcs); a=rng.randrange(0,len(src)-1000); n=rng.choice([1,50,1000,20000,60000,150000]); data=src[a:a+n]
    q=rng.choice([0,1,2,4,5,6,9,10,11]); lgwin=rng.choice([10,12,16,18,22,24])
    if q>=10 and len(data)>60000: data=data[:60000]
    params=[(1,q),(2,lgwin)]
    if rng.random()<0.3: params.append((0,rng.choice([0,1,2])))          # mode
    if rng.random()<0.3 and q>=2:
        npf=rng.choice([0,1,2,3]); params.append((7,npf)); params.append((8,rng.choice([0,1,2,5,15])<<npf))
    if rng.random()<0.2: params.append((4,1))                               # disable literal context modelling
    if rng.random()<0.2 and q>=2: params.append((3,rng.choice([16,17,18,20])))   # lgblock
    st=L.BrotliEncoderCreateInstance(None,None,None)
    for k,v in params: L.BrotliEncoderSetParameter(st,k,v)
    cap=2*len(data)+4096; out=C.create_string_buffer(cap); buf=C.create_string_buffer(data,len(data))
    ao,no=C.c_size_t(cap),C.c_void_p(C.addressof(out)); off=0
    ops=[]
    while off<len(data):
        m=min(len(data)-off, rng.choice([len(data),5000,70000])); ops.append((off,m, 2 if off+m==len(data) else rng.choice([0,0,1]))); off+=m
    for (o,m,op) in ops:
        ai,ni=C.c_size_t(m),C.c_void_p(C.addressof(buf)+o)
        while True:
            assert L.BrotliEncoderCompressStream(st,op,C.byref(ai),C.byref(ni),C.byref(ao),C.byref(no),None)
            if ai.value==0 and not L.BrotliEncoderHasMoreOutput(st): break
    L.BrotliEncoderDestroyInstance(st)
    comp=out.raw[:cap-ao.value]
    got,res=sim.decode(comp,len(data),reverse=rng.choice([0,1,2,3]))
    ok = res[0][2]==0 and res[0][3]==1 and got==data and (res[0][1]+7)//8==len(comp)
    if not ok: bad+=1; print("MISMATCH", it, q, lgwin, params, len(data), res[0], flush=True)
print("done bad", bad)
