"""tools/find_pinned_structs.py KERNEL_SYMBOL [dev.ll]: which stack objects a kernel keeps in scratch memory, and the
accesses that pin them there (a member indexed through a computed address: SROA then gives up the whole struct).
dev.ll: the device LLVM IR of the library,
  hipcc -O3 -ffp-contract=off -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -emit-llvm -Iinclude \
        brotli_amd/csrc/hip_layer.hip -o dev.ll
Round 6 found k_parse_deep / k_parse_quick / k_parse / k_store / k_wide_* this way."""
import re,sys
kern=sys.argv[1]
txt=open(sys.argv[2] if len(sys.argv) > 2 else 'dev.ll').read().splitlines()
start=[i for i,l in enumerate(txt) if l.startswith('define') and kern in l][0]
end=next(i for i in range(start,len(txt)) if txt[i]=='}')
body=txt[start:end]
allocas=[re.match(r'\s+(%\d+) = alloca (.*?),',l).groups() for l in body if ' = alloca ' in l]
print(allocas)
def uses(name):
    return [l for l in body if re.search(re.escape(name)+r'(?!\d)',l) and not l.strip().startswith(name+' =')]
for a,ty in allocas:
    frontier=[(a,0)]
    seen=set()
    while frontier:
        n,depth=frontier.pop()
        if n in seen: continue
        seen.add(n)
        for l in uses(n):
            ls=l.strip()
            m=re.match(r'(%\d+) = getelementptr (inbounds )?(nuw )?(\S+), ptr addrspace\(5\) '+re.escape(n)+r', (.*)',ls)
            if m:
                idx=m.group(5)
                if re.search(r'%\d+',idx):
                    print('DYNAMIC GEP on',a,ty,':',ls[:160])
                frontier.append((m.group(1),depth+1))
                continue
            if re.match(r'(%\d+ = )?(load|store) ',ls) and not re.search(r'store ptr addrspace\(5\) '+re.escape(n),ls):
                continue
            if 'lifetime' in ls: continue
            print('ODD use of',n,'(from',a,ty,'):',ls[:200])
