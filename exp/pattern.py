import re, sys
s=open(sys.argv[1]).read()
m=re.search(r'^(_ZN\S*vq_search_mfma_kernelILi256ELi1ELb1ELb0ELb0E\S*):', s, re.M)
start=m.start(); end=s.index('.end_amdhsa_kernel', start); body=s[start:end]
print(set(re.findall(r'; NumVgprs: \d+|; ScratchSize: \d+|; Occupancy: \d+', s[start:end+6000])))
lines=[l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.')]
def cls(l):
    op=l.split()[0]
    if op.startswith('v_mfma'): return 'M'
    if op.startswith('ds_read'): return 'L'
    if op.startswith('global_load_lds') or ('lds' in l and op.startswith('global_load')): return 'D'
    if op.startswith('global_') or op.startswith('buffer_'): return 'G'
    if op.startswith('s_barrier'): return 'B'
    if op.startswith('s_waitcnt'): return 'w'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'J'
    if op.startswith('s_'): return 's'
    if op.startswith('v_'): return 'v'
    return '?'
pat=''.join(cls(l) for l in lines)
i=pat.index('MM')
out=[];prev=None;n=0
for c in pat[i:i+int(sys.argv[2]) if len(sys.argv)>2 else None]:
    if c==prev: n+=1
    else:
        if prev: out.append(f'{prev}{n if n>1 else ""}')
        prev=c;n=1
print(' '.join(out))
