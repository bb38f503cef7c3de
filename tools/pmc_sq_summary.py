import csv, sys, glob, collections, re
csv.field_size_limit(10**9)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p, newline='')):
        k = re.sub(r'\(.*$', '', r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', ''))[:60]
        if not k.startswith('vq'): continue
        acc[(k, r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for (k, g), cs in acc.items():
    print(k, 'grid', g)
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    for c, v in sorted(m.items()):
        print(f'   {c:28s} {v:16.0f}')
    if 'SQ_WAVE_CYCLES' in m:
        wc = m['SQ_WAVE_CYCLES']
        for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY'):
            if c in m: print(f'   {c}/WAVE_CYCLES = {m[c] / wc:.3f}')
