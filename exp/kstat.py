import csv, sys
csv.field_size_limit(10**9)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r['Name']:
        print(r['Name'][:70], 'calls', r['Calls'], 'avg_us %.1f min_us %.1f' % (float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
