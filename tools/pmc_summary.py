"""Per-kernel means of rocprofv3 --pmc counter CSVs (p_counter_collection.csv ...)."""
import collections, csv, sys
for path in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    print('==', path)
    for kn, c in agg.items():
        if not any(w in kn for w in ('nd_amd',)):
            continue
        print(kn)
        for n, v in c.items():
            print('   %-22s n=%d mean=%.5g' % (n, len(v), sum(v) / len(v)))
