import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('dsa::', '')
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    calls[(k, r['Counter_Name'])] += 1
for k, d in acc.items():
    print(k, {c: "%.3g" % (v / calls[(k, c)]) for c, v in d.items()})
