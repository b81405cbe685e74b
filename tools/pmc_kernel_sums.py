import csv, glob, sys, collections
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + '/pass*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:40]
        res[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[k][r['Counter_Name']] += 1
for k in res:
    print(k)
    for c in sorted(res[k]):
        print('   %-32s %16.0f  (%d rec)' % (c, res[k][c], cnt[k][c]))
