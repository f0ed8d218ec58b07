import csv, sys, collections, glob
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen=set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key=(r["Dispatch_Id"]); 
    if (k,key) not in seen: seen.add((k,key)); cnt[k]+=1
names = sorted({c for v in agg.values() for c in v})
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["kernel", "launches"] + names)
for k, n in cnt.most_common(14):
    w.writerow([k, n] + [round(agg[k][c] / n, 1) for c in names])
