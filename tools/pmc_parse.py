import csv, glob, sys, collections, re
d=sys.argv[1]; pat=re.compile(sys.argv[2])
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for fn in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        m=pat.search(r["Kernel_Name"])
        if m:
            k=m.group(0)
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
            n[(k,r["Counter_Name"])]+=1
for k in acc:
    print(k)
    for c,v in sorted(acc[k].items()):
        print(f"   {c:32s} {v/n[(k,c)]:16.1f} per launch ({n[(k,c)]} launches)")
