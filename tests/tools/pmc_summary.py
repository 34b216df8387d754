import csv, collections, sys
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in rows:
    k=r["Kernel_Name"][:24]
    if "dune" not in k and "nrmp" not in k: continue
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
for k in agg:
    print(k, {c: round(v/n[(k,c)]) for c,v in agg[k].items()})
