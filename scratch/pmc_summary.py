import csv, glob, sys, collections
tag=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f'gpurun_out/pmc_{tag}/p*/vm/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'moe_gemm' not in k: continue
        name='gate_up' if 'ILi0E' in k else 'down'
        agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
for name,cs in agg.items():
    print(name)
    for c,v in cs.items():
        print(f"  {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
