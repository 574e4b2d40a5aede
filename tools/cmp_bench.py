import json,sys
a=json.load(open(sys.argv[1]));b=json.load(open(sys.argv[2]))
ka={x['entry']:x for x in a['roofline']['kernels']};kb={x['entry']:x for x in b['roofline']['kernels']}
rows=[]
for n in set(ka)|set(kb):
    ta=ka.get(n,{}).get('ms_per_step',0);tb=kb.get(n,{}).get('ms_per_step',0)
    rows.append((tb-ta,n,ta,tb))
rows.sort(reverse=True)
print(a['ms_per_step'],b['ms_per_step'])
for d,n,ta,tb in rows[:12]: print("%-28s %7.3f -> %7.3f  (%+.3f)"%(n,ta,tb,d))
