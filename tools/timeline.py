import csv, sys, glob
f = sorted(glob.glob(sys.argv[1]))[-1]
rows=list(csv.DictReader(open(f)))
names=[(r['Kernel_Name'].replace('void ','').replace('fh::(anonymous namespace)::','').split('(')[0], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
names.sort(key=lambda x:x[1])
gen=[i for i,n in enumerate(names) if n[0]=='k_generate']
i0=gen[len(gen)//2]; i1=gen[len(gen)//2+1] if len(gen)//2+1 < len(gen) else len(names)
t0=names[i0][1]; prev=None
for n,s,e in names[i0:i1]:
    gap=(s-prev)/1000 if prev else 0
    print(f"{n:36s} start {(s-t0)/1000:9.1f} dur {(e-s)/1000:8.1f} gap {gap:6.1f}")
    prev=e
print("total us", (names[i1-1][2]-t0)/1000)
