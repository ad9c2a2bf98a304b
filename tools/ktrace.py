import csv,collections,sys
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.OrderedDict()
for r in rows:
    k=r["Kernel_Name"][:80]+" grid="+r["Grid_Size_X"]+"x"+r["Grid_Size_Y"]
    d.setdefault(k,[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in d.items(): print(f"{sum(v)/len(v):8.1f} us x{len(v):3d}  {k}")
