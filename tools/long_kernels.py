import csv,sys
rows=[]
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name=r["Kernel_Name"].split("(")[0].replace("void ","").replace("bf::","")
        rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),name,int(r.get("Grid_Size_X",0))//max(int(r.get("Workgroup_Size_X",1)),1),int(r.get("Grid_Size_Y",1))))
rows.sort()
starts=[i for i,r in enumerate(rows) if r[2].startswith("k_one_hot") and (i==0 or not rows[i-1][2].startswith("k_one_hot"))]
rows=rows[starts[-1]:]
t0=rows[0][0]
for i,(s,e,n,bx,by) in enumerate(rows):
    if n.startswith("k_fft") and (e-s)>100000:
        print(f"+{(s-t0)/1e6:7.3f} ms {n:28s} {bx:6d} x {by:3d} {(e-s)/1e3:8.1f} us   prev: {rows[i-1][2]}")
