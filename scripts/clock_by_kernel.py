import csv,glob,re,sys
from collections import defaultdict
root=sys.argv[1]
dur={}
for f in glob.glob(root+"/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]]=(r["Kernel_Name"],int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
acc=defaultdict(list)
for f in glob.glob(root+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]!="GRBM_GUI_ACTIVE": continue
        k,d=dur.get(r["Dispatch_Id"],(r["Kernel_Name"],0))
        m=re.search(r"(k_\w+)",k)
        if m and d>200000: acc[m.group(1)+("<G2>" if "Fp2" in k else "")].append((float(r["Counter_Value"]),d))
for k,v in acc.items():
    c=sum(x for x,_ in v); d=sum(y for _,y in v)
    print(f"{k:32s} {len(v):3d} launches avg {d/len(v)/1e3:9.1f} us   clock {c/d/8*1e3:7.0f} MHz")
