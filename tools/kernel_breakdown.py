import csv, collections, sys
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    agg[r['Kernel_Name'][:78]].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
npre=len(agg[[k for k in agg if 'k_preproc' in k][0]])
tot=0; out=[]
for k,v in agg.items():
    per=len(v)/npre; us=sum(v)/len(v)/1e3
    out.append((us*per,per,us,k)); tot+=us*per
for a in sorted(out,reverse=True): print("%8.1f us/step %5.2f x %8.1f  %s"%a)
print("total %.1f us per step over %d steps"%(tot,npre))
