"""gaps between consecutive kernels of the launch stream in one step of a rocprofv3 kernel trace, by (previous, next) kernel (tuning aid)"""
import csv,glob,re,collections
import sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'nchw_to_nhwc' in r['Kernel_Name']]
back=int(sys.argv[2]) if len(sys.argv)>2 else 14
st=rows[idx[-back-1]:idx[-back]]; t0=int(st[0]['Start_Timestamp'])
def short(n):
    m=re.match(r'_ZN4ieee(\d+)(.*)',n)
    if m: return m.group(2)[:int(m.group(1))]
    return n.split('(')[0].replace('void ','').replace('ieee::','').split('<')[0]
main=collections.Counter(r['Stream_Id'] for r in st).most_common(1)[0][0]
ks=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),short(r['Kernel_Name'])) for r in st if r['Stream_Id']==main]
pairs=collections.Counter(); tot=collections.Counter()
prev_end=ks[0][1]
allg=[]
for i in range(1,len(ks)):
    g=(ks[i][0]-prev_end)/1e3
    allg.append(g)
    if g>3:
        pairs[(ks[i-1][2],ks[i][2])]+=1; tot[(ks[i-1][2],ks[i][2])]+=g
    prev_end=max(prev_end,ks[i][1])
import statistics
print('median gap %.2f us, mean %.2f, n=%d, sum %.0f'%(statistics.median(allg),sum(allg)/len(allg),len(allg),sum(allg)))
for k,v in sorted(tot.items(), key=lambda kv:-kv[1])[:15]: print('%6.0f us  n=%2d  %s -> %s'%(v,pairs[k],k[0],k[1]))
