"""Per-basic-block instruction counts (vector / scalar / LDS / other) of one kernel from hipcc's assembly: where a kernel's
vector-issue budget goes, before any GPU run.  hipcc ... -S --cuda-device-only file.hip -o out.s ; python tools/isa_blocks.py out.s <kernel substring>"""
import re,sys
lines=open(sys.argv[1]).read().split('\n')
name=sys.argv[2]
start=[i for i,l in enumerate(lines) if l.startswith('_ZN5viorb') and name in l and ': ' in l][0]
end=[i for i,l in enumerate(lines) if i>start and '.Lfunc_end' in l][0]
body=lines[start+1:end]
cur='entry'; cnt={'entry':[0,0,0,0]}; order=['entry']
for l in body:
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: cur=m.group(1); cnt[cur]=[0,0,0,0]; order.append(cur); continue
    t=l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    op=t.split()[0]
    if op.startswith('v_'): cnt[cur][0]+=1
    elif op.startswith('s_'): cnt[cur][1]+=1
    elif op.startswith('ds_'): cnt[cur][2]+=1
    else: cnt[cur][3]+=1
    if op.startswith('s_cbranch') or op=='s_branch':
        cnt[cur].append(t.replace('\t',' '))
for b in order:
    c=cnt[b]
    print(b, 'V',c[0],'S',c[1],'DS',c[2],'oth',c[3], ' | '.join(c[4:]))
