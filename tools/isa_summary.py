import re, sys
fn=None; stats={}
for line in open(sys.argv[1]):
    m=re.match(r'^(_Z\w+):', line)
    if m: fn=m.group(1); stats[fn]=dict(n=0,sst=0,sld=0,call=0,valu=0,flat=0,mad=0); continue
    if fn is None: continue
    if line.startswith('.Lfunc_end'): fn=None; continue
    t=line.strip().split()
    if not t or not re.match(r'^[vsdfgb][a-z0-9_]+$', t[0]): continue
    s=stats[fn]; s['n']+=1; op=t[0]
    if op.startswith('scratch_store'): s['sst']+=1
    elif op.startswith('scratch_load'): s['sld']+=1
    elif op.startswith('s_swappc'): s['call']+=1
    elif op.startswith('flat_') or op.startswith('ds_') or op.startswith('global_'): s['flat']+=1
    if op.startswith('v_'): s['valu']+=1
    if op.startswith('v_mad_u64') or op.startswith('v_mad_i64'): s['mad']+=1
for k,s in stats.items():
    if s['n']>50: print("%6d valu %6d mad %5d sst %4d sld %4d mem %4d calls %3d  %s" % (s['n'],s['valu'],s['mad'],s['sst'],s['sld'],s['flat'],s['call'],k[:70]))
