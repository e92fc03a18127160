"""Runs of consecutive VOP2 `v_cndmask_b32 ..., vcc` in hipcc's assembly, per kernel. On gfx950 a VOP2 (e32) v_cndmask issues in ~2 cycles
when another VALU instruction separates it from the previous one, but back-to-back e32 v_cndmasks cost ~4 each for two and 10-18 cycles
each from the third on (tools/ubench/cndmask_vcc.hip, profiles/r04_ubench_cndmask_vcc.txt); the VOP3 (e64) form costs 4.2 in any
context. usage: python tools/isa_cndmask_runs.py out.s [kernel substring]"""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
sel = sys.argv[2] if len(sys.argv) > 2 else ''
cur = None; run = 0; hist = {}
def close():
    global run
    if cur is not None and run: hist[cur][run] += 1
    run = 0
for l in lines:
    m = re.match(r'^(_Z\w+):', l)
    if m:
        close(); cur = m.group(1); hist[cur] = collections.Counter(); continue
    if cur is None: continue
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    op = t.split()[0]
    if op.startswith('v_cndmask_b32') and t.rstrip().endswith('vcc') and '_e64' not in op:
        run += 1
    elif op.startswith('v_') or op.startswith('ds_') or op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_') or op.startswith('scratch_'):
        close()
    # scalar instructions do not close a run (they issue from another port)
close()
for k, h in hist.items():
    if sel in k and sum(h.values()):
        tot = sum(n * c for n, c in h.items()); bad = sum((n - 2) * c for n, c in h.items() if n > 2)
        print('%-70s e32 cndmask %4d, in runs >= 3: %4d instructions beyond the second  %s' % (k[:70], tot, bad, dict(sorted(h.items()))))
