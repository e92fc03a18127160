"""Sums the per-phase s_memtime ticks a -DVIORB_FAST_TIMING build of k_fast_cells3 prints (one line per sampled workgroup)."""
import re, sys, collections
acc = collections.Counter(); n = 0
for l in open(sys.argv[1]):
    if "fast3" not in l: continue
    for k, v in re.findall(r"([a-z][\w+]*)=(\d+)", l): acc[k] += int(v)
    n += 1
cells = max(acc["cells"], 1)
print("workgroups", n, "cells", cells, "second attempts per cell %.2f  survivors per cell %.1f  corners per cell %.1f" % (acc["att2"] / cells, acc["surv"] / cells, acc["corn"] / cells))
keys = ("tile+clear", "setup", "pass1", "pass2", "tail", "nms", "end", "loop")
tot = sum(acc[k] for k in keys)
for k in keys: print("%-12s %6.1f %%  %8.0f ticks per cell" % (k, 100 * acc[k] / tot, acc[k] / cells))
print("total ticks per cell", tot / cells)
