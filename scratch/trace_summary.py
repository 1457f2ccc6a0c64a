"""Busy time, gaps and small-kernel share of the steady-state part of a kernel trace: trace_summary.py <kernel_trace.csv> <steps>."""
import csv
import re
import sys
import collections

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
ia = [i for i, e in enumerate(ev) if 'k_gp_a_mfma' in e[2] or 'k_gp_fused_mfma' in e[2]]
# last `steps` steps: 5 penalty passes each
a = ia[-5 * steps]
sub = ev[a:]
span = sub[-1][1] - sub[0][0]
busy = sum(e[1] - e[0] for e in sub)
small = [e for e in sub if e[1] - e[0] < 30000]
print(f"per step: span {span / steps / 1e6:.3f} ms, busy {busy / steps / 1e6:.3f} ms, gaps {(span - busy) / steps / 1e6:.3f} ms, "
      f"{len(sub) / steps:.0f} launches, small kernels {len(small) / steps:.0f} = {sum(e[1] - e[0] for e in small) / steps / 1e6:.3f} ms")
acc = collections.defaultdict(lambda: [0, 0])
for e in sub:
    n = re.sub(r'\(anonymous namespace\)::', '', e[2])
    n = re.sub(r'\(.*', '', n.replace('void ', '').replace('at::native::', ''))[:70]
    acc[n][0] += 1
    acc[n][1] += e[1] - e[0]
for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{t / steps / 1e6:8.3f} ms {c / steps:7.1f} x {t / c / 1e3:8.1f} us  {n}")
