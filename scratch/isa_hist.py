"""Histogram of the instructions of one kernel in a hipcc -S listing: isa_hist.py file.s kernel-substring [--loop]
--loop: only the largest loop body (from the label the last backward branch targets to that branch)."""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
only_loop = "--loop" in sys.argv
lines = open(path).read().split("\n")
start = end = None
for i, l in enumerate(lines):
    if re.match(r"^_Z\S*:", l) and key in l:
        start = i
    if start is not None and end is None and l.strip().startswith(".end_amdhsa_kernel") or (start is not None and l.startswith("\ts_endpgm") and end is None):
        pass
if start is None:
    sys.exit("kernel not found")
for i in range(start, len(lines)):
    if lines[i].strip().startswith("s_endpgm"):
        end = i
        break
body = lines[start:end + 1]
if only_loop:
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\S+):", l)
        if m:
            labels[m.group(1)] = i
    best = None
    for i, l in enumerate(body):
        m = re.match(r"\s+s_cbranch_\S+\s+(\.LBB\S+)", l) or re.match(r"\s+s_branch\s+(\.LBB\S+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = (labels[m.group(1)], i)
            nm = sum(1 for x in body[span[0]:span[1]] if 'v_mfma' in x)
            if best is None or nm > best[2] or (nm == best[2] and span[1] - span[0] > best[1] - best[0]):   # the outermost of the loops that hold the MFMAs
                best = (span[0], span[1], nm)
    body = body[best[0]:best[1] + 1]
    print("loop lines", best)
hist = collections.Counter()
for l in body:
    l = l.strip()
    if not l or l.startswith((";", ".", "//")) or l.endswith(":"):
        continue
    op = l.split()[0]
    hist[op] += 1
groups = collections.Counter()
for op, n in hist.items():
    if op.startswith("v_mfma"):
        g = "mfma"
    elif op.startswith("ds_"):
        g = "lds"
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        g = "vmem"
    elif op.startswith("v_"):
        g = "valu"
    elif op.startswith("s_waitcnt"):
        g = "waitcnt"
    elif op.startswith("s_nop"):
        g = "nop"
    elif op.startswith("s_"):
        g = "salu"
    else:
        g = "other"
    groups[g] += n
print(dict(groups))
for op, n in hist.most_common():
    print(f"{n:5d} {op}")
