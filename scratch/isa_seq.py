"""Instruction-class sequence of the MFMA-richest loop of one kernel in a hipcc -S listing: isa_seq.py file.s kernel-substring
M bf16 MFMA, F fp32 MFMA, v VALU, P packed-f32 VALU, l LDS read, w LDS write, g global, W s_waitcnt, n s_nop, s SALU"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = None
for i, l in enumerate(lines):
    if re.match(r"^_Z\S*:", l) and key in l:
        start = i
        break
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end + 1]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\S+):", l))}
best = None
for i, l in enumerate(body):
    m = re.match(r"\s+s_cbranch_\S+\s+(\.LBB\S+)", l) or re.match(r"\s+s_branch\s+(\.LBB\S+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        a, b = labels[m.group(1)], i
        nm = sum(1 for x in body[a:b] if "v_mfma" in x)
        if best is None or nm > best[2] or (nm == best[2] and b - a < best[1] - best[0]):
            best = (a, b, nm)
seq = []
for l in body[best[0]:best[1] + 1]:
    l = l.strip()
    if not l or l.startswith((";", ".", "//")) or l.endswith(":"):
        continue
    op = l.split()[0]
    if op.startswith("v_mfma"):
        c = "M" if "bf16" in op else "F"
    elif op.startswith("ds_read") or op.startswith("ds_load"):
        c = "l"
    elif op.startswith("ds_"):
        c = "w"
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        c = "g"
    elif op.startswith("v_pk_") and "f32" in op:
        c = "P"
    elif op.startswith("v_"):
        c = "v"
    elif op.startswith("s_waitcnt"):
        c = "W"
    elif op.startswith("s_nop"):
        c = "n"
    elif op.startswith("s_"):
        c = "s"
    else:
        c = "?"
    seq.append(c)
s = "".join(seq)
for i in range(0, len(s), 100):
    print(s[i:i + 100])
