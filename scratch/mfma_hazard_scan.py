"""Scan a kernel listing for VALU / LDS-read writes to registers that a matrix instruction issued shortly before reads
(A, B or C operand) or writes: mfma_hazard_scan.py file.s kernel-substring [window]"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
win = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
ins = [(i, l.strip()) for i, l in enumerate(lines[start:end]) if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    if m: return {int(m.group(1))}
    return set()
def ops(text):
    body = text.split(None, 1)
    return [t.strip() for t in body[1].split(",")] if len(body) > 1 else []
found = 0
for k, (i, t) in enumerate(ins):
    if not t.startswith("v_mfma"): continue
    o = ops(t)
    dst, a, b, c = regs(o[0]), regs(o[1]), regs(o[2]), regs(o[3].split()[0])
    src = a | b | c
    n = 0
    for (i2, t2) in ins[k + 1:]:
        if t2.startswith("s_nop"):
            n += int(t2.split()[1]) + 1; continue
        n += 1
        if n > win: break
        if t2.startswith(("v_", "ds_read", "global_load", "scratch_load")) and not t2.startswith("v_cmp"):
            o2 = ops(t2)
            if not o2: continue
            d2 = regs(o2[0])
            if t2.startswith("v_mfma"):
                continue
            hit_src, hit_dst = d2 & src, d2 & dst
            if hit_src or hit_dst:
                found += 1
                print(f"line {i}: {t}\n   +{n}: line {i2}: {t2}   overlaps {'src' if hit_src else ''}{sorted(hit_src)} {'dst' if hit_dst else ''}{sorted(hit_dst)}")
print("candidates:", found)
