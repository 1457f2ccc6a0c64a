"""List row swaps (v_permlane16/32_swap) of a kernel whose operands were written by a packed-f32 instruction within the
last N instructions: pk_swap_scan.py file.s kernel-substring [N]"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
win = int(sys.argv[3]) if len(sys.argv) > 3 else 12
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
ins = [(i, l.strip()) for i, l in enumerate(lines[start:end]) if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()
def ops(t):
    b = t.split(None, 1)
    return [x.strip() for x in b[1].split(",")] if len(b) > 1 else []
n = hits = 0
for k, (i, t) in enumerate(ins):
    if not t.startswith("v_permlane"): continue
    n += 1
    use = set().union(*[regs(o) for o in ops(t)])
    for r in sorted(use):            # nearest writer of each operand
        for d in range(1, win + 1):
            if k - d < 0: break
            i2, t2 = ins[k - d]
            o2 = ops(t2)
            if t2.startswith(("v_", "ds_read", "global_load")) and o2 and r in regs(o2[0]) | (regs(o2[1]) if t2.startswith("v_permlane") else set()):
                if t2.startswith("v_pk_") and "f32" in t2:
                    hits += 1
                    print(f"swap at {i}: {t}   <- -{d}: {t2}")
                break
print("swaps", n, "fed by packed f32 ops:", hits)
