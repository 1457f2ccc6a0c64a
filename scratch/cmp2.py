import sys, torch
b = torch.load(sys.argv[1])
d = (b["out"] - b["out2"]).abs().amax(-1)     # [B, L]
bad = (d > 2e-6).nonzero()
print("bad entries", bad.shape[0], "of", d.numel())
rows = bad[:, 0].unique()
print("rows with bad", rows.numel(), rows[:20].tolist())
pos = bad[:, 1]
print("pos mod 64 histogram", torch.bincount(pos % 64, minlength=64).tolist())
print("pos // 64 histogram", torch.bincount(pos // 64, minlength=17).tolist())
r0 = rows[0].item()
print("row", r0, "bad positions", bad[bad[:, 0] == r0][:, 1].tolist()[:80])
