import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
print("absmax", a["absmax"].tolist(), b["absmax"].tolist())
step = float(a["absmax"][3]) / 32767
for k in ("out", "out2", "yb"):
    d = (b[k].double() - a[k].double())
    n = d.numel()
    print(k, "max |d| %.3e  mean d %.3e  mismatched %.4f%%  d>0 %.4f%% d<0 %.4f%%  (step %.3e)  sum a %.6f sum b %.6f" % (
        d.abs().max(), d.mean(), 100 * (d != 0).sum() / n, 100 * (d > 0).sum() / n, 100 * (d < 0).sum() / n, step, a[k].double().sum(), b[k].double().sum()))
print("out vs out2 in a:", (a["out"] - a["out2"]).abs().max().item(), " in b:", (b["out"] - b["out2"]).abs().max().item())
