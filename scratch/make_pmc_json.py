"""Turn the per-kernel counter averages (pmc_summary.py output) into profiles/<name>.json:
make_pmc_json.py <summary.txt> <out.json> "<command>".  HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB:
on gfx950 FETCH_SIZE tallies the 128-byte requests of a wide coalesced read stream at 64 bytes, WRITE_SIZE is exact."""
import json
import re
import sys

src, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
res = {"command": cmd,
       "correction": "hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: gfx950 counts a 128-byte read request as 64 bytes; "
                     "WRITE_SIZE taken as is; per launch, averaged over the launches of the run"}
for line in open(src):
    m = re.match(r"(.+?) \(n=(\d+)\): (.*)", line.strip())
    if not m:
        continue
    name = m.group(1).replace(", ", ",").strip()
    c = {k: float(v) for k, v in (kv.split("=") for kv in m.group(3).split())}
    e = {"launches": int(m.group(2)), **c}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["hbm_read_bytes"] = int(2 * c["FETCH_SIZE"] * 1024)
        e["hbm_write_bytes"] = int(c["WRITE_SIZE"] * 1024)
        e["hbm_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
    res[name] = e
if "w1024::k_stats" in res and "w1024::k_bwd" in res and "hbm_bytes" in res["w1024::k_stats"]:
    res["spectral"] = {"hbm_bytes": res["w1024::k_stats"]["hbm_bytes"] + res["w1024::k_bwd"]["hbm_bytes"],
                       "note": "k_stats + k_bwd, the two streaming kernels of SpectralLoss forward + backward"}
json.dump(res, open(out, "w"), indent=1)
print("wrote", out, len(res) - 2, "kernels")
