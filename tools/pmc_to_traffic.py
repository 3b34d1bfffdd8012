#!/usr/bin/env python
"""profiles/<tag>_pmc_hbm.txt (tools/pmc_summary.py output of the FETCH_SIZE and WRITE_SIZE passes) -> profiles/pmc_traffic.json:
HBM-side bytes per launch for the conv kernel classes bench.py reports, with the gfx950 correction of MI355X_MICROARCH.md
(FETCH_SIZE counts wide coalesced reads at half their size): traffic = 2 * FETCH_SIZE + WRITE_SIZE, counters in KB."""
import json, os, re, sys
src = sys.argv[1]
prefix = sys.argv[2] if len(sys.argv) > 2 else "bf16:"     # key prefix of the 2-byte trunk's classes: "bf16:" (single-pass builds) | "x3:" (split precision)
sec, cur, vals = None, None, {}
for line in open(src):
    if line.startswith("## "):
        sec = line[3:].strip(); continue
    if line.startswith("#") or not line.strip():
        continue
    if not line.startswith(" "):
        cur = line.strip(); continue
    m = re.match(r"\s+(\w+)\s+([0-9.]+)\s+\(n=(\d+)\)", line)
    if m and cur:
        vals.setdefault(cur, {})[m.group(1)] = (float(m.group(2)), int(m.group(3)))
CLASS = [("conv_wgrad_x3_kernel<64>", "conv_wgrad_3x3_n64"), ("conv_fwd_x3_kernel<2, 2, 1, 32, true>", "conv_fwd_3x3_n64"),
         ("conv_dgrad_x3_kernel<2, 2, 1, 32, true>", "conv_dgrad_3x3_n64"), ("conv_wgrad_x3_kernel<32>", "conv_wgrad_3x3_n32"),
         ("conv_fwd_kernel<4, 1, 1, 32, false>", "conv_fwd_3x3_n32"), ("conv_dgrad_x3_kernel<4, 1, 1, 32, false>", "conv_dgrad_3x3_n32")]
out = {}
if os.path.exists("profiles/pmc_traffic.json"):                # other modes' entries stay
    out = json.load(open("profiles/pmc_traffic.json"))
out.pop(prefix + "stem16_wgrad_kernel", None)               # (round 5's key of the stem's kernel-gradient pass: "stem16_wgrad" since round 6)
out.update({"source": src, "note": "per-launch average over all dispatches of the kernel in a step (3x3 and 1x1 launches of one template together); "
       "traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024"})
# bf16 trunk: bench.py's classes ARE the kernel instantiations
for k, d in vals.items():
    m = re.search(r"((?:conv_bf16_kernel|wgrad3_bf16_kernel|wgrad1_bf16_kernel)(?:<[^>]*>)?)", k)
    if m and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        f, w = d["FETCH_SIZE"][0], d["WRITE_SIZE"][0]
        out[prefix + m.group(1)] = {"source": src, "kernel": m.group(1), "fetch_size_kb": f, "write_size_kb": w, "traffic_bytes": int((2 * f + w) * 1024), "dispatches": d["FETCH_SIZE"][1]}
    # (round 6) the HBM-bound passes bench.py lists beside the conv classes: the stem's four passes and the weight gradients' reduction, keyed as bench.py names them
    m = re.search(r"(stem16_stats_kernel|stem16_pool_kernel|stem16_bwd_reduce|stem16_wgrad(?=2?_kernel)|wgrad_reduce_multi_kernel)", k)
    if m and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        f, w = d["FETCH_SIZE"][0], d["WRITE_SIZE"][0]
        out[prefix + m.group(1)] = {"source": src, "kernel": m.group(1), "fetch_size_kb": f, "write_size_kb": w, "traffic_bytes": int((2 * f + w) * 1024), "dispatches": d["FETCH_SIZE"][1]}
for kname, cls in CLASS:
    for k, d in vals.items():
        if kname in k and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            f, w = d["FETCH_SIZE"][0], d["WRITE_SIZE"][0]
            out[cls] = {"kernel": kname, "fetch_size_kb": f, "write_size_kb": w, "traffic_bytes": int((2 * f + w) * 1024), "dispatches": d["FETCH_SIZE"][1]}
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
