#!/usr/bin/env python
"""profiles/<tag>_pmc_{rd,ws,fs,wr}.csv (rocpd summaries of the four PMC passes of tools/profile_round.sh) -> the per-launch
L2->fabric bytes of the step's kernels in profiles/pmc_traffic.json, which bench.py reports as roofline.traffic.
    python tools/pmc_traffic.py r02 [commit]"""
import csv, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT).decode().strip()


def load(name):
    out, sec = {}, None
    for line in open(os.path.join(ROOT, "profiles", f"{tag}_{name}.csv")):
        if line.startswith("# PMC"):
            sec = "pmc"; continue
        if line.startswith("#"):
            sec = "k"; continue
        if line.startswith("name,"):
            continue
        row = next(csv.reader([line]))
        if sec == "pmc" and len(row) >= 5:
            out.setdefault(row[0], {})[row[1]] = (int(row[2]), float(row[3]))
    return out


rd, ws, fs = load("pmc_rd"), load("pmc_ws"), load("pmc_fs")
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
doc = json.load(open(path))
prod = doc.setdefault("products", {})
for k, r in rd.items():
    short = "gat_fused_rows_kernel" if "gat_fused_rows" in k else ("fused_conv_kernel" if "fused_conv_kernel" in k else None)
    if short is None:
        continue
    req, r128 = r["TCC_EA0_RDREQ_sum"][1], r["TCC_EA0_RDREQ_128B_sum"][1]
    assert abs(req - r128) <= 1e-3 * req, "not every L2->fabric read is a 128-byte request: byte count needs the size split"
    rb, fb = req * 128, 2 * fs[k]["FETCH_SIZE"][1] * 1024
    assert abs(rb - fb) <= 5e-3 * rb, (rb, fb)          # the gfx950 FETCH_SIZE correction (MI355X_MICROARCH.md, HBM)
    prod[short] = {"hbm_read_bytes": int(rb), "hbm_write_bytes": int(ws[k]["WRITE_SIZE"][1] * 1024),
                   "source": f"profiles/{tag}_pmc_rd.csv (TCC_EA0_RDREQ_sum x 128 B; = 2 x FETCH_SIZE x 1024 of {tag}_pmc_fs.csv), "
                             f"profiles/{tag}_pmc_ws.csv (WRITE_SIZE x 1024)",
                   "round": tag, "commit": commit, "launches_averaged": r["TCC_EA0_RDREQ_sum"][0]}
    print(short, f"read {rb / 1e9:.2f} GB write {prod[short]['hbm_write_bytes'] / 1e9:.2f} GB")
# the other configs' kernels (round 5): profiles/<tag>_pmc_<workload>_{rd,fs,ws}.csv, every kernel of ours that moves more than 1 MB
for wl in ("sage", "arxiv", "batched"):
    try:
        rd = {}
        for name in ("rd", "ws", "fs"):
            rd[name] = load(f"pmc_{wl}_{name}")
    except FileNotFoundError:
        continue
    sec = doc[wl] = {}
    for k, r in rd["rd"].items():
        if "gnnmp::" not in k or k not in rd["ws"] or k not in rd["fs"]:
            continue
        req = r["TCC_EA0_RDREQ_sum"][1]
        r128, r64, r32 = r["TCC_EA0_RDREQ_128B_sum"][1], r["TCC_EA0_RDREQ_64B_sum"][1], r["TCC_EA0_RDREQ_32B_sum"][1]
        rb = r128 * 128 + r64 * 64 + r32 * 32 + max(0.0, req - r128 - r64 - r32) * 64      # (requests of unlisted size: 64-byte ones)
        wb = rd["ws"][k]["WRITE_SIZE"][1] * 1024
        if rb + wb < 1e6 or r["TCC_EA0_RDREQ_sum"][0] < 10:      # (graph prep and other one-off launches: not the layer's kernels)
            continue
        short = k.split("gnnmp::")[1].split("(")[0]
        sec[short] = {"hbm_read_bytes": int(rb), "hbm_write_bytes": int(wb), "fetch_size_x2_bytes": int(2 * rd["fs"][k]["FETCH_SIZE"][1] * 1024),
                      "read_requests": int(req), "read_requests_128B": int(r128),
                      "source": f"profiles/{tag}_pmc_{wl}_rd.csv (TCC_EA0_RDREQ by size), {tag}_pmc_{wl}_fs.csv (FETCH_SIZE, x 2 x 1024: the gfx950 "
                                f"correction), {tag}_pmc_{wl}_ws.csv (WRITE_SIZE x 1024); tools/small_configs.py {wl}",
                      "round": tag, "commit": commit, "launches_averaged": r["TCC_EA0_RDREQ_sum"][0]}
        print(wl, short[:60], f"read {rb / 1e6:.1f} MB write {wb / 1e6:.1f} MB per launch")
# the products-shape propagate kernel on its own: config 4's mean aggregation (E = 61.9 M edges without self loops, D = 100) — the bench
# command launches csr_rows_kernel for the split rows' pre-pass and for the event-timed propagate, whose counters average to nothing useful
sg = [v for k, v in doc.get("sage", {}).items() if k.startswith("csr_rows_kernel")]
if sg:
    e = dict(sg[0])
    e["note"] = "propagate(copy_xj, mean) of SAGEConv on the products shape (tools/small_configs.py sage): E = 61 859 140, D = 100, no self loops"
    prod["csr_rows_kernel"] = e
doc["_round"], doc["_commit"] = tag, commit
json.dump(doc, open(path, "w"), indent=1)
