#!/usr/bin/env python
"""Turn the rocprofv3 outputs of scripts/profile_round.sh into the committed summaries:
   <dst>/kernel_stats.csv (the --stats table), <dst>/pmc_summary.csv (mean FETCH_SIZE / WRITE_SIZE per launch, KB, raw)
   and profiles/pmc_latest.json (what bench.py reads for roofline.traffic)."""
import glob
import json
import os
import re
import shutil
import sys

import pandas as pd

src, dst = sys.argv[1], sys.argv[2]
num_envs = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
os.makedirs(dst, exist_ok=True)


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+)(<[^(]*>)?\(", name + "(")
    base = m.group(1) if m else name
    tpl = m.group(2) or "" if m else ""
    return base + tpl if base.startswith("k_") else base[:48]


stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, "kernel_stats.csv"))
cstats = glob.glob(os.path.join(src, "contact", "**", "*kernel_stats.csv"), recursive=True)
if cstats:                                # kernel trace of the contact-rich regime (scripts/contact_regime.py ... contact-only)
    shutil.copy(cstats[0], os.path.join(dst, "contact_rich_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "contact.log")):
    shutil.copy(os.path.join(src, "contact.log"), os.path.join(dst, "contact_regime.txt"))
rows = {}
for which in ("fetch", "write"):
    f = glob.glob(os.path.join(src, which, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    d = pd.read_csv(f[0])
    d["k"] = d["Kernel_Name"].map(short)
    for k, v in d.groupby("k")["Counter_Value"].mean().items():
        rows.setdefault(k, {})[which + "_kb"] = float(v)
    for k, v in d.groupby("k")["Counter_Value"].size().items():
        rows.setdefault(k, {})["launches_" + which] = int(v)
pd.DataFrame.from_dict(rows, orient="index").sort_index().to_csv(os.path.join(dst, "pmc_summary.csv"))
latest = {}
for k, r in rows.items():
    if not k.startswith("k_"):
        continue
    base = re.sub(r"<.*", "", k)
    gated = "<true" in k
    if gated:
        continue                      # the device-gated launches are no-ops unless an env reset
    rec = latest.setdefault(base, {"num_envs": num_envs, "fetch_kb": 0.0, "write_kb": 0.0, "variants": 0})
    rec["fetch_kb"] += r.get("fetch_kb", 0.0)
    rec["write_kb"] += r.get("write_kb", 0.0)
    rec["variants"] += 1
for rec in latest.values():               # template variants of one kernel: mean over variants
    rec["fetch_kb"] /= rec["variants"]
    rec["write_kb"] /= rec["variants"]
    del rec["variants"]
# instruction-issue counters (scripts/pmc_issue.sh): mean per k_physics4<false> launch, headline regime and the SETTLED contact-rich
# regime (the last 100 launches of scripts/contact_regime.py ... contact-only: steps 121-220 after the teleport)
issue = {"num_envs": num_envs}
for tag, key, last in (("h", "headline", None), ("c", "contact_settled", 100)):
    rec = {}
    for part in ("1", "2"):
        f = glob.glob(os.path.join(src, f"issue_{tag}{part}", "**", "*counter_collection.csv"), recursive=True)
        if not f:
            continue
        d = pd.read_csv(f[0])
        d = d[d["Kernel_Name"].str.contains("k_physics4<false>", regex=False) | d["Kernel_Name"].str.contains("k_physics4<(bool)0>", regex=False)]
        for cname, g in d.groupby("Counter_Name"):
            g = g.sort_values("Dispatch_Id")
            vals = g["Counter_Value"].values
            if last:
                vals = vals[-last:]
            rec[cname] = float(vals.mean())
            rec["launches"] = int(len(vals))
    if rec:
        issue[key] = rec
if len(issue) > 1:
    latest["issue"] = issue
    with open(os.path.join(dst, "issue_counters.json"), "w") as fh:
        json.dump(issue, fh, indent=1, sort_keys=True)
json.dump(latest, open(os.path.join(os.path.dirname(os.path.abspath(dst)), "pmc_latest.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(latest, indent=1))
