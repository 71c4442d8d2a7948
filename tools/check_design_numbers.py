#!/usr/bin/env python3
"""profiles/FIGURES.md holds the table `figure | quoted | file | key` of DESIGN.md section 8: every quoted headline figure with the
committed profile it comes from (inside DESIGN.md itself through round 5).  This script reads that table and compares: a quoted value passes when it equals the file's value rounded to the
quoted number of significant digits (or lies within 1.5 % of it).  Keys: `a.b.c` walks a JSON object; for .jsonl files
`k1=v1,k2=v2:field` picks the record whose fields print as given.     python tools/check_design_numbers.py   (exit code 1 on a mismatch)"""
import json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
text = open(os.path.join(ROOT, "profiles", "FIGURES.md")).read()
start = text.index("**Figures checked by `tools/check_design_numbers.py`**")
rows = re.findall(r"^\| ([^|]+) \| ([^|]+) \| `([^`]+)` \| `([^`]+)` \|$", text[start:], flags=re.M)


def lookup(path, key):
    full = os.path.join(ROOT, path)
    if path.endswith(".jsonl"):
        sel, field = key.rsplit(":", 1)
        want = [kv.split("=", 1) for kv in re.split(r",(?=[A-Za-z_]+=)", sel)]
        for line in open(full):
            line = line.strip()
            if not line.startswith("{"):
                continue
            rec = json.loads(line)
            if all(str(rec.get(k)) == v for k, v in want):
                return rec[field]
        raise KeyError(f"{path}: no record with {sel}")
    obj = json.loads([ln for ln in open(full).read().splitlines() if ln.startswith("{")][-1])
    for part in key.split("."):
        obj = obj[part]
    return obj


design = open(os.path.join(ROOT, "DESIGN.md")).read()
bad = 0
for figure, quoted, path, key in rows:
    # (the figure must be what DESIGN.md's text says: the quoted string -- or, for a time kept in seconds, its milliseconds -- occurs there)
    qs = quoted.strip()
    alt = f"{float(qs) * 1e3:g}" if key.endswith(":median") else qs
    if qs not in design and alt not in design:
        print(f"NOT IN DESIGN.md  {figure.strip()}: {qs}")
        bad += 1
    q = float(quoted.replace("·10", "e").strip())
    try:
        v = float(lookup(path, key))
    except Exception as e:  # noqa: BLE001
        print(f"MISSING  {figure}: {path} {key}: {e}")
        bad += 1
        continue
    digits = len(re.sub(r"[^0-9]", "", quoted.split("e")[0].lstrip("0.").strip() or "0")) or 1
    ok = abs(v - q) <= 0.015 * abs(v) or float(f"{v:.{max(digits, 1)}g}") == q
    print(f"{'ok      ' if ok else 'MISMATCH'} {figure.strip()}: quoted {quoted.strip()}, {path} says {v}")
    bad += not ok
print(f"{len(rows)} figures, {bad} not reproduced")
sys.exit(1 if bad or not rows else 0)
