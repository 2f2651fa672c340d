#!/usr/bin/env python3
"""Developer tool: picks keys out of JSON lines (the tools' output) -- python tools/fmt.py key1 key2 ... < lines; a key may be
a.b for nested values; values are cut to 90 characters.  Lines may carry an "ab.sh"-style "name: " prefix."""
import json
import sys

for ln in sys.stdin:
    k = ln.find("{")
    if k < 0:
        continue
    try:
        d = json.loads(ln[k:])
    except ValueError:
        continue
    out = [ln[:k].strip()] if ln[:k].strip() else []
    for key in sys.argv[1:]:
        v = d
        for part in key.split("."):
            v = v.get(part) if isinstance(v, dict) else None
        out.append(str(v)[:90])
    print(" | ".join(out), flush=True)
