#!/usr/bin/env python3
"""Print the stage times of a bench.py JSON line read from stdin: tools/stage_ms.py < bench.json"""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("value", round(d["value"]), d["unit"], " ms_per_step", round(d["ms_per_step"], 2))
for k, v in d.get("next_rows", {}).items():
    if isinstance(v, dict) and "ms" in v:
        print("  %-26s %8.2f ms" % (k, v["ms"]))
