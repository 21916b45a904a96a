#!/usr/bin/env python3
"""Print the `tiny` / `h8` / `mid` legs of the last bench line in gpurun_out/b.json (helper for the GPU-box one-liners)."""
import json, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.loads(open(os.path.join(R, "gpurun_out", "b.json")).read().strip().splitlines()[-1])
print("tiny", d.get("tiny"))
for k, v in d.get("mid", {}).items():
    print("mid", k, {a: round(b, 4) for a, b in v.items() if "ms" in a})
for k, v in d.get("h8", {}).items():
    print("h8", k[:20], {a: round(b, 4) for a, b in v.items() if "ms" in a})
print("configs[1] train", d.get("train", {}).get("configs[1] shape", {}).get("ms_per_step"), "secondary_error", d.get("secondary_error"))
