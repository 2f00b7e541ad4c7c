"""dev: one-line digest of a bench.py JSON line read from stdin"""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
pk = d["roofline"].get("per_kernel", [])
print(tag, d["value"], d["step_ms"]["each"], [(e.get("kernel", e.get("name")), e.get("avg_us")) for e in pk] if isinstance(pk, list) else pk)
