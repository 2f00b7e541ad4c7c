"""dev: one-line digest of a bench.py JSON line: tools/dev_bench_line.py [tag] < line.json   or   tools/dev_bench_line.py tag line.json
(a file argument is read instead of stdin: a bare call inside `gpurun` would sit on an empty stdin until the time limit)"""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
text = open(sys.argv[2]).read() if len(sys.argv) > 2 else sys.stdin.read()
d = json.loads(text.strip().splitlines()[-1])
pk = d["roofline"].get("per_kernel", [])
print(tag, d["value"], d["step_ms"]["each"], [(e.get("kernel", e.get("name")), e.get("avg_us")) for e in pk] if isinstance(pk, list) else pk)
