"""dev: digest of a config[2] bench line from stdin"""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
c = d["config"]
print(sys.argv[1] if len(sys.argv) > 1 else "", "value %.1f M pixel*iters/s  %.1f ms per step  pairs per launch %s  in flight %s" % (
    d["value"], d["ms_per_step"], c.get("pairs_per_launch"), c.get("pairs_in_flight_per_gpu")))
