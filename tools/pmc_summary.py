"""Summarise rocprofv3 output for profiles/: per-kernel averages of a --kernel-trace --stats run and
the HBM traffic of the sweep kernels from two --pmc passes (FETCH_SIZE, WRITE_SIZE).

  python tools/pmc_summary.py <fetch_dir> <write_dir> <out_csv> <out_json> "<command profiled>" [kernel,substrings]
  python tools/pmc_summary.py --merge <traffic.json> <entry.json> <config> <pairs_per_launch>
      (profiles/traffic_latest.json holds one entry per workload shape: bench.py only quotes PMC
      bytes taken at the same config and pairs per launch; the entry is stamped with the fingerprint of the
      sweep kernels' sources, and bench.py reports "traffic_stale": true when they have changed since)
  python tools/pmc_summary.py --stamp <traffic.json>    (stamp entries that have no fingerprint with the current one:
      only right while the sources are what the entries were profiled with)
"""
import csv, glob, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the sources the sweep kernels are compiled from: an entry of traffic_latest.json carries their fingerprint, and bench.py
# marks the PMC / SQ figures it quotes from an entry as stale when the sources have changed since (bench.sweep_source_hash)
SWEEP_SOURCES = ("vm_sweep_kernels.hip", "vm_morph_common.h", "vm_internal.h")


def sweep_source_hash():
    h = hashlib.sha256()
    for f in SWEEP_SOURCES:
        h.update(open(os.path.join(ROOT, "videomorphing_amd", "csrc", f), "rb").read())
    return h.hexdigest()


SWEEP = ("k_optimize", "k_step", "k_decide", "k_commit", "k_pass", "k_sparse")


def counters(d, name):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != name:
                continue
            k = r["Kernel_Name"].split("(")[1].split(")")[-1] if False else r["Kernel_Name"]
            k = k.replace("(anonymous namespace)::", "").split("(")[0]
            e = out.setdefault(k, [0, 0.0])
            e[0] += 1
            e[1] += float(r["Counter_Value"])
    return out


def merge():
    path, entry, config, pairs = sys.argv[2], json.load(open(sys.argv[3])), int(sys.argv[4]), int(sys.argv[5])
    try:
        cur = json.load(open(path))
    except Exception:
        cur = {}
    ents = [e for e in cur.get("entries", []) if not (e.get("config") == config and e.get("pairs_per_launch") == pairs)]
    ents.append({"config": config, "pairs_per_launch": pairs, "per_kernel": entry["per_kernel"],
                 "per_kernel_launches": entry.get("per_kernel_launches", {}), "source": entry["source"],
                 "sweep_source_sha256": sweep_source_hash()})
    ents.sort(key=lambda e: (e["config"], e["pairs_per_launch"]))
    json.dump({"entries": ents, "correction": entry["correction"]}, open(path, "w"), indent=1)


def merge_sq():
    """adds the per-kernel SQ / GRBM counters of a tools/pmc_table.py table to the entry of a workload shape"""
    path, table, config, pairs = sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    cur = json.load(open(path))
    lines = open(table).read().splitlines()
    cols = lines[0].split(",")
    sq = {}
    for ln in lines[1:]:
        parts = ln.rsplit(",", len(cols) - 1)          # kernel names may hold commas
        name = parts[0].strip('"')
        sq[name] = {c.replace("_per_launch", ""): float(v) for c, v in zip(cols[1:], parts[1:]) if v != ""}
    for e in cur["entries"]:
        if e.get("config") == config and e.get("pairs_per_launch") == pairs:
            e["sq_per_kernel"] = sq
            e["sq_source"] = "rocprofv3 --pmc SQ_* / GRBM_GUI_ACTIVE (separate passes, --kernel-trace only), per launch; table: " + os.path.basename(table)
    json.dump(cur, open(path, "w"), indent=1)


def main():
    global SWEEP
    if sys.argv[1] == "--merge":
        return merge()
    if sys.argv[1] == "--merge-sq":
        return merge_sq()
    if sys.argv[1] == "--stamp":
        cur = json.load(open(sys.argv[2]))
        for e in cur["entries"]:
            e.setdefault("sweep_source_sha256", sweep_source_hash())
        json.dump(cur, open(sys.argv[2], "w"), indent=1)
        return
    fd, wd, out_csv, out_json, cmd = sys.argv[1:6]
    if len(sys.argv) > 6:
        SWEEP = tuple(sys.argv[6].split(","))
    F, W = counters(fd, "FETCH_SIZE"), counters(wd, "WRITE_SIZE")
    rows = []
    tot_l = tot_f = tot_w = 0
    for k in sorted(set(F) | set(W)):
        nf, f = F.get(k, [0, 0.0])
        nw, w = W.get(k, [0, 0.0])
        n = max(nf, nw)
        rows.append((k, n, f / max(nf, 1), w / max(nw, 1)))
        if any(s in k for s in SWEEP):
            tot_l += n
            tot_f += f
            tot_w += w
    with open(out_csv, "w") as fo:
        fo.write("kernel,launches,FETCH_SIZE_KB_per_launch_raw,WRITE_SIZE_KB_per_launch_raw,HBM_bytes_per_launch_corrected\n")
        for k, n, f, w in rows:
            fo.write("\"%s\",%d,%.3f,%.3f,%.0f\n" % (k, n, f, w, (2 * f + w) * 1024))
    fk, wk = tot_f / max(tot_l, 1), tot_w / max(tot_l, 1)
    per_kernel, per_kernel_launches = {}, {}
    for k, n, f, w in rows:
        name = k.replace("void ", "").strip()
        per_kernel[name] = (2 * f + w) * 1024
        per_kernel_launches[name] = n
    json.dump({
        "per_kernel": per_kernel,
        "per_kernel_launches": per_kernel_launches,
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of `%s`; kernels summed below (%s), %d launches" % (cmd, ", ".join(SWEEP), tot_l),
        "fetch_size_kb_per_launch_raw": fk, "write_size_kb_per_launch_raw": wk,
        "hbm_bytes_per_launch": (2 * fk + wk) * 1024,
        "correction": "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section; calibrated there for 16 B/lane streams; these kernels read 4-16 B/lane, so the read side is an upper estimate); WRITE_SIZE as reported",
    }, open(out_json, "w"), indent=1)
    print(open(out_json).read())


if __name__ == "__main__":
    main()
