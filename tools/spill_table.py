"""dev: register / scratch figures of the FAST sweep kernels AND where the scratch accesses sit.
Compiles vm_sweep_kernels.hip device-only to assembly, reads the kernel descriptors (VGPRs, SGPRs, spilled VGPRs,
scratch bytes per lane) and walks the loop structure of each kernel: a loop is an EVALUATION loop when it holds a
v_rcp_f32 (one per SSIM term) and no smaller loop with one -- the bodies of the gradient and golden-section loops, i.e.
the dependent chain a phase waits for.  Scratch instructions are counted inside those loops and in the whole kernel.
usage: python tools/spill_table.py [exact] > profiles/rNN_spill_table.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exact = len(sys.argv) > 1 and sys.argv[1] == "exact"
flags = ["-DVM_EXACT=1", "-ffp-contract=off"] if exact else ["-DVM_EXACT=0", "-ffp-contract=off"]
src = os.path.join(ROOT, "videomorphing_amd", "csrc", "vm_sweep_kernels.hip")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--offload-device-only",
                           "-I" + os.path.join(ROOT, "include")] + flags + ["-S", src, "-o", out], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")


def demangle(n):
    try:
        return subprocess.check_output(["c++filt", n], text=True).strip()
    except Exception:
        return n


# kernel bodies: from "<name>:" to ".amdhsa_kernel <name>"
starts = {}
for i, l in enumerate(lines):
    m = re.match(r"^(_ZN[\w]+):", l)
    if m and m.group(1) not in starts:
        starts[m.group(1)] = i
rows = []
for i, l in enumerate(lines):
    m = re.match(r"^\s*\.amdhsa_kernel\s+(\S+)", l)
    if not m:
        continue
    name = m.group(1)
    body = lines[starts[name]:i]
    desc = {}
    for l2 in lines[i:i + 80]:
        m2 = re.match(r"^\s*\.amdhsa_(\w+)\s+(\S+)", l2)
        if m2:
            desc[m2.group(1)] = m2.group(2)
        if ".end_amdhsa_kernel" in l2:
            break
    meta = {}
    for j in range(i, min(i + 4000, len(lines))):
        pass
    labels = {}
    for k, l2 in enumerate(body):
        m2 = re.match(r"^(\.LBB\d+_\d+):", l2)
        if m2:
            labels[m2.group(1)] = k
    loops = []
    for k, l2 in enumerate(body):
        m2 = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l2)
        if m2 and m2.group(1) in labels and labels[m2.group(1)] < k:
            loops.append((labels[m2.group(1)], k))
    cnt = lambda a, b, pat: sum(1 for l2 in body[a:b + 1] if re.search(pat, l2))
    has_rcp = [(a, b) for a, b in loops if cnt(a, b, r"v_rcp_f32") > 0]
    ev = [(a, b) for a, b in has_rcp if not any((a2, b2) != (a, b) and a <= a2 and b2 <= b for a2, b2 in has_rcp)]
    ev_scr = sum(cnt(a, b, r"scratch_(load|store)") for a, b in ev)
    ev_valu = sum(cnt(a, b, r"^\s+v_") for a, b in ev)
    rows.append((demangle(name).replace("(anonymous namespace)::", "").split("(")[0], desc, cnt(0, len(body) - 1, r"scratch_load"),
                 cnt(0, len(body) - 1, r"scratch_store"), len(ev), ev_scr, ev_valu, cnt(0, len(body) - 1, r"v_(read|write)lane")))
# spilled VGPRs / scratch bytes come from the notes of the code object: take them from the resource-usage remarks
rem = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--offload-device-only", "-I" + os.path.join(ROOT, "include")] + flags +
                     ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.devnull], capture_output=True, text=True).stderr
usage, cur = {}, None
for l in rem.split("\n"):
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = demangle(m.group(1)).replace("(anonymous namespace)::", "").split("(")[0]
        usage[cur] = {}
    m = re.search(r"remark:\s+([A-Za-z][A-Za-z /\[\]]*?): (\d+) \[", l)
    if m and cur:
        usage[cur][m.group(1).strip()] = int(m.group(2))
print("%-44s %5s %5s %7s %8s %9s %9s %10s %14s %12s" % ("kernel (%s build)" % ("EXACT" if exact else "FAST"), "VGPR", "SGPR", "spilled", "scratch", "scr.loads", "scr.stores",
                                                  "eval loops", "scratch in them", "SGPR spills"))
for name, desc, nl, ns, nev, evs, evv, lanes in rows:
    u = usage.get(name, {})
    print("%-44s %5s %5s %7s %8s %9d %9d %10d %14d %12s" % (name[:44], u.get("VGPRs", "?"), u.get("TotalSGPRs", "?"), u.get("VGPRs Spill", "?"),
                                                      str(u.get("ScratchSize [bytes/lane]", "?")) + " B", nl, ns, nev, evs, u.get("SGPRs Spill", "?")))
