#!/usr/bin/env python
"""Per-kernel register / LDS / scratch / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage remarks.

    python tools/resources.py > profiles/r03_resources.md

Compiles every kernel source of ecrad_amd/csrc with the flags of its Makefile (object code is discarded) and prints one
row per kernel instantiation: the compiler's own numbers, not a claim.  Occupancy is waves per SIMD as the compiler
reports it; dynamic LDS (the level records) is set at launch and is not in the static figure."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ecrad_amd", "csrc")
flags, extra, mkvars = None, {}, {"ARCH": "gfx950"}


def expand(text):      # $(NAME) of the Makefile's own variables
    return re.sub(r"\$\((\w+)\)", lambda m: mkvars.get(m.group(1), ""), text)


for line in open(os.path.join(CSRC, "Makefile")):
    m = re.match(r"^(\w+)\s*\??=\s*(.*)$", line)
    if not m:
        continue
    name, value = m.group(1), expand(m.group(2).strip())
    mkvars.setdefault(name, value) if name == "ARCH" else mkvars.__setitem__(name, value)
    if name == "CXXFLAGS":
        flags = value.split()
    if name.startswith("EXTRA_"):       # per-file flags
        extra[name[len("EXTRA_"):] + ".hip"] = value.split()
only = set(sys.argv[1:])      # (optional: restrict to these kernel_*.hip files)
srcs = sorted(f for f in os.listdir(CSRC) if f.startswith("kernel_") and f.endswith(".hip") and (not only or f in only))
print("# Kernel resource usage (hipcc -Rpass-analysis=kernel-resource-usage, gfx950)\n")
print("Flags: `" + " ".join(flags) + "`.  One row per kernel instantiation; `occ` = waves per SIMD the register budget admits,")
print("`scratch` = bytes of private (spill) memory per lane, `LDS` = static bytes per block (the level records are dynamic LDS on top).\n")
for src in srcs:
    p = subprocess.run(["/opt/rocm/bin/hipcc", *flags, *extra.get(src, []), "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                       cwd=CSRC, capture_output=True, text=True)
    rows, cur = [], None
    for ln in p.stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs Spill|VGPRs Spill): (\S+)", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k.replace(" ", "").split("[")[0]] = v
    print(f"## {src}" + (f" (+ `{' '.join(extra[src])}`)" if src in extra else "") + "\n\n| kernel | VGPRs | AGPRs | SGPRs | SGPR spills | VGPR spills | scratch | occ | LDS |\n|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(ecrad::.*\)$", "", name).replace("void ecrad::", "").replace("ecrad::", "")
        print(f"| `{name}` | {r.get('VGPRs')} | {r.get('AGPRs')} | {r.get('TotalSGPRs')} | {r.get('SGPRsSpill')} | {r.get('VGPRsSpill')} | {r.get('ScratchSize')} | {r.get('Occupancy')} | {r.get('LDSSize')} |")
    print()
