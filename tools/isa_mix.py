#!/usr/bin/env python
"""Instruction mix of the kernels in a gfx950 assembly listing (hipcc -S --cuda-device-only): per kernel the static
counts of VALU / packed VALU / MFMA / LDS / VMEM / SALU instructions and the most common opcodes.  The fused encoder
kernels are straight-line per tile, so static counts ~ per-tile dynamic counts."""
import collections, re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else "k_"
parts = re.split(r"\n(_Z\w+):[^\n]*\n", txt)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split("s_endpgm")[0]
    if pat not in name:
        continue
    c = collections.Counter()
    for line in body.split("\n"):
        line = line.strip()
        m = re.match(r"([a-z_0-9]+)\s", line + " ")
        if m and not line.endswith(":") and line[0] not in ".;":
            c[m.group(1)] += 1
    grp = lambda f: sum(v for k, v in c.items() if f(k))
    print(name[:70])
    print("  total", sum(c.values()), "valu", grp(lambda k: k.startswith("v_") and not k.startswith("v_mfma")),
          "pk", grp(lambda k: k.startswith("v_pk")), "mfma", grp(lambda k: k.startswith("v_mfma")),
          "lds", grp(lambda k: k.startswith("ds_")), "vmem", grp(lambda k: k.startswith(("global_", "buffer_", "scratch_"))),
          "salu", grp(lambda k: k.startswith("s_")))
    print("  ", " ".join(f"{k}:{v}" for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30)))
