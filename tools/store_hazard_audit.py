#!/usr/bin/env python
"""128-bit buffer stores whose soffset is an SGPR, followed within WINDOW instructions by a write of one of their data
registers (gfx950 assembly of csrc/*.hip).  hipcc pads the store-data hazard (a >64-bit store reads its data VGPRs for a
few cycles after issue) only for stores WITHOUT a register soffset; round 5 found z2 rows of the column-transformer
forward with the third data dword of `buffer_store_dwordx4 v[44:47], v196, s[24:27], s94 offen` replaced by the result of
the `v_lshlrev_b32 v46, ...` that followed it directly (rows 25/27/29 of a tile, channels 100-125 — the "defect (a)" of
round 4, which -fno-strict-aliasing only hid by changing the schedule)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "models-for-relational-multimodal-data_amd", "csrc")
WINDOW = 3


def assembly(path):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fno-strict-aliasing", "--offload-arch=gfx950",
                               "--cuda-device-only", "-S", path, "-o", out], stderr=subprocess.DEVNULL)
        return [ln.strip() for ln in open(out)]


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def audit(lines):
    code = [ln for ln in lines if ln and not ln.startswith(";") and not re.match(r"[.\w$]+:", ln) and not ln.startswith(".")]
    wide_sgpr, hits = 0, []
    for i, ln in enumerate(code):
        m = re.match(r"buffer_store_dwordx[34] (v\[\d+:\d+\]), (\w+), s\[\d+:\d+\], (\S+)", ln)
        if not m or not re.fullmatch(r"s\d+|vcc_lo|vcc_hi|m0|ttmp\d+", m.group(3)):
            continue
        wide_sgpr += 1
        data = regs(m.group(1))
        for nxt in code[i + 1:i + 1 + WINDOW]:
            if nxt.startswith(("s_", "buffer_store", "global_store", "ds_write", ";;")):
                continue
            ops = re.split(r"[ ,]+", nxt)
            if len(ops) > 1 and regs(ops[1]) & data:          # first operand of a VALU / load = destination
                hits.append((ln, nxt))
                break
    return wide_sgpr, hits


if __name__ == "__main__":
    names = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    total = 0
    for name in names:
        n, hits = audit(assembly(os.path.join(CSRC, name)))
        total += len(hits)
        print(f"{name}: {n} wide stores with a register soffset, {len(hits)} followed by a write of their data registers")
        for a, b in hits[:6]:
            print("     ", a, "  ->  ", b)
    sys.exit(1 if total else 0)
