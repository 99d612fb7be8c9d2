#!/usr/bin/env python
"""Run HERE (the container with .git) after `gpurun -- bash tools/collect_profiles.sh <tag>` has merged its output:
copies the rocprofv3 summaries into profiles/rNN_* and stamps every file's provenance in profiles/rNN_manifest.json
and inside the PMC JSONs — commit (git rev-parse HEAD), sha256 of the kernel sources (bench.py refuses a PMC file whose
hash differs from the tree it runs in: no git on the GPU box), the bench's ms_per_step of the profiled run, and the
sum of kernel time per step.  usage: tools/publish_profiles.py <tag> <round number>"""
import csv, glob, hashlib, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def source_sha():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "models-for-relational-multimodal-data_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()


def stats(path, steps):
    rows = list(csv.DictReader(open(path)))
    return {"kernel_ms_per_step": sum(int(r["TotalDurationNs"]) for r in rows) / steps / 1e6,
            "launches_per_step": sum(int(r["Calls"]) for r in rows) / steps,
            "top": [(r["Name"].split("(")[0].replace("void ", "")[:60], round(int(r["TotalDurationNs"]) / steps / 1e6, 3)) for r in rows[:8]]}


def bench_line(log):
    for line in reversed(open(log, errors="replace").read().splitlines()):
        if line.startswith("{") and '"ms_per_step"' in line:
            return json.loads(line)
    return None


def main():
    tag, rnd = sys.argv[1], int(sys.argv[2])
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    pre = f"r{rnd:02d}"
    commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], text=True).strip()
    dirty = bool(subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "models-for-relational-multimodal-data_amd", "bench.py"], text=True).strip())
    man = {"commit": commit, "tree_dirty_at_publish": dirty, "kernel_source_sha256": source_sha(), "collected_by": "tools/collect_profiles.sh " + tag,
           "files": {}}
    names = {"main": (f"{pre}_bench_kernel_stats.csv", 6), "arxiv": (f"{pre}_arxiv_kernel_stats.csv", 6),
             "wide": (f"{pre}_wide64_kernel_stats.csv", 6), "graph": (f"{pre}_graph_replay_b200_kernel_stats.csv", None)}
    for leg, (name, steps) in names.items():
        f = glob.glob(os.path.join(src, leg, "**", "p_kernel_stats.csv"), recursive=True)
        if not f:
            continue
        shutil.copy(f[0], os.path.join(dst, name))
        entry = {}
        b = bench_line(os.path.join(src, leg + ".log"))
        if b:
            entry["bench_ms_per_step_of_the_profiled_run"] = b.get("ms_per_step")
        if steps:
            entry.update(stats(f[0], steps))
        man["files"][name] = entry
    b = bench_line(os.path.join(src, "main.log")) or {}
    wl = {"batch_size": 8192, "E": None, "N": None, "F": 128, "dtype": "bf16"}
    cfg = b.get("config", {})
    wl["E"], wl["N"] = cfg.get("edges_per_step"), cfg.get("nodes_per_step")
    stamp = {"commit": commit, "kernel_source_sha256": man["kernel_source_sha256"], "steps_profiled": 6,
             "bench_ms_per_step_of_the_profiled_run": (bench_line(os.path.join(src, "fetch.log")) or {}).get("ms_per_step")}
    if os.path.isdir(os.path.join(src, "fetch")) and os.path.isdir(os.path.join(src, "write")):
        out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), os.path.join(src, "fetch"),
                                       os.path.join(src, "write"), json.dumps(wl)], text=True)
        d = json.loads(out)
        d.update(stamp)
        d["how"] = ("two separate rocprofv3 passes (`rocprofv3 --kernel-trace --pmc FETCH_SIZE ...` and `... --pmc WRITE_SIZE ...`, each "
                    "`-- python3 bench.py --steps 5 --warmup 1 --no-extras` = 6 train steps; tools/collect_profiles.sh); " + d["how"].split("; ", 1)[1])
        json.dump(d, open(os.path.join(dst, f"{pre}_pmc_hbm_traffic.json"), "w"), indent=1)
        man["files"][f"{pre}_pmc_hbm_traffic.json"] = {"hbm_GB_per_step": sum(k["hbm_bytes_per_launch"] * k["launches"] for k in d["kernels"].values()) / 6 / 1e9}
    if os.path.isdir(os.path.join(src, "mfma")):
        out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "mfma_summary.py"), os.path.join(src, "mfma")], text=True)
        d = json.loads(out)
        d.update({"commit": commit, "kernel_source_sha256": man["kernel_source_sha256"]})
        json.dump(d, open(os.path.join(dst, f"{pre}_pmc_mfma_util.json"), "w"), indent=1)
        man["files"][f"{pre}_pmc_mfma_util.json"] = {}
    json.dump(man, open(os.path.join(dst, f"{pre}_manifest.json"), "w"), indent=1)
    print(json.dumps(man, indent=1)[:3000])


if __name__ == "__main__":
    main()
