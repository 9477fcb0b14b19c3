#!/usr/bin/env python3
"""Fold the rocprofv3 output of tools/profile_round.sh into the two tracked evidence files of a round:

    python tools/pmc_fold.py gpurun_out/<tag> <tag> [dst_dir = profiles/] [config = train5k] [dtype = f32]
      -> profiles/<tag>_kernel_stats.csv   (rocprofv3 --kernel-trace --stats summary, short kernel names)
      -> profiles/<tag>_pmc.json           (per kernel, averages per dispatch over every PMC pass)

Counter handling follows MI355X_MICROARCH.md (HBM / rocprofv3 sections): FETCH_SIZE and WRITE_SIZE are
collected in separate passes and are in KB; gfx950 reports HALF of a 16-byte-per-lane read stream in
FETCH_SIZE, so hbm_bytes = (2 FETCH_SIZE + WRITE_SIZE) * 1024.  Derived figures:
  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE     (share of LDS-array cycles that are conflict replays)
  mfma_util         = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)
  scratch_bytes, vgprs, lds_bytes: per-dispatch launch properties from the trace (scratch > 0 = spilled registers)
"""
import csv
import glob
import json
import os
import re
import sys

csv.field_size_limit(1 << 30)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"^void ", "", name).replace("mvh::", "")
    name = re.sub(r"\(.*$", "", name).replace(" ", "")
    return name[:100]


def fold_pmc(out_dir):
    table = {}
    for d in sorted(glob.glob(os.path.join(out_dir, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                t = table.setdefault(k, {"_n": {}, "_sum": {}, "scratch_bytes": 0, "vgprs": 0, "lds_bytes": 0,
                                         "workgroup": 0})
                c = r["Counter_Name"]
                t["_n"][c] = t["_n"].get(c, 0) + 1
                t["_sum"][c] = t["_sum"].get(c, 0.0) + float(r["Counter_Value"])
                t["scratch_bytes"] = max(t["scratch_bytes"], int(r["Scratch_Size"]))
                t["vgprs"] = max(t["vgprs"], int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"]))
                t["lds_bytes"] = max(t["lds_bytes"], int(r["LDS_Block_Size"]))
                t["workgroup"] = max(t["workgroup"], int(r["Workgroup_Size"]))
    res = {}
    for k, t in table.items():
        avg = {c: t["_sum"][c] / t["_n"][c] for c in t["_sum"]}
        e = {c: round(v, 2) for c, v in sorted(avg.items())}
        e.update(dispatches=max(t["_n"].values()), scratch_bytes_per_lane=t["scratch_bytes"], vgprs=t["vgprs"],
                 lds_bytes=t["lds_bytes"], workgroup=t["workgroup"])
        if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
            e["hbm_bytes"] = int((2 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024)
        if avg.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_frac"] = round(avg.get("SQ_LDS_BANK_CONFLICT", 0.0) / avg["SQ_LDS_IDX_ACTIVE"], 4)
        if avg.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
            e["mfma_util"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (avg["GRBM_GUI_ACTIVE"] / 8 * 256 * 4), 5)
        res[k] = e
    return res


def fold_stats(out_dir, dst):
    files = glob.glob(os.path.join(out_dir, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        return None
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -int(r["TotalDurationNs"]))
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    with open(dst, "w") as o:
        o.write("kernel,calls,total_us,avg_us,min_us,max_us,percent\n")
        for r in rows:
            o.write(f'"{short(r["Name"])}",{r["Calls"]},{int(r["TotalDurationNs"]) / 1e3:.1f},'
                    f'{float(r["AverageNs"]) / 1e3:.2f},{int(r["MinNs"]) / 1e3:.2f},{int(r["MaxNs"]) / 1e3:.2f},'
                    f'{100.0 * int(r["TotalDurationNs"]) / tot:.2f}\n')
        o.write(f'"TOTAL",,{tot / 1e3:.1f},,,,100\n')
    return rows


def source_sha():
    """Same hash as bench.source_sha(): bench.py quotes a profile only for the sources it was taken on."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "mesh-vae_amd", "csrc", "*.h*")) +
                    [os.path.join(ROOT, "include", "meshvae_hip.h")]):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def main():
    out_dir, tag = sys.argv[1], sys.argv[2]
    prof = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles")
    config = sys.argv[4] if len(sys.argv) > 4 else "train5k"
    dtype = sys.argv[5] if len(sys.argv) > 5 else "f32"
    os.makedirs(prof, exist_ok=True)
    rows = fold_stats(out_dir, os.path.join(prof, f"{tag}_kernel_stats.csv"))
    pmc = fold_pmc(out_dir)
    keep = {k: v for k, v in pmc.items() if k.startswith("k_")}
    head = ""
    try:
        import subprocess
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except OSError:
        pass
    sha_file = os.path.join(out_dir, "src_sha.txt")          # written on the GPU box: the sources the run was made on
    sha = open(sha_file).read().strip() if os.path.exists(sha_file) else source_sha()
    json.dump({"_how": __doc__.strip(), "_tag": tag, "_head": head, "_src_sha": sha, "_config": config, "_dtype": dtype,
               "kernels": keep},
              open(os.path.join(prof, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
    bj = os.path.join(out_dir, "bench_n1.json")
    if os.path.exists(bj) and os.path.getsize(bj) > 0:
        open(os.path.join(prof, f"{tag}_bench_n1.json"), "w").write(open(bj).read())
    print(f"{tag}: {len(keep)} kernels with counters, stats rows: {0 if rows is None else len(rows)}")
    for k, v in sorted(keep.items(), key=lambda kv: -kv[1].get("hbm_bytes", 0))[:12]:
        print(f"  {k:44s} hbm {v.get('hbm_bytes', 0) / 1e6:8.1f} MB  scratch/lane {v['scratch_bytes_per_lane']:5d} B  "
              f"lds-conflict {v.get('lds_conflict_frac', 0):.3f}  mfma {v.get('mfma_util', 0):.4f}")


if __name__ == "__main__":
    main()
