"""Refresh profiles/pmc_traffic.json from a PMC summary (scripts/pmc_summary.py output holding FETCH_SIZE and WRITE_SIZE):

    python scripts/update_pmc_traffic.py CONFIG gpurun_out/prof_TAG_pmc.txt [--note "..."]

Per kernel of the summary: HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 counts 128-byte read requests as
64 bytes: MI355X_MICROARCH.md, HBM / rocprofv3 section), stored under CONFIG -> kernel family with the hash of the sweep's
sources at the time: bench.py hands the figure out only while that hash still matches (a changed kernel has unknown traffic)."""
import hashlib, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["tissue_analysis_amd/csrc/kernels_scan.hip", "tissue_analysis_amd/csrc/ta_sweep_common.h", "tissue_analysis_amd/csrc/ta_pin_tables.inc"]
FAMILIES = ("scan_wide_kernel", "scan_two_rows_kernel", "scan_noadj_kernel", "scan_kernel")


def sources_hash():
    h = hashlib.sha256()
    for f in SOURCES:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    config, path = sys.argv[1], sys.argv[2]
    note = sys.argv[sys.argv.index("--note") + 1] if "--note" in sys.argv else ""
    vals = {}
    for line in open(path):
        m = re.search(r"(scan_\w+kernel|scan_kernel)\S*.*?\b(FETCH_SIZE|WRITE_SIZE)\s+([0-9.e+]+)\s+\((\d+) launches\)", line)
        if m and int(m.group(4)) >= 3:                # (the two measuring launches of another shape are not a mean)
            fam = next(f for f in FAMILIES if m.group(1).startswith(f))
            vals.setdefault(fam, {})[m.group(2)] = float(m.group(3))
    out_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    db = json.load(open(out_path)) if os.path.exists(out_path) else {}
    if "entries" not in db:
        db = {"entries": {}, "_note": "HBM bytes per launch of the dominant kernel = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 counts 128-B read "
                                      "requests as 64 B; MI355X_MICROARCH.md, HBM section), keyed by bench config and kernel family, each with the hash of "
                                      "the sweep's sources it was measured on: bench.py reports roofline.traffic only while the hash matches"}
    try:
        commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT).decode().strip()
    except Exception:
        commit = None
    for fam, v in vals.items():
        if "FETCH_SIZE" not in v:
            continue
        b = int((2.0 * v["FETCH_SIZE"] + v.get("WRITE_SIZE", 0.0)) * 1024)
        db["entries"].setdefault(config, {})[fam] = {"bytes": b, "fetch_size": v["FETCH_SIZE"], "write_size": v.get("WRITE_SIZE"),
                                                    "sources_sha16": sources_hash(), "commit": commit, "from": os.path.basename(path), "note": note}
        print(config, fam, b)
    json.dump(db, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
