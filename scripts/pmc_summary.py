"""Summarise rocprofv3 --pmc CSV output: mean counter value per launch for kernels matching a pattern.

    python scripts/pmc_summary.py DIR [DIR ...] [--kernel REGEX]
"""
import argparse, csv, glob, os, re, collections

ap = argparse.ArgumentParser()
ap.add_argument("dirs", nargs="+")
ap.add_argument("--kernel", default="scan_kernel|scan_two_rows_kernel|scan_noadj_kernel")
args = ap.parse_args()
pat = re.compile(args.kernel)
for d in args.dirs:
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")
                if not pat.search(k):
                    continue
                short = re.sub(r"\(.*", "", k)[:60]
                a = acc[(short, row["Counter_Name"])]
                a[0] += float(row["Counter_Value"]); a[1] += 1
    print("==", d)
    for (k, c), (s, n) in sorted(acc.items()):
        print("%-62s %-28s %14.4g  (%d launches)" % (k, c, s / n, n))
