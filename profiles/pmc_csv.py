"""Average PMC counter values per kernel from rocprofv3 csv output (--output-format csv, one or more passes):
    python profiles/pmc_csv.py <dir> [kernel-substring]
walks <dir> for *counter_collection.csv and prints, per kernel, every counter's average per dispatch."""
import csv
import os
import sys


def main():
    root = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = {}
    for d, _, files in os.walk(root):
        for f in files:
            if not f.endswith("counter_collection.csv"):
                continue
            with open(os.path.join(d, f)) as fh:
                for row in csv.DictReader(fh):
                    k = row.get("Kernel_Name", "")
                    if sub not in k:
                        continue
                    key = (k.split("(")[0][:60], row["Counter_Name"])
                    a = acc.setdefault(key, {})
                    did = row.get("Dispatch_Id", "0")
                    a[did] = a.get(did, 0.0) + float(row["Counter_Value"])
    for (k, c), per in sorted(acc.items()):
        vals = list(per.values())
        print("%-60s %-40s dispatches %4d  avg %16.1f" % (k, c, len(vals), sum(vals) / len(vals)))


if __name__ == "__main__":
    main()
