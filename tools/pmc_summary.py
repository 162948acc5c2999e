"""Summarise a rocprofv3 --pmc counter_collection CSV: mean counter value per kernel name.

    python tools/pmc_summary.py <dir with *_counter_collection.csv> [name substring ...]
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    filters = sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if filters and not any(s in name for s in filters):
                    continue
                acc[name.split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, cs in acc.items():
        out[k] = {c: {"mean": sum(v) / len(v), "dispatches": len(v)} for c, v in cs.items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
