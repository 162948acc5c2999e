#!/usr/bin/env python3
"""TTFT-vs-QPS plot from per-QPS CSV outputs (<prefix>_output_<qps>.csv) or sweep JSONL files.

    plot.py --series "sticky+offload=stack" --series "round-robin=naive" [--out ttft.png]
    plot.py --jsonl "prefix caching=profiles/sweep_a.jsonl" --jsonl "recompute=profiles/sweep_b.jsonl"

Without matplotlib the table is printed instead."""
import argparse
import csv
import glob
import json
import re
import statistics


def from_csv(prefix):
    pts = []
    for fn in glob.glob(f"{prefix}_output_*.csv"):
        qps = float(re.search(r"_output_([\d.]+)\.csv$", fn).group(1))
        with open(fn) as f:
            ttft = [float(r["ttft"]) for r in csv.DictReader(f) if float(r["ttft"]) >= 0]
        if ttft:
            pts.append((qps, statistics.fmean(ttft)))
    return sorted(pts)


def from_jsonl(path):
    with open(path) as f:
        return sorted((r["qps"], r["mean_ttft_s"]) for r in map(json.loads, f) if "mean_ttft_s" in r)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--series", action="append", default=[], help="label=csv-prefix")
    ap.add_argument("--jsonl", action="append", default=[], help="label=file.jsonl")
    ap.add_argument("--out", default="ttft_vs_qps.png")
    ap.add_argument("--ymax", type=float, default=10.0)
    a = ap.parse_args()
    data = {}
    for s in a.series:
        label, prefix = s.split("=", 1)
        data[label] = from_csv(prefix)
    for s in a.jsonl:
        label, path = s.split("=", 1)
        data[label] = from_jsonl(path)
    for label, pts in data.items():
        print(label)
        for q, t in pts:
            print(f"  qps {q:5.2f}  mean TTFT {t:8.4f} s")
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except ImportError:
        print("matplotlib not installed: table only")
        return
    for label, pts in data.items():
        plt.plot([p[0] for p in pts], [p[1] for p in pts], marker="o", label=label)
    plt.xlabel("offered QPS")
    plt.ylabel("mean TTFT (s)")
    plt.ylim(bottom=0, top=a.ymax)
    plt.legend()
    plt.grid(alpha=0.3)
    plt.savefig(a.out, dpi=150)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
