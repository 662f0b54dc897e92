#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv: per-kernel stats over the STEADY-STATE steps only.

Steps are delimited by launches of a marker kernel (default yolo_finalize_kernel: one per training step);
the last --steps of them are kept, which drops MIOpen's first-step solver search from the statistics.
Kernel names are shortened (template arguments stripped) so the table stays readable.
usage: trace_summary.py kernel_trace.csv [--steps K] [--marker NAME] [--top N]
"""
import argparse
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if name.startswith("_ZN2ck") or name.startswith("ck::"):
        m = re.search(r"(kernel_[a-z_0-9]+)", name)
        return "ck::" + (m.group(1) if m else name[:40])
    if name.startswith("Cijk_"):
        m = re.search(r"(Cijk_[A-Za-z]+_[A-Za-z]+_[A-Z]+_MT\d+x\d+x\d+)", name)
        return (m.group(1) if m else name[:40]) + " (Tensile GEMM)"
    if name.startswith("at::native::"):
        outer = re.match(r"at::native::([A-Za-z_0-9]+)", name).group(1)
        inner = re.findall(r"([A-Za-z_0-9]+_kernel(?:_cuda)?|CUDAFunctor[A-Za-z_0-9]*|[A-Za-z]+Functor[A-Za-z_0-9]*|multi_tensor_apply_kernel)", name[len("at::native::") + len(outer):])
        inner = [i for i in inner if i != outer][:2]
        return ("at::" + outer + (":" + "+".join(inner) if inner else ""))[:90]
    name = re.sub(r"<.*", "", name)                   # drop template arguments
    name = re.sub(r"\(.*$", "", name)                 # drop argument list
    return name[:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--marker", default="yolo_finalize_kernel")
    ap.add_argument("--top", type=int, default=60)
    a = ap.parse_args()
    rows = []
    with open(a.trace, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
    if len(marks) < a.steps + 1:
        sys.exit(f"only {len(marks)} marker launches; need {a.steps + 1}")
    lo, hi = marks[-a.steps - 1] + 1, marks[-1] + 1
    sel = rows[lo:hi]
    wall = sel[-1][1] - sel[0][0]
    agg = defaultdict(list)
    for s, e, n in sel:
        agg[short(n)].append(e - s)
    busy = sum(sum(v) for v in agg.values())
    print(f"# steady state: last {a.steps} steps, {len(sel)} kernel launches, wall {wall / 1e6:.3f} ms "
          f"({wall / 1e6 / a.steps:.3f} ms/step), GPU busy {busy / 1e6:.3f} ms ({100.0 * busy / wall:.1f}%)")
    print("kernel,calls,calls_per_step,total_us,avg_us,min_us,max_us,pct_of_busy")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[: a.top]:
        print(f"{k},{len(v)},{len(v) / a.steps:.1f},{sum(v) / 1e3:.1f},{sum(v) / len(v) / 1e3:.2f},"
              f"{min(v) / 1e3:.2f},{max(v) / 1e3:.2f},{100.0 * sum(v) / busy:.2f}")


if __name__ == "__main__":
    main()
