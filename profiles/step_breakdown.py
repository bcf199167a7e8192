"""Kernel-by-kernel account of ONE training step of the bench workload from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace -d <dir> -o sb --output-format csv -- python3 profiles/gemm_breakdown.py run
    python3 profiles/step_breakdown.py <dir>/.../sb_kernel_trace.csv

Steps are delimited by adam_kernel (one per step); the second step is reported: per kernel name launches, total and
average time, plus the idle time between consecutive kernels (launch gaps)."""
import collections
import csv
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)
    m = re.match(r"_ZN3adn16gemm_bf16_kernelILi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)E(\w+?)EEvNS", name)
    if m:
        lay = {"10": "NN", "11": "NT", "00": "TN"}[m.group(4) + m.group(5)]
        return "gemm_bf16<%sx%s,%s,%s>" % (m.group(1), m.group(2), lay, "bf16" if "DF16b" in m.group(6) else "f32")
    return name.replace("adn::", "").replace("(anonymous namespace)::", "").replace("void ", "")[:60]


def main():
    rows = [r for r in csv.DictReader(open(sys.argv[1]))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    assert len(ends) >= 2, "need two optimiser steps in the trace"
    step = rows[ends[0] + 1: ends[1] + 1]
    agg = collections.OrderedDict()
    busy = 0.0
    for r in step:
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg.setdefault(short(r["Kernel_Name"]), [0, 0.0])
        a[0] += 1; a[1] += us; busy += us
    span = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3
    print("%-44s %6s %10s %9s %7s" % ("kernel", "calls", "total us", "avg us", "share"))
    for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-44s %6d %10.1f %9.1f %6.1f%%" % (k, n, us, us / n, 100 * us / span))
    print("step span %.1f us, kernels busy %.1f us, gaps %.1f us (%d launches)" % (span, busy, span - busy, len(step)))


if __name__ == "__main__":
    main()
