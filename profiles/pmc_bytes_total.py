"""HBM bytes per kernel and per step from the two PMC passes (FETCH_SIZE alone; WRITE_SIZE [+ TCC hit / miss]) of one process:

    python3 profiles/pmc_bytes_total.py <fetch pass csv> <write pass csv> <steps in the process> [algorithmic bytes per step]

FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE doubled per the gfx950 correction (MI355X_MICROARCH.md)."""
import collections
import csv
import sys


def load(path, counter, scale):
    tot, n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("adn::", "")[:70]
            tot[k] += scale * float(r["Counter_Value"]); n[k] += 1
    return tot, n


fa, fb, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
fetch, nf = load(fa, "FETCH_SIZE", 2.0 * 1024.0)
write, nw = load(fb, "WRITE_SIZE", 1024.0)
print("%-72s %8s %12s %12s" % ("kernel", "launches", "read MB/step", "write MB/step"))
for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch[k] + write[k])):
    print("%-72s %8d %12.1f %12.1f" % (k, nf[k] // max(steps, 1), fetch[k] / steps / 1e6, write[k] / steps / 1e6))
total = (sum(fetch.values()) + sum(write.values())) / steps
print("total HBM traffic per step: %.1f MB (read %.1f, write %.1f)" % (total / 1e6, sum(fetch.values()) / steps / 1e6, sum(write.values()) / steps / 1e6))
if len(sys.argv) > 4:
    alg = float(sys.argv[4])
    print("algorithmic bytes per step: %.1f MB -> traffic ratio %.2f" % (alg / 1e6, total / alg))
