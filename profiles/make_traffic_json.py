"""HBM traffic of the dominant kernel class from two rocprofv3 PMC passes (the TCC cannot hold FETCH_SIZE and
WRITE_SIZE in one pass, MI355X_MICROARCH.md) over a process that ran NOTHING but `train_steps` train steps of the headline
workload (bench.py --only-train-steps: no pre-warm, no evaluation, no B=26 sub-run -- every launch the counters saw is a
launch of the B=520 step):

    rocprofv3 --pmc FETCH_SIZE                          -d A -o a --output-format csv -- python3 bench.py --only-train-steps --steps 3 --warmup 1
    rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d B -o b --output-format csv -- python3 bench.py ... (same)
    python3 profiles/make_traffic_json.py A/.../a_counter_collection.csv B/.../b_counter_collection.csv bf16 <commit> 4 > profiles/rNN/pmc_traffic_bf16.json

FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH_SIZE is doubled per the gfx950 correction (128-byte requests
tallied at 64 B).  bench.py reads the file named pmc_traffic_<precision>.json for `roofline.traffic` (bytes per GEMM launch)
and `roofline_lstm_*.traffic` (bytes per train step of the recurrent kernels / the LSTM time steps of a train step).
The launch counts must be whole multiples of `train_steps`; the script refuses to write a file otherwise."""
import collections
import csv
import json
import sys


def per_kernel(path, match):
    tot, n = collections.Counter(), collections.Counter()
    matches = match if isinstance(match, (tuple, list)) else (match,)
    for r in csv.DictReader(open(path)):
        if any(mm in r["Kernel_Name"] for mm in matches):
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            n[r["Counter_Name"]] += 1
    return tot, n


def summarise(fa, fb, match, train_steps):
    ta, na = per_kernel(fa, match)
    tb, nb = per_kernel(fb, match)
    launches = na["FETCH_SIZE"]
    if launches != nb["WRITE_SIZE"] or launches % train_steps:
        raise SystemExit("%s: %d launches in the fetch pass, %d in the write pass, %d train steps -- the passes did not see "
                         "the same whole train steps" % (match, launches, nb["WRITE_SIZE"], train_steps))
    fetch_total = 2.0 * 1024.0 * ta["FETCH_SIZE"]
    write_total = 1024.0 * tb["WRITE_SIZE"]
    hit, miss = tb["TCC_HIT_sum"], tb["TCC_MISS_sum"]
    return {"launches": launches, "launches_per_train_step": launches // train_steps,
            "fetch_bytes_per_launch": fetch_total / max(1, launches), "write_bytes_per_launch": write_total / max(1, launches),
            "l2_hit_rate": hit / max(1.0, hit + miss),
            "traffic_bytes_per_launch": (fetch_total + write_total) / max(1, launches),
            "traffic_bytes_per_train_step": (fetch_total + write_total) / train_steps}


def main():
    fa, fb, prec = sys.argv[1:4]
    commit = sys.argv[4] if len(sys.argv) > 4 else "unknown"
    train_steps = int(sys.argv[5]) if len(sys.argv) > 5 else 4
    # (gemm_bf16_kernel and gemm_bf16_pp_kernel; over planes also the fused-plane kernel gemm_x3f_kernel)
    match = ("gemm_bf16", "gemm_x3f", "skinny_n", "skinny_tn_kernel") if prec in ("bf16", "bf16x3", "mixed") else ("gemm_f32_kernel",)
    out = {"kernel_class": "%s (all instantiations)" % " + ".join(match), "commit": "PMC passes taken at commit %s" % commit,
           "train_steps": train_steps}
    out.update(summarise(fa, fb, match, train_steps))
    # the LSTM kernels: one launch covers all T time steps of up to 3 LSTMs (bench.py divides the bytes of a train step by
    # the LSTM time steps of a train step)
    lstm = {"bf16": "lstm_%s_cluster_kernel", "bf16x3": "lstm_%s_cluster_x3_kernel", "mixed": "lstm_%s_cluster_x3_kernel"}.get(prec, "lstm_%s_step_kernel")
    for key, m in (("lstm_fwd", lstm % "fwd"), ("lstm_bwd", lstm % "bwd")):
        if prec == "mixed" and key == "lstm_bwd":
            m = "lstm_bwd_cluster_kernel"        # (the mode back-propagates through the recurrences on the bf16 mode's kernel)
        d = summarise(fa, fb, m, train_steps)
        if d["launches"]:
            d["kernel"] = m
            out[key] = d
    out["source"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (separate passes) around "
                     "`bench.py --only-train-steps --precision %s`: %d train steps at B=520, T=40 and nothing else; "
                     "FETCH_SIZE doubled per the gfx950 correction" % (prec, train_steps))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
