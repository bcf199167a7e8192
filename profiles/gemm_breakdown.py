"""Per-launch GEMM breakdown of one training step of the bench workload.

Run under rocprofv3 with ADN_GEMM_TRACE=1 (see the header of profiles/r01/gemm_breakdown_*.txt):
    ADN_GEMM_TRACE=1 rocprofv3 --kernel-trace -d <dir> -o bd -- python3 profiles/gemm_breakdown.py run 2> trace.txt
    python3 profiles/gemm_breakdown.py join trace.txt <dir>/.../bd_kernel_trace.csv
`run` executes 1 warm-up + 1 marked training step; `join` pairs the i-th "ADN_GEMM" line with the i-th GEMM kernel
dispatch of the trace and prints time / TFLOP/s per launch, aggregated by shape.
"""
import csv
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    import bench
    from ip_avsr_amd.model import AdeNetModel
    device = torch.device("cuda", 0)
    model = AdeNetModel(bench.build_spec())
    model.set_precision(os.environ.get("ADN_PRECISION", "bf16"))
    bench.synthetic_params(model)
    xs, y, m_d, mask = bench.synthetic_batch(torch, 0, int(os.environ.get("BD_BATCH", bench.B_PER_GPU)), device)   # BD_BATCH=26: the reference's minibatch
    lens = None if os.environ.get("BD_PADDED") else mask.sum(axis=1).astype("int32")     # frame compaction, as bench.py announces it (BD_PADDED=1: off)
    if os.environ.get("BD_INPUTS") == "bench":         # the resident form bench.py hands over: bfloat16 (bf16) / hi-lo planes (bf16x3, mixed)
        prec = os.environ.get("ADN_PRECISION", "bf16")
        if prec == "bf16":
            xs = [x.to(torch.bfloat16) for x in xs]
        elif prec in ("bf16x3", "mixed"):
            from ip_avsr_amd.model import PlaneInput
            xs = [PlaneInput.split(x) for x in xs]

    def step():
        if lens is not None:
            model.set_batch_lengths(lens)
        model.train_step(xs, y, m_d, bench.THETA, bench.LR, want_loss=False)
    for _ in range(int(os.environ.get("BD_WARM", 0))):     # (small batches: past the idle-clock transient)
        step()
    torch.cuda.synchronize()
    for _ in range(2):
        sys.stderr.write("ADN_STEP\n"); sys.stderr.flush()
        step()
        torch.cuda.synchronize()


def join(trace_txt, kernel_csv):
    calls, step = [], -1
    for line in open(trace_txt):
        if line.startswith("ADN_STEP"):
            step += 1
        elif line.startswith("ADN_GEMM"):
            f = line.split()
            kv = dict(x.split("=") for x in f[2:])
            calls.append((step, f[1], {k: int(v) for k, v in kv.items()}))
    rows = [r for r in csv.DictReader(open(kernel_csv))
            if ("gemm_" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"]) or ("skinny_" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"])]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    assert len(rows) == len(calls), (len(rows), len(calls))
    agg = collections.OrderedDict()
    for (step, lay, kv), r in zip(calls, rows):
        if step != 1:
            continue
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        key = (lay, kv["M"], kv["N"], kv["K"], kv["tile"], kv["split"], kv["lean"], kv["acc"], kv.get("groups", 1), kv.get("planes", 0))
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1; a[1] += us
    tot = sum(a[1] for a in agg.values())
    print("layout      M      N      K   tile split lean acc grp pl  count   us/launch   TFLOP/s   share   (grp = problems per launch; "
          "pl = bf16x3 over hi / lo planes; K = executed k; TFLOP/s = executed, all problems of the launch; tile 1001 / 1002 / 1003 = the "
          "skinny kernels nn / nk / tn of gemm_skinny.hip)")
    for key, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lay, M, N, K, tile, split, lean, acc, grp, pl = key
        tf = 2.0 * M * N * K * grp * n / us / 1e6
        print(f"{lay:>4} {M:8d} {N:6d} {K:6d} {tile:6d} {split:5d} {lean:4d} {acc:3d} {grp:3d} {pl:2d} {n:6d} {us / n:11.1f} {tf:9.1f} {us / tot:7.1%}")
    print(f"total GEMM kernel time per step: {tot / 1e3:.3f} ms")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        join(sys.argv[2], sys.argv[3])
