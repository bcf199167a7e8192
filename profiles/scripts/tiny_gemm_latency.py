"""How long the small GEMMs of the reference minibatch (1040 rows) take when launched back to back (warm instruction cache,
warm L2) against what they take inside a train step (profiles/r04/gemm_breakdown_bf16_B26.txt), where ~30 different kernels
alternate.  python profiles/scripts/tiny_gemm_latency.py  (on an MI355X)"""
import ctypes as C, sys, time
sys.path.insert(0, "/root/repo")
import torch
from ip_avsr_amd import _lib
lib = _lib.load()
torch.cuda.set_device(0)
def bench(M, N, K, reps=300, interleave=None):
    Kp = ((K + 63) // 64) * 64
    A = torch.randn(M, Kp, device="cuda"); B = torch.randn(Kp, ((N + 63) // 64) * 64, device="cuda"); Cm = torch.empty(M, ((N + 63) // 64) * 64, device="cuda")
    bias = torch.zeros(((N + 63) // 64) * 64, device="cuda")
    A16 = A.to(torch.bfloat16); B16 = B.to(torch.bfloat16)
    s = torch.cuda.current_stream().cuda_stream
    p = lambda t: C.c_void_p(t.data_ptr())
    def call():
        _lib.check(lib.adn_op_gemm_shadow(0, M, N, K, p(A), Kp, p(B), B.shape[1], p(Cm), Cm.shape[1], p(A16), p(B16), None, 0, C.c_void_p(s)))
    for _ in range(20): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    big = torch.randn(2048, 2048, device="cuda")
    e0.record()
    for _ in range(reps):
        call()
        if interleave: interleave(big)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for shape in ((1040, 26, 250), (1040, 50, 500), (1040, 500, 1000), (1040, 2000, 1200)):
    a = bench(*shape)
    print("M N K = %s: back to back %.1f us per launch" % (shape, a), flush=True)
