"""dK/dV kernel of the attention backward: the hand-placed (generated-assembly) main loop against the compiler-built
kernel, same inputs (fused entry point, pre-scaled q): bit-exactness of dk / dv / the sqk partial sums, and interleaved
timing of the whole backward (dq + dkv) at the benchmarked shape.   python tools/attn_dkv_asm_ab.py [B H T]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvit_amd import ops, _lib
from nvit_amd._lib import BF16

dev = torch.device("cuda:0")
lib = _lib.load()
d = 64
MODES = [int(x) for x in os.environ.get("MODES", "1").split(",")]   # 0: compiler-built; 1: hand-placed (2: the ping-pong probe build of commit 240ec07 only)


def make(B, H, T, seed=0):
    C, M = H * d, B * T
    g = torch.Generator(device=dev).manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)
    sqk = (1.0 / 32) * (1.0 + 0.05 * torch.tanh(rn(C)))
    se = (sqk * 32.0).reshape(1, H, 1, d)
    q = (se * torch.nn.functional.normalize(rn(B, H, T, d), dim=-1)).bfloat16()
    k = (se * torch.nn.functional.normalize(rn(B, H, T, d), dim=-1)).bfloat16()
    v = (rn(B, H, T, d) * 0.05).bfloat16()
    gt = (rn(M, C) * 1e-3).bfloat16()
    rq = 1.0 + rn(M, H).abs() * 0.1
    rk = 1.0 + rn(M, H).abs() * 0.1
    scale = math.sqrt(d)
    qpre = ops.attn_q_prescale(d)
    qs = (q.float() * qpre).bfloat16()
    o, lse = ops.attn_fwd(BF16, 1, qs, k, v, scale, sqk, 32.0, q_prescale=qpre)
    return dict(B=B, H=H, T=T, C=C, M=M, sqk=sqk, qs=qs, k=k, v=v, gt=gt, rq=rq, rk=rk, scale=scale, qpre=qpre, o=o, lse=lse)


def run(c, asm):
    lib.nvit_set_attn_dkv_asm(int(asm))
    dqkv = torch.zeros(c["M"], 3 * c["C"], device=dev, dtype=torch.bfloat16)
    pq, pk = ops.attn_bwd_qknorm(c["gt"], c["qs"], c["k"], c["v"], c["o"], c["lse"], c["scale"], c["rq"], c["rk"], c["sqk"], 32.0,
                                 dqkv, 3 * c["C"], dqkv[:, c["C"]:], dqkv[:, 2 * c["C"]:], 3 * c["C"], q_prescale=c["qpre"])
    torch.cuda.synchronize()
    return dqkv, pq, pk


def check(B, H, T):
    c = make(B, H, T, seed=B * 1000 + T)
    a, pqa, pka = run(c, 0)
    ok = True
    for mode in MODES:
        b, pqb, pkb = run(c, mode)
        nd = int((a != b).sum().item())
        npk = int((pka != pkb).sum().item())
        C = c["C"]
        dk_bad = int((a[:, C:2 * C] != b[:, C:2 * C]).sum().item())
        dv_bad = int((a[:, 2 * C:] != b[:, 2 * C:]).sum().item())
        print(f"B={B} H={H} T={T} mode {mode}: differing elements dq|dk|dv total {nd} (dk {dk_bad}, dv {dv_bad}), sqk partials {npk}; "
              f"finite {bool(torch.isfinite(b.float()).all().item())}, |dk|max {a[:, C:2 * C].float().abs().max().item():.3e}")
        ok = ok and nd == 0 and npk == 0
    return ok


def t_of(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


if __name__ == "__main__":
    ok = True
    shapes = [(2, 2, 64), (2, 3, 128), (1, 2, 200), (2, 2, 784), (3, 12, 784), (1, 1, 16), (2, 2, 49)]
    if len(sys.argv) > 3:
        shapes = [(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))]
    if os.environ.get("TIME_ONLY"):
        shapes = []
    for s in shapes:
        ok = check(*s) and ok
    print("BIT-EXACT" if ok else "MISMATCH")
    if os.environ.get("NO_TIME"):
        sys.exit(0 if ok else 1)
    c = make(128, 12, 784)
    dqkv = torch.empty(c["M"], 3 * c["C"], device=dev, dtype=torch.bfloat16)
    def bwd():
        ops.attn_bwd_qknorm(c["gt"], c["qs"], c["k"], c["v"], c["o"], c["lse"], c["scale"], c["rq"], c["rk"], c["sqk"], 32.0, dqkv,
                            3 * c["C"], dqkv[:, c["C"]:], dqkv[:, 2 * c["C"]:], 3 * c["C"], q_prescale=c["qpre"])
    modes = [0] + MODES
    res = {m: [] for m in modes}
    for rnd in range(9):
        for asm in modes:
            lib.nvit_set_attn_dkv_asm(asm)
            res[asm].append(t_of(bwd, n=12))
    names = {0: "compiler-built", 1: "hand-placed (2 x 4 waves)", 2: "hand-placed ping-pong (8 waves)"}
    for asm in modes:
        ts = sorted(res[asm])
        print(f"backward (dq + dkv), dkv {names[asm]}: median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f} us")
    sys.exit(0 if ok else 1)
