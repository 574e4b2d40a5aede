"""GPU: correctness + timing of s2t_gemm_f32 against torch (hipBLASLt) on the C3 layer shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from speech2text_amd import _native as N

dev = torch.device("cuda")
L = N.lib()


def gemm(mode, A, B, C, M, Nn, K, bias=None, resid=None, act_src=None, act_kind=0, pro_a=0, pro_b=0,
         colsum=None, accumulate=0):
    rc = L.s2t_gemm_f32(mode, N.fp(A), A.stride(0), N.fp(B), B.stride(0), N.fp(C), C.stride(0), M, Nn, K,
                        N.fp(bias), N.fp(resid), 0 if resid is None else resid.stride(0),
                        N.fp(act_src), 0 if act_src is None else act_src.stride(0), act_kind, pro_a,
                        pro_b, N.fp(colsum), accumulate, N.stream())
    assert rc == 0, rc


def timeit(fn, it=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1000.0


def swoosh(x, kind):
    off, c = (4.0, 0.035) if kind == 1 else (1.0, 0.313261687)
    return torch.logaddexp(torch.zeros((), device=x.device), x - off) - 0.08 * x - c


def swd(x, kind):
    off = 4.0 if kind == 1 else 1.0
    return torch.sigmoid(x - off) - 0.08


def main():
    torch.manual_seed(0)
    # (rows, K, N) of the C3 layer projections: stack 0 (D=192), stacks 1/5 (D=256), stack 3
    shapes = [(31680, 192, 384), (31680, 192, 512), (31680, 192, 640), (31680, 512, 192),
              (31680, 192, 272), (31680, 192, 432), (31680, 144, 192), (31680, 192, 48),
              (31680, 48, 192), (15872, 256, 576), (15872, 256, 768), (15872, 256, 960),
              (15872, 768, 256), (15872, 960, 256), (15872, 256, 512), (15872, 256, 272),
              (7936, 256, 768), (7936, 768, 256), (3968, 256, 768), (3968, 768, 256),
              (12672, 500, 256), (1001, 132, 68)]
    print(f"{'M':>6} {'K':>5} {'N':>5} | {'NT us':>8} {'torch':>8} {'TF/s':>6} | {'NN us':>8} {'torch':>8} | {'TN us':>8} {'torch':>8} {'TF/s':>6} | maxerr")
    for (M, K, Nn) in shapes:
        x = torch.randn(M, K, device=dev)
        W = torch.randn(Nn, K, device=dev) * 0.1
        b = torch.randn(Nn, device=dev)
        g = torch.randn(M, Nn, device=dev)
        res = torch.randn(M, Nn, device=dev)
        # NT fwd with bias + residual + swoosh prologue
        y = torch.empty(M, Nn, device=dev)
        gemm(0, x, W, y, M, Nn, K, bias=b)
        ref = F.linear(x, W, b)
        e1 = (y - ref).abs().max().item() / ref.abs().max().item()
        gemm(0, x, W, y, M, Nn, K, bias=b, resid=res, pro_a=1)
        ref2 = F.linear(swoosh(x, 1), W, b) + res
        e2 = (y - ref2).abs().max().item() / ref2.abs().max().item()
        t_nt = timeit(lambda: gemm(0, x, W, y, M, Nn, K, bias=b))
        t_nt_t = timeit(lambda: F.linear(x, W, b))
        # NN dgrad with act deriv + residual
        dx = torch.empty(M, K, device=dev)
        gemm(1, g, W, dx, M, K, Nn)
        refd = g @ W
        e3 = (dx - refd).abs().max().item() / refd.abs().max().item()
        r2 = torch.randn(M, K, device=dev)
        gemm(1, g, W, dx, M, K, Nn, act_src=x, act_kind=2, resid=r2)
        refd2 = (g @ W) * swd(x, 2) + r2
        e4 = (dx - refd2).abs().max().item() / refd2.abs().max().item()
        t_nn = timeit(lambda: gemm(1, g, W, dx, M, K, Nn))
        t_nn_t = timeit(lambda: torch.mm(g, W))
        # TN wgrad with bias grad + swoosh prologue on x
        dW = torch.zeros(Nn, K, device=dev)
        db = torch.zeros(Nn, device=dev)
        gemm(2, g, x, dW, Nn, K, M, colsum=db)
        refw = g.t() @ x
        e5 = (dW - refw).abs().max().item() / refw.abs().max().item()
        e6 = (db - g.sum(0)).abs().max().item() / g.sum(0).abs().max().item()
        dW.zero_()
        gemm(2, g, x, dW, Nn, K, M, pro_b=2)
        refw2 = g.t() @ swoosh(x, 2)
        e7 = (dW - refw2).abs().max().item() / refw2.abs().max().item()
        t_tn = timeit(lambda: gemm(2, g, x, dW, Nn, K, M, colsum=db))
        t_tn_t = timeit(lambda: torch.mm(g.t(), x))
        fl = 2.0 * M * K * Nn
        print(f"{M:6d} {K:5d} {Nn:5d} | {t_nt:8.1f} {t_nt_t:8.1f} {fl / t_nt / 1e6:6.1f} | {t_nn:8.1f} {t_nn_t:8.1f} | "
              f"{t_tn:8.1f} {t_tn_t:8.1f} {fl / t_tn / 1e6:6.1f} | {max(e1, e2, e3, e4, e5, e6, e7):.1e}", flush=True)


if __name__ == "__main__":
    main()
