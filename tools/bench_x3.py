"""GPU: bf16x3-split GEMM (s2t_gemm_x3_nt) vs hipBLASLt fp32 (zk.lt_matmul) and the exact-f32 MFMA
kernel: time and error against fp64 on the C3 / C2 layer shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech2text_amd import _native as N
from speech2text_amd import zip_kernels as zk
from tools.bench_gemm import timeit, gemm

dev = torch.device("cuda")
L = N.lib()


def planes_of(W):
    n = W.numel()
    pl = torch.empty(3 * n, dtype=torch.int16, device=dev)
    N.check(L.s2t_split_planes(N.fp(W), n, N.raw(pl), n, N.stream()), "split")
    return pl


def x3(A, pl, Nn, K, C, bias=None, resid=None):
    M = A.shape[0]
    N.check(L.s2t_gemm_x3_nt(N.fp(A), A.stride(0), N.raw(pl), K, Nn * K, N.fp(C), C.stride(0), M, Nn, K,
                             N.fp(bias), N.fp(resid), 0 if resid is None else resid.stride(0), 1.0,
                             N.stream()), "x3")


def frag_of(W, transposed=False):
    """W (N,K) [or the (K,N) storage of its transpose] -> fragment-major bf16 pieces."""
    Nn, K = (W.shape[1], W.shape[0]) if transposed else W.shape
    pl = torch.empty(3 * Nn * K, dtype=torch.int16, device=dev)
    N.check(L.s2t_split_planes_frag(N.fp(W), W.stride(0), Nn, K, int(transposed), N.raw(pl), N.stream()),
            "split_frag")
    return pl


def x3f(A, pf, Nn, K, C, bias=None, resid=None, tnw=0):
    M = A.shape[0]
    N.check(L.s2t_gemm_x3f_nt(N.fp(A), A.stride(0), N.raw(pf), N.fp(C), C.stride(0), M, Nn, K,
                              N.fp(bias), N.fp(resid), 0 if resid is None else resid.stride(0), 1.0,
                              tnw, N.stream()), "x3f")


def main_f():
    torch.manual_seed(0)
    shapes = [(31744, 192, 384), (31744, 192, 512), (31744, 192, 640), (31744, 512, 192), (31744, 192, 272),
              (15872, 256, 576), (15872, 256, 768), (15872, 256, 960), (15872, 768, 256), (15872, 960, 256),
              (7936, 256, 768), (7936, 768, 256), (3968, 256, 768), (3968, 768, 256), (7936, 256, 512)]
    print(f"{'M':>7} {'K':>5} {'N':>5} | lt us  TF/s | x3f auto us TF/s | tnw=1..4 us | err x3f / lt (vs fp64)")
    for (M, K, Nn) in shapes:
        if Nn % 32 or K % 16:
            continue
        x = torch.randn(M, K, device=dev)
        W = torch.randn(Nn, K, device=dev) * 0.1
        b = torch.randn(Nn, device=dev)
        res = torch.randn(M, Nn, device=dev)
        pf = frag_of(W)
        y = torch.empty(M, Nn, device=dev)
        x3f(x, pf, Nn, K, y, b, res)
        rows = slice(max(0, M - 4096), M)
        ref = torch.nn.functional.linear(x[rows].double(), W.double(), b.double()) + res[rows].double()
        e1 = ((y[rows].double() - ref).abs().max() / ref.abs().max()).item()
        yl = zk.lt_matmul(0, x, W, b, res)
        e2 = ((yl[rows].double() - ref).abs().max() / ref.abs().max()).item()
        t2 = timeit(lambda: zk.lt_matmul(0, x, W, b, res))
        t1 = timeit(lambda: x3f(x, pf, Nn, K, y, b, res))
        ts = [timeit(lambda: x3f(x, pf, Nn, K, y, b, res, t)) for t in (1, 2, 3, 4)]
        fl = 2.0 * M * K * Nn
        print(f"{M:7d} {K:5d} {Nn:5d} | {t2:6.1f} {fl / t2 / 1e6:5.0f} | {t1:6.1f} {fl / t1 / 1e6:5.0f} | "
              + " ".join(f"{t:6.1f}" for t in ts) + f" | {e1:.2e} {e2:.2e}", flush=True)


def main():
    torch.manual_seed(0)
    shapes = [(31680, 192, 384), (31680, 192, 512), (31680, 512, 192), (31680, 192, 272), (15872, 256, 576),
              (15872, 256, 768), (15872, 256, 960), (15872, 768, 256), (15872, 960, 256), (7936, 256, 768),
              (7936, 768, 256), (3968, 256, 768), (7936, 256, 2048), (7936, 2048, 256), (7936, 256, 512),
              (601920, 128, 384), (15872, 144, 192), (1001, 136, 72)]
    print(f"{'M':>7} {'K':>5} {'N':>5} | {'x3 us':>8} {'TF/s':>6} | {'lt us':>8} {'TF/s':>6} | {'f32mfma':>8} | err x3 / lt / f32mfma (vs fp64, rel to max)")
    for (M, K, Nn) in shapes:
        x = torch.randn(M, K, device=dev)
        W = torch.randn(Nn, K, device=dev) * 0.1
        b = torch.randn(Nn, device=dev)
        pl = planes_of(W)
        y = torch.empty(M, Nn, device=dev)
        x3(x, pl, Nn, K, y, b)
        rows = slice(0, min(M, 4096))
        ref = torch.nn.functional.linear(x[rows].double(), W.double(), b.double())
        e1 = ((y[rows].double() - ref).abs().max() / ref.abs().max()).item()
        yl = zk.lt_matmul(0, x, W, b)
        e2 = ((yl[rows].double() - ref).abs().max() / ref.abs().max()).item()
        y3 = torch.empty(M, Nn, device=dev)
        gemm(0, x, W, y3, M, Nn, K, bias=b)
        e3 = ((y3[rows].double() - ref).abs().max() / ref.abs().max()).item()
        t1 = timeit(lambda: x3(x, pl, Nn, K, y, b))
        t2 = timeit(lambda: zk.lt_matmul(0, x, W, b))
        res = torch.randn(M, Nn, device=dev)
        t1r = timeit(lambda: x3(x, pl, Nn, K, y, b, res))
        t2r = timeit(lambda: zk.lt_matmul(0, x, W, b, res))
        t3 = timeit(lambda: gemm(0, x, W, y3, M, Nn, K, bias=b))
        fl = 2.0 * M * K * Nn
        print(f"{M:7d} {K:5d} {Nn:5d} | {t1:8.1f} {fl / t1 / 1e6:6.1f} | {t2:8.1f} {fl / t2 / 1e6:6.1f} | {t3:8.1f} | "
              f"{e1:.2e} {e2:.2e} {e3:.2e} | +resid x3 {t1r:.1f} lt {t2r:.1f}", flush=True)


if __name__ == "__main__":
    main_f() if "f" in sys.argv[1:] else main()
