"""GPU: the pre-split-weight bf16x3 GEMM (s2t_gemm_x3p) against s2t_linear_lt (hipBLASLt / the
round-3 kernel, whichever its plan cache picks) on the C3 layer shapes: error against fp64, time
per tile choice, fused epilogues.  usage: python tools/bench_x3p.py [quick]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from speech2text_amd import flat
from speech2text_amd import zip_kernels as zk
from tools.bench_gemm import timeit, swoosh, swd

dev = torch.device("cuda")

# (M, N, K) of mode-0 products y[M,N] = x[M,K] W[N,K]^T at C3 (B = 64 x 10 s), see DESIGN.md
FWD = [(31680, 272, 192), (31680, 432, 192), (31680, 192, 144), (31680, 48, 192), (31680, 192, 48),
       (31680, 384, 192), (31680, 192, 384), (31680, 512, 192), (31680, 192, 512), (31680, 640, 192),
       (31680, 192, 640), (31680, 192, 192),
       (15872, 272, 256), (15872, 576, 256), (15872, 256, 576), (15872, 256, 192), (15872, 512, 256),
       (15872, 256, 256), (15872, 768, 256), (15872, 256, 768), (15872, 960, 256), (15872, 256, 960),
       (7936, 576, 256), (7936, 256, 576), (7936, 768, 256), (7936, 256, 768), (7936, 960, 256),
       (7936, 256, 960), (7936, 512, 256), (7936, 256, 256),
       (3968, 544, 256), (3968, 576, 256), (3968, 768, 256), (3968, 256, 768), (3968, 960, 256),
       (3968, 256, 960), (3968, 512, 256), (3968, 256, 256)]


# the arithmetic is the library's (S2T_GEMM_ARITH, read per call); the default candidate list follows it
ARITH = zk.gemm_arith()
CFGS = [int(c) for c in os.environ.get(
    "X3P_CFGS", "0,222,321,312,411,2022,2021,2012,2011" if ARITH == 3 else
    "0,222,321,312,411,322,2022,2021,2012,2011,2222,2221,2212,2211").split(",")]


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    torch.manual_seed(0)
    shapes = FWD[::4] if quick else FWD
    ws = {}
    for (_, Nn, K) in shapes:
        if (Nn, K) not in ws:
            ws[(Nn, K)] = (torch.nn.Parameter(torch.randn(Nn, K, device=dev) * 0.1),
                           torch.nn.Parameter(torch.randn(Nn, device=dev)))
    plist = [p for wb in ws.values() for p in wb]
    store = flat.FlatStore(plist)
    print(f"{'mode':>4} {'M':>6} {'N':>5} {'K':>5} | {'lt us':>7} {'TF':>5} | x3p: auto  t22    t21    t12    t11  | best TF  speedup | err x3p / lt")
    tot_lt = tot_x = tot_auto = 0.0
    print("arithmetic:", zk.gemm_arith_name(), "x3p configs (100 wgs/CU + tile; 2000 + (200) + tile: LDS-DMA form (32-deep)):", CFGS)
    for mode in (0, 1):
        for (M, Nn, K) in shapes:
            W, b = ws[(Nn, K)]
            cols, inner = (Nn, K) if mode == 0 else (K, Nn)
            x = torch.randn(M, inner, device=dev)
            res = torch.randn(M, cols, device=dev)
            bias = b if mode == 0 else None
            y = zk.x3p_matmul(mode, x, W, bias, res)
            if y is None:
                print(mode, M, Nn, K, "not served")
                continue
            rows = slice(max(0, M - 2048), M)
            Wd = W.detach().double()
            ref = (x[rows].double() @ (Wd.t() if mode == 0 else Wd)) + res[rows].double()
            if bias is not None:
                ref = ref + bias.detach().double()
            e1 = ((y[rows].double() - ref).abs().max() / ref.abs().max()).item()
            yl = zk._lt_matmul_lib(mode, x, W.detach(), bias, res)
            e2 = ((yl[rows].double() - ref).abs().max() / ref.abs().max()).item()
            t_lt = timeit(lambda: zk._lt_matmul_lib(mode, x, W.detach(), bias, res))
            ts = [timeit(lambda: zk.x3p_matmul(mode, x, W, bias, res, tile=t), it=15) for t in CFGS]
            fl = 2.0 * M * Nn * K
            best = min(ts)
            tot_lt += t_lt
            tot_x += best
            tot_auto += ts[0]
            print(f"{mode:4d} {M:6d} {Nn:5d} {K:5d} | {t_lt:7.1f} {fl / t_lt / 1e6:5.0f} | "
                  + " ".join(f"{t:5.1f}" for t in ts) + f" | {CFGS[ts.index(best)]:4d} {fl / best / 1e6:5.0f}  {t_lt / best:5.2f}x | {e1:.1e} {e2:.1e}",
                  flush=True)
    print(f"sum lt {tot_lt:.0f} us, sum best x3p {tot_x:.0f} us ({tot_lt / tot_x:.2f}x), auto {tot_auto:.0f} us")

    # fused epilogues: dgrad through Swoosh, forward with the kept activation as second output
    M, Nn, K = 15872, 768, 256
    W, b = ws[(Nn, K)] if (Nn, K) in ws else list(ws.values())[0]
    Nn, K = W.shape
    x = torch.randn(M, K, device=dev)
    h, a = zk.x3p_matmul(0, x, W, b, None, act2="swoosh_l")
    href = F.linear(x.double(), W.detach().double(), b.detach().double())
    print("fwd+act2: h err", ((h.double() - href).abs().max() / href.abs().max()).item(),
          "a err", ((a.double() - swoosh(href, 1)).abs().max()).item())
    t_f = timeit(lambda: zk.x3p_matmul(0, x, W, b, None, act2="swoosh_l"))
    t_p = timeit(lambda: zk.swoosh_forward(zk.x3p_matmul(0, x, W, b, None), True))
    print(f"fwd + swoosh: fused {t_f:.1f} us, GEMM + pass {t_p:.1f} us")
    g = torch.randn(M, Nn, device=dev)          # gradient w.r.t. the activation's input side: dx = (g W) * act'(hk)
    hk = torch.randn(M, K, device=dev) * 3
    d = zk.x3p_matmul(1, g, W, None, None, act_src=hk, act_kind="swoosh_l")
    dref = (g.double() @ W.detach().double()) * swd(hk.double(), 1)
    print("dgrad*act': err", ((d.double() - dref).abs().max() / dref.abs().max()).item())
    t_f = timeit(lambda: zk.x3p_matmul(1, g, W, None, None, act_src=hk, act_kind="swoosh_l"))
    t_p = timeit(lambda: zk.swoosh_backward(hk, zk.x3p_matmul(1, g, W, None, None), True))
    print(f"dgrad * swoosh': fused {t_f:.1f} us, GEMM + pass {t_p:.1f} us")
    from speech2text_amd import planes
    t_s = timeit(lambda: store.arena.refresh())
    print(f"split of {store.numel} weights ({planes.SPLITS[0]} launches so far): {t_s:.1f} us")


if __name__ == "__main__":
    main()
