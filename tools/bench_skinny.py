"""GPU: skinny GEMMs of the frontend's im2col convolutions -- s2t_gemm_f32 (NT / NN) vs hipBLASLt."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tools.bench_gemm import dev, gemm, timeit  # noqa: E402

# (rows, K, N): conv1 (1->8), conv2 (8->32), conv3 (32->128), encoder_embed.out
for (M, K, Nn) in [(64 * 996 * 80, 12, 8), (64 * 497 * 39, 72, 32), (64 * 495 * 19, 288, 128), (31680, 2432, 192)]:
    x = torch.randn(M, K, device=dev)
    W = torch.randn(Nn, K, device=dev) * 0.1
    b = torch.randn(Nn, device=dev)
    g = torch.randn(M, Nn, device=dev)
    y = torch.empty(M, Nn, device=dev)
    dx = torch.empty(M, K, device=dev)
    t_nt = timeit(lambda: gemm(0, x, W, y, M, Nn, K, bias=b))
    ref = torch.addmm(b, x, W.t())
    err = ((y - ref).abs().max() / ref.abs().max()).item()
    t_nt_t = timeit(lambda: torch.addmm(b, x, W.t()))
    t_nn = timeit(lambda: gemm(1, g, W, dx, M, K, Nn))
    t_nn_t = timeit(lambda: torch.mm(g, W))
    print(f"M={M} K={K} N={Nn}: NT ours {t_nt:7.1f} us lib {t_nt_t:7.1f} | NN ours {t_nn:7.1f} lib {t_nn_t:7.1f} | err {err:.1e}",
          flush=True)
