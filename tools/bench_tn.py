"""GPU: the TN (weight-gradient) mode of s2t_gemm_f32 on the C3 shapes; tile / slice count come
from S2T_TN_TILE / S2T_TN_BLOCKS (read once per process), so sweep them from the shell."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tools.bench_gemm import dev, gemm, timeit  # noqa: E402

if os.environ.get("TN_SHAPES") == "sym":      # Whiten statistics: x^T x
    SHAPES_ = [(31680, 192, 192), (31680, 144, 144), (31680, 128, 128), (15872, 256, 256), (15872, 192, 192),
               (7936, 256, 256), (3968, 256, 256)]
else:
    SHAPES_ = None
SHAPES = SHAPES_ or [(31680, 192, 384), (31680, 192, 640), (31680, 512, 192), (31680, 192, 272),
          (31680, 192, 48), (15872, 256, 768), (15872, 960, 256), (15872, 256, 272),
          (7936, 256, 768), (3968, 768, 256)]
tag = (f"W={os.environ.get('S2T_TN_W', '1')} tile={os.environ.get('S2T_TN_TILE', 'auto')} "
       f"blocks={os.environ.get('S2T_TN_W_BLOCKS', os.environ.get('S2T_TN_BLOCKS', 'auto'))}")
out = []
tot = 0.0
for (M, K, Nn) in SHAPES:
    x = torch.randn(M, K, device=dev)
    g = torch.randn(M, Nn, device=dev)
    dW = torch.zeros(Nn, K, device=dev)
    db = torch.zeros(Nn, device=dev)
    gemm(2, g, x, dW, Nn, K, M, colsum=db)
    ref = g.t() @ x
    err = ((dW - ref).abs().max() / ref.abs().max()).item()
    assert err < 1e-4 or os.environ.get('S2T_GEMM_DEBUG'), (M, K, Nn, err)
    t = timeit(lambda: gemm(2, g, x, dW, Nn, K, M, colsum=db))
    tot += t
    out.append(f"{t:6.1f}({2.0 * M * K * Nn / t / 1e6:5.1f})")
print(f"{tag:24s} " + " ".join(out) + f"  sum {tot:.0f} us", flush=True)
