"""Step-by-step probe of the LDS-DMA x3p tiles: small shapes first, one launch per line, flushed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speech2text_amd import flat, zip_kernels as zk

dev = torch.device("cuda")
torch.manual_seed(0)
tiles = [int(t) for t in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["2011", "2012", "2021", "2022"])]
shapes = [(64, 64, 16), (64, 64, 64), (128, 128, 64), (300, 192, 272), (4096, 256, 256), (15872, 768, 256)]
for (M, N, K) in shapes:
    W = torch.nn.Parameter(torch.randn(N, K, device=dev) * 0.1)
    b = torch.nn.Parameter(torch.randn(N, device=dev))
    store = flat.FlatStore([W, b])
    x = torch.randn(M, K, device=dev)
    ref = torch.nn.functional.linear(x.double(), W.detach().double(), b.detach().double())
    for t in tiles:
        print(f"M {M} N {N} K {K} tile {t} ...", end="", flush=True)
        y = zk.x3p_matmul(0, x, W, b, None, tile=t)
        torch.cuda.synchronize()
        if y is None:
            print(" refused", flush=True)
            continue
        e = ((y.double() - ref).abs().max() / ref.abs().max()).item()
        print(f" err {e:.2e}", flush=True)
        if not e < 1e-5:
            bad = ((y.double() - ref).abs() > 1e-4 * ref.abs().max()).nonzero()
            print("   first bad:", bad[:5].tolist(), " count", bad.shape[0], flush=True)
