"""Latency of one Zipformer2 streaming_step at the C3 (YAML) dimensions: eager vs hipGraph.
Usage: python tools/bench_stream.py [batch ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech2text_amd.model.encoder.zipformer import Zipformer2, Zipformer2Config  # noqa: E402
from speech2text_amd.model.encoder.zipformer_streaming import StreamingSession  # noqa: E402

CFG = dict(feature_dim=80, downsampling_factor=(1, 2, 4, 8, 4, 2),
           num_encoder_layers=(2, 2, 2, 2, 2, 2), feedforward_dim=(512, 768, 768, 768, 768, 768),
           encoder_dim=(192, 256, 256, 256, 256, 256),
           encoder_unmasked_dim=(192, 192, 192, 192, 192, 192),
           num_heads=(4, 4, 4, 8, 4, 4), query_head_dim=(32,), value_head_dim=(12,),
           pos_head_dim=(4,), pos_dim=48, cnn_module_kernel=(31, 31, 15, 15, 15, 31), causal=True,
           chunk_size=(32,), left_context_frames=(128,))


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = Zipformer2(Zipformer2Config(**CFG)).to(dev).eval()
    for B in [int(a) for a in sys.argv[1:]] or [1, 16]:
        T = 2 * 32 + 13
        xs = [torch.randn(B, T, 80, device=dev) for _ in range(8)]
        st = m.get_init_states(B, dev)
        outs = []
        for x in xs:                                  # warm-up + eager reference
            y, st = m.streaming_step(x, st)
            outs.append(y.clone())
        torch.cuda.synchronize()
        st = m.get_init_states(B, dev)
        t0 = time.perf_counter()
        n = 40
        for i in range(n):
            y, st = m.streaming_step(xs[i % 8], st)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / n * 1e3
        sess = StreamingSession(m, B, dev)
        err = 0.0
        for i, x in enumerate(xs):
            err = max(err, float((sess.step(x) - outs[i]).abs().max()))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            sess.step(xs[i % 8])
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / n * 1e3
        # 32 output-rate-50Hz frames = 0.64 s of audio per step
        print(f"B={B}: eager {eager:.2f} ms/step, hipGraph {graph:.2f} ms/step "
              f"(max |graph-eager| {err:.2e}); real-time factor {graph / 640.0 / 1.0:.5f} per "
              f"stream, {B * 0.64 / (graph * 1e-3):.0f} audio-s/s", flush=True)


if __name__ == "__main__":
    main()
