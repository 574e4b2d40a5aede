"""Oracle: conformer encoder (Subsampling + torchaudio-style Conformer + 1x1 output conv),
functional torch-CPU restatement over a state_dict with the product's / torchaudio's names.

PARITY UNPINNED for the conformer stack: torchaudio==0.13.1 (requirements.txt:16) is absent;
the block structure restated is torchaudio.models.Conformer as called at
model/encoder/conformer.py:170-178,193.  `Subsampling` follows model/encoder/conformer.py:32-135.
Used to check the product's fused time-major path (HIP GLU+depthwise kernel) against the
plain (B,C,T) formulation."""
import torch
import torch.nn.functional as F


def subsampling4(sd, pfx, x, length):
    x = x.unsqueeze(1)
    x = F.relu(F.conv2d(x, sd[pfx + "conv.0.weight"], sd[pfx + "conv.0.bias"], stride=2))
    x = F.relu(F.conv2d(x, sd[pfx + "conv.2.weight"], sd[pfx + "conv.2.bias"], stride=2))
    b, c, t, f = x.shape
    out = F.linear(x.transpose(1, 2).reshape(b, t, c * f), sd[pfx + "linear.0.weight"],
                   sd[pfx + "linear.0.bias"])
    length = ((length - 1) // 2 - 1) // 2
    mask = torch.arange(t).unsqueeze(0) >= length.unsqueeze(1)
    return out.masked_fill(mask.unsqueeze(-1), 0.0), length


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"])


def keep_scale(seed, shape, p):
    """The product's stateless dropout mask (csrc/conf_elem.hip keep_scale, csrc/conf_attn.hip
    keep_elem): element i of the contiguous tensor is kept iff the high word of splitmix64's
    finaliser of seed + i * golden is >= p * 2^32; kept elements carry 1 / (1 - p).  torch's own
    dropout streams (Philox on the GPU) cannot be reproduced, so parity under dropout is checked
    with THIS mask on both sides and the mask's statistics separately."""
    import numpy as np
    n = 1
    for d in shape:
        n *= int(d)
    thr = int(p * 4294967296.0)
    if p > 0 and thr == 0:
        thr = 1
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    keep = (z >> np.uint64(32)) >= np.uint64(thr)
    return torch.from_numpy(keep.astype("float32") / (1.0 - p)).reshape(*shape)


class _Drop:
    """Dropout sites in the executor's order; seeds = the values the product drew (None: p = 0)."""

    def __init__(self, p, seeds):
        self.p, self.i = p, 0
        self.seeds = seeds if isinstance(seeds, str) or seeds is None else list(seeds)

    def __call__(self, x):
        if not self.p:
            return x
        if self.seeds == "torch":              # timing baseline: torch's own CPU dropout stream
            return F.dropout(x, self.p, True)
        m = keep_scale(self.seeds[self.i], x.shape, self.p).to(x.dtype)
        self.i += 1
        return x * m

    def mask(self, shape, dtype):
        if self.seeds == "torch":
            return F.dropout(torch.ones(shape, dtype=dtype), self.p, True)
        m = keep_scale(self.seeds[self.i], shape, self.p).to(dtype)
        self.i += 1
        return m


def _ffn(sd, p, x, drop):
    x = _ln(sd, p + "sequential.0.", x)
    x = drop(F.silu(F.linear(x, sd[p + "sequential.1.weight"], sd[p + "sequential.1.bias"])))
    return drop(F.linear(x, sd[p + "sequential.4.weight"], sd[p + "sequential.4.bias"]))


def _mha_dropout(sd, p, y, num_heads, kpm, drop):
    """nn.MultiheadAttention with dropout on the attention probabilities, mask injected."""
    T, B, D = y.shape
    dh = D // num_heads
    qkv = F.linear(y, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"])
    q, k, v = (qkv[..., i * D:(i + 1) * D].reshape(T, B, num_heads, dh).permute(1, 2, 0, 3)
               for i in range(3))
    s = torch.matmul(q, k.transpose(-1, -2)) / dh ** 0.5
    s = s.masked_fill(kpm.view(B, 1, 1, T), float("-inf"))
    pr = s.softmax(-1) * drop.mask((B, num_heads, T, T), s.dtype)
    o = torch.matmul(pr, v).permute(2, 0, 1, 3).reshape(T, B, D)
    return F.linear(o, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])


def _conv_module(sd, p, x, training):
    """x (T,B,D) -> via (B,D,T) exactly as torchaudio arranges it."""
    y = _ln(sd, p + "layer_norm.", x.transpose(0, 1)).transpose(1, 2)
    y = F.glu(F.conv1d(y, sd[p + "sequential.0.weight"], sd[p + "sequential.0.bias"]), dim=1)
    C = y.shape[1]
    k = sd[p + "sequential.2.weight"].shape[-1]
    y = F.conv1d(y, sd[p + "sequential.2.weight"], sd[p + "sequential.2.bias"], padding=k // 2,
                 groups=C)
    y = F.batch_norm(y, sd[p + "sequential.3.running_mean"].clone(),
                     sd[p + "sequential.3.running_var"].clone(), sd[p + "sequential.3.weight"],
                     sd[p + "sequential.3.bias"], training=training)
    y = F.silu(y)
    y = F.conv1d(y, sd[p + "sequential.5.weight"], sd[p + "sequential.5.bias"])
    return y.permute(2, 0, 1)


def conformer_forward(sd, x, lengths, num_layers, num_heads, training=False, dropout=0.0,
                      seeds=None):
    """x (B,T,80) -> (logits (B,T',V), lengths).  dropout > 0: every nn.Dropout site of the block
    (after the FFN SiLU and Linear, on the attention probabilities, after out_proj, after the conv
    module) uses the product's hashed mask for the seeds it drew, in its order."""
    drop = _Drop(dropout if training else 0.0, seeds)
    x, lengths = subsampling4(sd, "_subsampling_module.", x, lengths)
    T = x.shape[1]
    kpm = torch.arange(T).unsqueeze(0) >= lengths.unsqueeze(1)
    x = x.transpose(0, 1)
    for l in range(num_layers):
        p = f"_conformer_module.conformer_layers.{l}."
        x = _ffn(sd, p + "ffn1.", x, drop) * 0.5 + x
        res = x
        y = _ln(sd, p + "self_attn_layer_norm.", x)
        if drop.p:
            y = _mha_dropout(sd, p, y, num_heads, kpm, drop)
        else:
            y, _ = F.multi_head_attention_forward(
                y, y, y, y.shape[-1], num_heads, sd[p + "self_attn.in_proj_weight"],
                sd[p + "self_attn.in_proj_bias"], None, None, False, 0.0,
                sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"],
                training=False, key_padding_mask=kpm, need_weights=False)
        x = drop(y) + res
        x = x + drop(_conv_module(sd, p + "conv_module.", x, training))
        x = _ffn(sd, p + "ffn2.", x, drop) * 0.5 + x
        x = _ln(sd, p + "final_layer_norm.", x)
    x = x.transpose(0, 1)
    logits = F.linear(x, sd["_output_layer.weight"].squeeze(-1), sd["_output_layer.bias"])
    return logits, lengths
