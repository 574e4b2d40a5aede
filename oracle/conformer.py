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


def _ffn(sd, p, x):
    x = _ln(sd, p + "sequential.0.", x)
    x = F.silu(F.linear(x, sd[p + "sequential.1.weight"], sd[p + "sequential.1.bias"]))
    return F.linear(x, sd[p + "sequential.4.weight"], sd[p + "sequential.4.bias"])


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


def conformer_forward(sd, x, lengths, num_layers, num_heads, training=False):
    """eval / train-without-dropout forward; x (B,T,80) -> (logits (B,T',V), lengths)."""
    x, lengths = subsampling4(sd, "_subsampling_module.", x, lengths)
    T = x.shape[1]
    kpm = torch.arange(T).unsqueeze(0) >= lengths.unsqueeze(1)
    x = x.transpose(0, 1)
    for l in range(num_layers):
        p = f"_conformer_module.conformer_layers.{l}."
        x = _ffn(sd, p + "ffn1.", x) * 0.5 + x
        res = x
        y = _ln(sd, p + "self_attn_layer_norm.", x)
        y, _ = F.multi_head_attention_forward(
            y, y, y, y.shape[-1], num_heads, sd[p + "self_attn.in_proj_weight"],
            sd[p + "self_attn.in_proj_bias"], None, None, False, 0.0,
            sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"],
            training=False, key_padding_mask=kpm, need_weights=False)
        x = y + res
        x = x + _conv_module(sd, p + "conv_module.", x, training)
        x = _ffn(sd, p + "ffn2.", x) * 0.5 + x
        x = _ln(sd, p + "final_layer_norm.", x)
    x = x.transpose(0, 1)
    logits = F.linear(x, sd["_output_layer.weight"].squeeze(-1), sd["_output_layer.bias"])
    return logits, lengths
