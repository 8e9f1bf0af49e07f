"""The multilayer-branch variant of ZiRa's side branches (reference
models/GroundingDINO/groundingdino_dual_zero_rep_multilayer_branch.py:62-226, registry name
``dualzerorepmultilayerbranchgroundingdino``).  An ablation beside the main model
(``groundingdino_dual_zero_rep_branch.py``, rsb.py here); it differs in four ways:

* the zero-interference loss is plain L1-to-zero (not SmoothL1) and ``scaling`` starts at 1.0;
* the vision branch ends in its own GroupNorm whose affine parameters start at 1e-8 (``freeze_gn``, trained --
  its name contains "adapter" -- and NOT touched by ``__rep__``), and is added to the output of the whole
  ``input_proj`` level (conv + GroupNorm) instead of in front of that GroupNorm;
* the language branch is unconditional, named ``rep_language_adapter`` / ``loss_language_adapter``;
* ``RepZeroTransformerLayer`` -- a self-attention + FFN twin with free 1e-8 FFN weights beside the frozen ones --
  is defined (the reference keeps it commented out at its only call site, :324).

Same module names, parameter names and ``forward -> (output, loss)`` / ``__rep__`` contracts as the reference.
The variant is off the benchmarked path: it runs on PyTorch-ROCm ops (GEMM / conv libraries) on the GPU.
"""
import torch
from torch import Tensor, nn

zero_value = 1e-8


def _l1_to_zero(x: Tensor) -> Tensor:
    return x.abs().mean()       # nn.L1Loss(reduction="mean")(x, zeros_like(x))


class ZeroGroupNorm(nn.GroupNorm):
    """GroupNorm whose affine parameters start at 1e-8 (reference :62-67)."""

    def reset_parameters(self) -> None:
        if self.affine:
            nn.init.constant_(self.weight, zero_value)
            nn.init.constant_(self.bias, zero_value)


class _Merge:
    """``__rep__`` shared by the conv and linear branches: twin += scaling * branch; branch <- 1e-8; scaling <- 1."""

    def _merge(self, twin):
        with torch.no_grad():
            twin.weight.data = self.weight.data * self.scaling + twin.weight.data
            twin.bias.data = self.bias.data * self.scaling + twin.bias.data
            self.scaling = nn.Parameter(torch.ones(1).to(self.weight.data) * 1.0)
            nn.init.constant_(self.weight, zero_value)
            if self.bias is not None:
                nn.init.constant_(self.bias, zero_value)


class RepZeroConv2dGN(nn.Conv2d, _Merge):
    """Vision branch (reference :69-113): GroupNorm(scaling * conv(x) + twin(x)), loss = L1(branch) + L1(output)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 padding_mode="zeros", device=None, dtype=None):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         padding_mode, device, dtype)
        self.scaling = nn.Parameter(torch.ones(1) * 1.0)
        nn.init.constant_(self.weight, zero_value)
        if self.bias is not None:
            nn.init.constant_(self.bias, zero_value)
        self.freeze_conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                                     padding_mode, device, dtype)
        self.freeze_gn = ZeroGroupNorm(32, out_channels)
        nn.init.constant_(self.freeze_conv.weight, 0.0)
        if self.bias is not None:
            nn.init.constant_(self.freeze_conv.bias, 0.0)

    def forward(self, input: Tensor):
        if not self.training:
            return self.freeze_conv(input), input.new_zeros(1)
        branch = super().forward(input) * self.scaling
        output = self.freeze_gn(branch + self.freeze_conv(input))
        return output, _l1_to_zero(branch) + _l1_to_zero(output)

    def __rep__(self):
        self._merge(self.freeze_conv)


class RepZeroLinear(nn.Linear, _Merge):
    """Language branch of the variant (reference :116-146): as rsb.RepZeroLinear with L1 and scaling 1.0."""

    def __init__(self, in_features, out_features, bias=True, device=None, dtype=None):
        super().__init__(in_features, out_features, bias, device, dtype)
        self.scaling = nn.Parameter(torch.ones(1) * 1.0)
        nn.init.constant_(self.weight, zero_value)
        self.freeze_linear = nn.Linear(in_features, out_features, bias, device, dtype)
        nn.init.constant_(self.freeze_linear.weight, 0.0)
        if self.bias is not None:
            nn.init.constant_(self.freeze_linear.bias, 0.0)

    def forward(self, input: Tensor):
        if not self.training:
            return self.freeze_linear(input), input.new_zeros(1)
        branch = self.scaling * super().forward(input)
        output = branch + self.freeze_linear(input)
        return output, _l1_to_zero(branch) + _l1_to_zero(output)

    def __rep__(self):
        self._merge(self.freeze_linear)


class RepZeroTransformerLayer(nn.Module):
    """Self-attention + FFN with a free (1e-8) FFN beside the frozen one (reference :149-226).  Input / output are
    sequence-first ``[T, B, C]`` as ``nn.MultiheadAttention`` takes them."""

    def __init__(self, embed_dim, nhead=8, down_dim=2048, ffn_drop=0, activation=None, output_dim=None, **kwargs):
        super().__init__()
        output_dim = embed_dim if output_dim is None else output_dim
        self.freeze_self_attn = nn.MultiheadAttention(embed_dim, nhead, dropout=ffn_drop)
        self.freeze_linear1 = nn.Linear(embed_dim, down_dim)
        self.dropout = nn.Dropout(ffn_drop)
        self.freeze_linear2 = nn.Linear(down_dim, output_dim)
        self.freeze_norm1 = nn.LayerNorm(embed_dim)
        self.freeze_norm2 = nn.LayerNorm(output_dim)
        self.dropout1 = nn.Dropout(ffn_drop)
        self.dropout2 = nn.Dropout(ffn_drop)
        self.nhead = nhead
        self.activation = nn.ReLU(inplace=True) if activation is None else activation
        nn.init.zeros_(self.freeze_linear2.weight)
        nn.init.zeros_(self.freeze_linear2.bias)
        self.free_linear1 = nn.Linear(embed_dim, down_dim)
        self.free_linear2 = nn.Linear(down_dim, output_dim)
        for lin in (self.free_linear1, self.free_linear2):
            nn.init.constant_(lin.weight, zero_value)
            nn.init.constant_(lin.bias, zero_value)

    def __rep__(self):
        with torch.no_grad():
            for twin, free in ((self.freeze_linear1, self.free_linear1), (self.freeze_linear2, self.free_linear2)):
                twin.weight.data = free.weight.data + twin.weight.data
                twin.bias.data = free.bias.data + twin.bias.data
                nn.init.constant_(free.weight, zero_value)
                nn.init.constant_(free.bias, zero_value)

    def forward(self, src: Tensor, pos=None):
        q = k = src if pos is None else src + pos
        src = self.freeze_norm1(src + self.dropout1(self.freeze_self_attn(q, k, value=src)[0]))
        if not self.training:
            hidden = self.dropout(self.activation(self.freeze_linear1(src)))
            return self.freeze_norm2(self.dropout2(self.freeze_linear2(hidden))), src.new_zeros(1)
        branch1 = self.free_linear1(src)
        hidden = self.dropout(self.activation(self.freeze_linear1(src) + branch1))
        branch2 = self.free_linear2(hidden)
        out = self.freeze_norm2(self.dropout2(self.freeze_linear2(hidden) + branch2))
        return out, _l1_to_zero(branch1) + _l1_to_zero(branch2) + _l1_to_zero(out)
