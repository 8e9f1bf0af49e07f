"""ZiRa's reparameterizable dual side branch (RSB) -- ``RepZeroLinear`` / ``RepZeroConv2d``.

Mirror of the reference modules (groundingdino/models/GroundingDINO/
groundingdino_dual_zero_rep_branch.py:62-135): same constructor arguments, same parameter /
state-dict names (``weight, bias, scaling, freeze_linear.* / freeze_conv.*``), same
``forward -> (output, zero_interference_loss)`` contract and same ``__rep__`` merge.

    train: branch = scaling * F(x; W, b);  out = branch + F(x; W_f, b_f)
           loss   = mean(smooth_l1(branch, 0)) + mean(smooth_l1(out, 0))
    eval : out = F(x; W_f, b_f);  loss = zeros(1)
    __rep__ (after every task): W_f += scaling*W; b_f += scaling*b; scaling <- init; W, b <- 1e-8

The two dense contractions go to the GEMM / convolution libraries (MFMA); everything after
them -- scale, add, both SmoothL1-to-zero means, and in the backward all three gradients --
is one pass in the HIP kernels of csrc/rsb.hip (C ABI ``zira_rsb_fwd_f32 / zira_rsb_bwd_f32``).
On GPU tensors that kernel path is the only one (it raises if the extension is missing); CPU
tensors evaluate the defining expression with torch ops, as the reference does.
"""
import torch
import torch.nn.functional as F
from torch import Tensor, nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from .dense import conv2d_as_gemm, conv2d_pair_as_gemm

zero_value = 1e-8
lan_scale = 0.1
vis_scale = 0.1


class _RSBEpilogue(Function):
    """(y_branch, y_twin, scaling) -> (out, loss) on the GPU through the C ABI."""

    @staticmethod
    def forward(ctx, y_branch, y_twin, scaling):
        lib = _lib.load()
        y_branch = y_branch.contiguous()
        y_twin = y_twin.contiguous()
        n = y_branch.numel()
        out = torch.empty_like(y_branch)
        loss = torch.empty(1, dtype=torch.float32, device=y_branch.device)
        ws = torch.empty(int(lib.zira_rsb_workspace_floats(n)), dtype=torch.float32,
                         device=y_branch.device)
        with torch.cuda.device(y_branch.device):
            rc = lib.zira_rsb_fwd_f32(y_branch.data_ptr(), y_twin.data_ptr(), scaling.data_ptr(), n,
                                      out.data_ptr(), loss.data_ptr(), ws.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("zira_rsb_fwd_f32 failed: hipError %d" % rc)
        ctx.save_for_backward(y_branch, y_twin, scaling)
        return out, loss

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out, grad_loss):
        y_branch, y_twin, scaling = ctx.saved_tensors
        lib = _lib.load()
        n = y_branch.numel()
        g_branch = torch.empty_like(y_branch)
        g_twin = torch.empty_like(y_twin)
        g_scaling = torch.empty(1, dtype=torch.float32, device=y_branch.device)
        ws = torch.empty(int(lib.zira_rsb_workspace_floats(n)), dtype=torch.float32,
                         device=y_branch.device)
        go = grad_out.contiguous() if grad_out is not None else None
        gl = grad_loss.contiguous() if grad_loss is not None else None
        with torch.cuda.device(y_branch.device):
            rc = lib.zira_rsb_bwd_f32(y_branch.data_ptr(), y_twin.data_ptr(), scaling.data_ptr(),
                                      go.data_ptr() if go is not None else None,
                                      gl.data_ptr() if gl is not None else None, n,
                                      g_branch.data_ptr(), g_twin.data_ptr(), g_scaling.data_ptr(),
                                      ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("zira_rsb_bwd_f32 failed: hipError %d" % rc)
        return g_branch, g_twin, g_scaling.to(scaling.dtype)


def rsb_epilogue(y_branch: Tensor, y_twin: Tensor, scaling: Tensor):
    """out = scaling*y_branch + y_twin and the zero-interference loss (0-dim, as the reference)."""
    if y_branch.is_cuda:
        dt = y_branch.dtype
        out, loss = _RSBEpilogue.apply(y_branch.float(), y_twin.float(), scaling.float())
        return out.to(dt), loss.reshape(())
    branch = scaling * y_branch
    out = branch + y_twin
    loss = (F.smooth_l1_loss(branch, torch.zeros_like(branch)) +
            F.smooth_l1_loss(out, torch.zeros_like(out)))
    return out, loss


class RepZeroConv2d(nn.Conv2d):
    """Vision side branch beside an ``input_proj`` conv (reference :66-103)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride=1, padding=0,
                 dilation=1, groups: int = 1, bias: bool = True, padding_mode: str = "zeros",
                 device=None, dtype=None, zero_value=zero_value) -> None:
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                         bias, padding_mode, device, dtype)
        self.scaling = nn.Parameter(torch.ones(1) * vis_scale)
        nn.init.constant_(self.weight, val=zero_value)
        if self.bias is not None:
            nn.init.constant_(self.bias, val=zero_value)
        self.freeze_conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding,
                                     dilation, groups, bias, padding_mode, device, dtype)
        nn.init.constant_(self.freeze_conv.weight, val=0.0)
        if self.bias is not None:
            nn.init.constant_(self.freeze_conv.bias, val=0.0)

    def _gemm_ok(self):
        return (self.dilation == (1, 1) and self.groups == 1 and self.padding_mode == "zeros"
                and not isinstance(self.padding, str))

    def forward(self, input: Tensor):
        if not self._gemm_ok():  # configurations the hot path never uses: plain conv modules
            if not self.training:
                return self.freeze_conv(input), input.new_zeros(1)
            return rsb_epilogue(super().forward(input), self.freeze_conv(input), self.scaling)
        twin = self.freeze_conv
        if not self.training:
            return conv2d_as_gemm(input, twin.weight, twin.bias, self.stride, self.padding), \
                input.new_zeros(1)
        # branch and twin read the same activations: one batched GEMM over both weight sets
        y_branch, y_twin = conv2d_pair_as_gemm(input, self.weight, self.bias, twin.weight, twin.bias,
                                               self.stride, self.padding)
        return rsb_epilogue(y_branch, y_twin, self.scaling)

    def __rep__(self):
        with torch.no_grad():
            self.freeze_conv.weight.data = self.weight.data * self.scaling + self.freeze_conv.weight.data
            self.freeze_conv.bias.data = self.bias.data * self.scaling + self.freeze_conv.bias.data
            self.scaling = nn.Parameter(torch.ones(1).to(self.weight.data) * vis_scale)
            nn.init.constant_(self.weight, val=zero_value)
            if self.bias is not None:
                nn.init.constant_(self.bias, val=zero_value)


class RepZeroLinear(nn.Linear):
    """Language side branch beside ``feat_map`` (reference :105-135).  As in the reference the
    branch *bias* keeps nn.Linear's default init at construction (only ``__rep__`` resets it)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None,
                 dtype=None) -> None:
        super().__init__(in_features, out_features, bias, device, dtype)
        self.scaling = nn.Parameter(torch.ones(1) * lan_scale)
        nn.init.constant_(self.weight, val=zero_value)
        self.freeze_linear = nn.Linear(in_features, out_features, bias, device, dtype)
        nn.init.constant_(self.freeze_linear.weight, val=0.0)
        if self.bias is not None:
            nn.init.constant_(self.freeze_linear.bias, val=0.0)

    def forward(self, input: Tensor):
        if not self.training:
            return self.freeze_linear(input), input.new_zeros(1)
        return rsb_epilogue(super().forward(input), self.freeze_linear(input), self.scaling)

    def __rep__(self):
        with torch.no_grad():
            self.freeze_linear.weight.data = self.weight.data * self.scaling + self.freeze_linear.weight.data
            self.freeze_linear.bias.data = self.bias.data * self.scaling + self.freeze_linear.bias.data
            self.scaling = nn.Parameter(torch.ones(1).to(self.weight.data) * lan_scale)
            nn.init.constant_(self.weight, val=zero_value)
            if self.bias is not None:
                nn.init.constant_(self.bias, val=zero_value)
