"""Host side of csrc/textside.hip: the text side of a fusion block with frozen, composed projections as two autograd nodes --
``text_prep`` (LayerNorm of the text tokens and the composed projections, written in the layouts the image side's GEMMs and
the bi-softmax kernel read) and ``text_out`` (the text output's projection, layer scale, stochastic depth and residual) --
six launches per block forward + backward instead of ~41 ATen kernels of ~3 us each on the step's critical path
(transformer.BiAttentionBlock.forward decides; reference fuse_modules.py:99-305)."""
import torch

from . import _lib


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _check(rc, name):
    if rc != 0:
        raise RuntimeError("%s failed: hipError %d" % (name, rc))


def _scratch(dev, B, T, H, Dv, Dl):
    """The partial products of a K split (consumed inside the call that writes them: allocated per call, stream-ordered)."""
    return torch.empty(_lib.load().zira_text_side_scratch_floats(B, T, H, Dv, Dl), device=dev, dtype=torch.float32)


def _f32c(*ts):
    return all(t is None or (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()) for t in ts)


class _TextPrep(torch.autograd.Function):
    """(l_ln [B, T, Dl], a [B, Dv, H T], c [B, H T], z [B, H T, Dv]) from the un-normalised text tokens; W1 = [AC | Z]."""

    @staticmethod
    def forward(ctx, l_in, ln_w, ln_b, eps, W1, b1, W1T, H, Dv):
        B, T, Dl = l_in.shape
        l_in = l_in.contiguous()
        dev = l_in.device
        l_ln = torch.empty_like(l_in)
        a = torch.empty((B, Dv, H * T), device=dev, dtype=torch.float32)
        c = torch.empty((B, H * T), device=dev, dtype=torch.float32)
        z = torch.empty((B, H * T, Dv), device=dev, dtype=torch.float32)
        stats = torch.empty((B * T, 2), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _check(_lib.load().zira_text_prep_fwd_f32(l_in.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), float(eps), W1.data_ptr(),
                                                      b1.data_ptr(), B, T, H, Dv, Dl, l_ln.data_ptr(), a.data_ptr(), c.data_ptr(),
                                                      z.data_ptr(), stats.data_ptr(), _stream(l_in)), "zira_text_prep_fwd_f32")
        ctx.save_for_backward(l_in, ln_w, stats, W1T)
        ctx.dims = (B, T, H, Dv, Dl)
        return l_ln, a, c, z

    @staticmethod
    def backward(ctx, g_ln, g_a, g_c, g_z):
        if not ctx.needs_input_grad[0]:
            return (None,) * 9
        l_in, ln_w, stats, W1T = ctx.saved_tensors
        B, T, H, Dv, Dl = ctx.dims
        gs = [None if g is None else g.contiguous() for g in (g_a, g_c, g_z, g_ln)]
        g_in = torch.empty_like(l_in)
        scratch = _scratch(l_in.device, B, T, H, Dv, Dl)
        ptr = lambda t: 0 if t is None else t.data_ptr()
        with torch.cuda.device(l_in.device):
            _check(_lib.load().zira_text_prep_bwd_f32(ptr(gs[0]), ptr(gs[1]), ptr(gs[2]), ptr(gs[3]), l_in.data_ptr(), ln_w.data_ptr(),
                                                      stats.data_ptr(), W1T.data_ptr(), B, T, H, Dv, Dl, scratch.data_ptr(),
                                                      g_in.data_ptr(), _stream(l_in)), "zira_text_prep_bwd_f32")
        return g_in, None, None, None, None, None, None, None, None


class _TextOut(torch.autograd.Function):
    """l_ln + gamma * keep * (o0 + (u / colsum) O) -> [B, T, Dl]."""

    @staticmethod
    def forward(ctx, u, colsum, l_ln, O, OT, o0, gamma, keep, H):
        B, T, Dl = l_ln.shape
        Dv = u.shape[-1]
        u, colsum, l_ln = u.contiguous(), colsum.contiguous(), l_ln.contiguous()
        out = torch.empty_like(l_ln)
        scratch = _scratch(u.device, B, T, H, Dv, Dl)
        with torch.cuda.device(u.device):
            _check(_lib.load().zira_text_out_fwd_f32(u.data_ptr(), colsum.data_ptr(), l_ln.data_ptr(), O.data_ptr(), o0.data_ptr(),
                                                     gamma.data_ptr(), 0 if keep is None else keep.data_ptr(), B, T, H, Dv, Dl,
                                                     scratch.data_ptr(), out.data_ptr(), _stream(u)), "zira_text_out_fwd_f32")
        ctx.save_for_backward(u, colsum, OT, gamma, keep)
        ctx.dims = (B, T, H, Dv, Dl)
        return out

    @staticmethod
    def backward(ctx, g):
        u, colsum, OT, gamma, keep = ctx.saved_tensors
        B, T, H, Dv, Dl = ctx.dims
        g = g.contiguous()
        g_u = g_cs = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            g_u, g_cs = torch.empty_like(u), torch.empty_like(colsum)
            with torch.cuda.device(u.device):
                _check(_lib.load().zira_text_out_bwd_f32(g.data_ptr(), u.data_ptr(), colsum.data_ptr(), OT.data_ptr(), gamma.data_ptr(),
                                                         0 if keep is None else keep.data_ptr(), B, T, H, Dv, Dl, g_u.data_ptr(),
                                                         g_cs.data_ptr(), _stream(u)), "zira_text_out_bwd_f32")
        return g_u, g_cs, (g if ctx.needs_input_grad[2] else None), None, None, None, None, None, None


def supported(l_in, ln, gamma_l, composed) -> bool:
    """Frozen fp32 GPU case: the LayerNorm's affine parameters and the layer scale take no gradient."""
    return (composed is not None and len(composed) >= 10 and l_in.dim() == 3 and l_in.shape[-1] <= 256
            and _f32c(l_in.contiguous(), ln.weight, ln.bias, gamma_l)
            and ln.weight is not None and ln.bias is not None and not ln.weight.requires_grad and not ln.bias.requires_grad
            and not gamma_l.requires_grad and not torch.is_autocast_enabled("cuda"))


def text_prep(l_in, ln, W1, b1, W1T, H, Dv):
    return _TextPrep.apply(l_in, ln.weight, ln.bias, ln.eps, W1, b1, W1T, H, Dv)


def text_out(u, colsum, l_ln, O, OT, o0, gamma, keep, H):
    return _TextOut.apply(u, colsum, l_ln, O, OT, o0, gamma, keep, H)
