"""Host side of ``zira_ffn_f16x2_f32`` (csrc/ffn_f16x2.hip): the FROZEN feed-forward block  linear2(relu(linear1(x)))  of the
encoder layer (reference transformer_for_adapter.py:877-886 under the freeze of groundingdino_dual_zero_rep_branch.py:722-745)
as ONE launch per direction on the f16 matrix cores, in fp32 accuracy (each fp32 operand, scaled by a power of two, is the sum
of two f16 numbers to 2^-22; three exact product terms; fp32 sums).  The [rows, d_ffn] activation never exists in memory: the
forward writes its sign bits (d_ffn / 8 bytes per row) and the backward reads them.

The two weights are packed once per direction into the order the kernel streams them (``PackedFFN``), refreshed IN PLACE when
a parameter changes (``data_ptr`` / ``_version``): a replayed hipGraph keeps reading the same buffers.

No autograd here: the callers are hand-written forward / backward pairs (transformer._FrozenFFN, _FrozenFFNNorm)."""
import torch

from . import _lib

D_MODEL = 256


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def supported(x2: torch.Tensor, d_ffn: int) -> bool:
    return (x2.is_cuda and x2.dtype == torch.float32 and x2.dim() == 2 and x2.is_contiguous() and x2.shape[1] == D_MODEL
            and d_ffn % 256 == 0 and x2.data_ptr() % 16 == 0)


def pack(w1: torch.Tensor, b1, w2: torch.Tensor, backward: bool, out: torch.Tensor = None) -> torch.Tensor:
    """w1 [F, 256], b1 [F], w2 [256, F] (fp32, contiguous, on the GPU) -> the packed stream of one direction (uint8)."""
    F = w1.shape[0]
    assert w1.is_cuda and w1.dtype == torch.float32 and w1.shape == (F, D_MODEL) and w1.is_contiguous()
    assert w2.dtype == torch.float32 and w2.shape == (D_MODEL, F) and w2.is_contiguous() and w2.device == w1.device
    lib = _lib.load()
    n = lib.zira_ffn_f16x2_pack_bytes(F)
    if n == 0:
        raise RuntimeError("zira_ffn_f16x2: unsupported d_ffn %d" % F)
    if out is None:
        out = torch.empty(n, device=w1.device, dtype=torch.uint8)
    assert out.numel() == n and out.dtype == torch.uint8 and out.is_contiguous()
    with torch.cuda.device(w1.device):
        if backward:   # P = W2^T (P[h][k] = w2[k][h]), Q = W1^T (Q[n][h] = w1[h][n]); no bias in front of the mask
            rc = lib.zira_ffn_f16x2_pack_f32(w2.data_ptr(), 1, F, w1.data_ptr(), 1, D_MODEL, 0, F, out.data_ptr(), _stream(w1))
        else:
            assert b1 is not None and b1.shape == (F,) and b1.dtype == torch.float32 and b1.is_contiguous()
            rc = lib.zira_ffn_f16x2_pack_f32(w1.data_ptr(), D_MODEL, 1, w2.data_ptr(), F, 1, b1.data_ptr(), F, out.data_ptr(), _stream(w1))
    if rc != 0:
        raise RuntimeError("zira_ffn_f16x2_pack_f32 failed with code %d" % rc)
    return out


class PackedFFN:
    """The packed weights of one frozen FFN in both directions, refreshed in place when a parameter changes."""

    def __init__(self):
        self.key = [None, None]
        self.buf = [None, None]

    def get(self, w1, b1, w2, backward: bool) -> torch.Tensor:
        d = 1 if backward else 0
        key = (w1.data_ptr(), w1._version, w2.data_ptr(), w2._version, w1.device) + (() if backward else (b1.data_ptr(), b1._version))
        if key != self.key[d]:
            with torch.no_grad():
                same = self.buf[d] is not None and self.buf[d].device == w1.device
                self.buf[d] = pack(w1.detach(), None if backward else b1.detach(), w2.detach(), backward, self.buf[d] if same else None)
            self.key[d] = key
        return self.buf[d]

    def refresh(self, lin1, lin2):
        self.get(lin1.weight, lin1.bias, lin2.weight, False)
        self.get(lin1.weight, lin1.bias, lin2.weight, True)


def mask_like(x2: torch.Tensor, d_ffn: int) -> torch.Tensor:
    """The sign-bit tensor of a forward call on ``x2`` [rows, 256]: int32 [rows, d_ffn / 32] (a private layout)."""
    return torch.empty((x2.shape[0], d_ffn // 32), device=x2.device, dtype=torch.int32)


_WORKSPACES = {}


def workspace(device, stream: int, M: int, d_ffn: int) -> torch.Tensor:
    """Scratch of the launches of one (device, stream, shape): zeroed once, left zeroed by every launch where it must be
    (the kernel's tickets), kept for the life of the process (captured graphs hold its address)."""
    key = (device, stream, M, d_ffn)
    ws = _WORKSPACES.get(key)
    if ws is None:
        n = _lib.load().zira_ffn_f16x2_workspace_bytes(M, d_ffn)
        # (zeroed by a fill KERNEL, not torch.zeros: that may be a memset node when this first call happens inside a graph
        #  capture, and hipMemsetAsync nodes are replayed out of order on ROCm 7.2 -- scripts/repro_memset_graph.py;
        #  only the tickets at the front need the zeros: at most 256 row blocks of the last round, 256 bytes each)
        ws = _WORKSPACES[key] = torch.empty(max(n, 16), device=device, dtype=torch.uint8)
        ws[:min(ws.numel(), 256 * 256)].view(torch.int32).fill_(0)
    return ws


def run(a: torch.Tensor, packed: torch.Tensor, d_ffn: int, backward: bool, mask: torch.Tensor, q_bias: torch.Tensor = None,
        aux: torch.Tensor = None, out: torch.Tensor = None, use_workspace: bool = True) -> torch.Tensor:
    """forward: relu(a P^T + b1) Q^T (+ q_bias) (+ aux), ``mask`` written; backward: ((a P^T) * mask) Q^T (+ q_bias) (+ aux).
    ``out`` may be ``aux``."""
    M = a.shape[0]
    assert supported(a, d_ffn) and packed.dtype == torch.uint8 and packed.device == a.device
    assert mask.shape == (M, d_ffn // 32) and mask.dtype == torch.int32 and mask.is_contiguous() and mask.device == a.device
    if out is None:
        out = torch.empty_like(a)
    assert out.shape == a.shape and out.is_contiguous() and out.dtype == torch.float32
    if q_bias is not None:
        assert q_bias.shape == (D_MODEL,) and q_bias.is_contiguous() and q_bias.dtype == torch.float32
    if aux is not None:
        assert aux.shape == a.shape and aux.is_contiguous() and aux.dtype == torch.float32
    with torch.cuda.device(a.device):
        st = _stream(a)
        ws = workspace(a.device, st, M, d_ffn).data_ptr() if use_workspace else 0
        rc = _lib.load().zira_ffn_f16x2_f32(a.data_ptr(), packed.data_ptr(), M, d_ffn, 1 if backward else 0,
                                            0 if q_bias is None else q_bias.data_ptr(), 0 if aux is None else aux.data_ptr(),
                                            mask.data_ptr(), out.data_ptr(), ws, st)
    if rc != 0:
        raise RuntimeError("zira_ffn_f16x2_f32 failed with code %d (M=%d d_ffn=%d backward=%d)" % (rc, M, d_ffn, int(backward)))
    return out


def enabled() -> bool:
    """Whether the callers should take the fused f16x2 block (``transformer.Switches.gemm_arith`` = "f16x2")."""
    from .transformer import Switches
    return Switches.gemm_arith == "f16x2"
