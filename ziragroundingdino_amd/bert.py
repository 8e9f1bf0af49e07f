"""Frozen text encoder: a plain-PyTorch BERT (bert-base-uncased geometry by default).

Out of kernel scope: frozen, upstream of the language side branch.
The reference wraps HuggingFace's BertModel (bertwarper.py ``BertModelWarper``) to feed it the
block-diagonal ``[bs, T, T]`` sub-sentence mask and per-phrase position ids; that wrapper does
not run on the transformers release in this image and no pretrained weights are available
offline, so the same computation is written out here with HF's parameter names
(``embeddings.word_embeddings.weight``, ``encoder.layer.N.attention.self.query.weight`` ...) so a
bert-base-uncased state dict loads unchanged.  Attention: two batched GEMMs and a softmax (captions are short;
torch SDPA beyond transformer.SMALL_ATTENTION_SCORES).
"""
import zlib
from types import SimpleNamespace

import torch
import torch.nn.functional as F
from torch import nn

from .transformer import SMALL_ATTENTION_SCORES, Switches, _attention_small


class BertConfig(SimpleNamespace):
    def __init__(self, vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2,
                 layer_norm_eps=1e-12, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1):
        super().__init__(**{k: v for k, v in locals().items() if k not in ("self", "__class__")})


class _Embeddings(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.word_embeddings = nn.Embedding(c.vocab_size, c.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(c.max_position_embeddings, c.hidden_size)
        self.token_type_embeddings = nn.Embedding(c.type_vocab_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.dropout = nn.Dropout(c.hidden_dropout_prob)
        # checkpoints written by transformers < 4.31 carry `embeddings.position_ids`; it is not a
        # parameter of the model: dropped on load so that strict loading of such checkpoints works
        self._register_load_state_dict_pre_hook(self._drop_position_ids)

    @staticmethod
    def _drop_position_ids(state_dict, prefix, *_):
        state_dict.pop(prefix + "position_ids", None)

    def forward(self, input_ids, token_type_ids, position_ids):
        x = (self.word_embeddings(input_ids) + self.token_type_embeddings(token_type_ids)
             + self.position_embeddings(position_ids))
        return self.dropout(self.LayerNorm(x))


class _SelfAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.h = c.num_attention_heads
        self.query = nn.Linear(c.hidden_size, c.hidden_size)
        self.key = nn.Linear(c.hidden_size, c.hidden_size)
        self.value = nn.Linear(c.hidden_size, c.hidden_size)
        self.p_drop = c.attention_probs_dropout_prob

    def forward(self, x, bias):
        B, T, C = x.shape
        split = lambda t: t.view(B, T, self.h, C // self.h).transpose(1, 2)
        q, k, v = split(self.query(x)), split(self.key(x)), split(self.value(x))
        p_drop = self.p_drop if self.training else 0.0  # (HF BertSelfAttention drops attention probabilities)
        if Switches.small_attention and B * self.h * T * T <= SMALL_ATTENTION_SCORES:
            o = _attention_small(q, k, v, bias, p_drop)   # captions are <= 256 tokens: see transformer._attention_small
        else:
            o = F.scaled_dot_product_attention(q, k, v, attn_mask=bias, dropout_p=p_drop)
        return o.transpose(1, 2).reshape(B, T, C)


class _SelfOutput(nn.Module):
    def __init__(self, c, in_features=None):
        super().__init__()
        self.dense = nn.Linear(in_features or c.hidden_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.dropout = nn.Dropout(c.hidden_dropout_prob)

    def forward(self, h, residual):
        return self.LayerNorm(self.dropout(self.dense(h)) + residual)


class _Attention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.self = _SelfAttention(c)
        self.output = _SelfOutput(c)

    def forward(self, x, bias):
        return self.output(self.self(x, bias), x)


class _Intermediate(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.intermediate_size)

    def forward(self, x):
        return F.gelu(self.dense(x))


class _Layer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.attention = _Attention(c)
        self.intermediate = _Intermediate(c)
        self.output = _SelfOutput(c, c.intermediate_size)

    def forward(self, x, bias):
        x = self.attention(x, bias)
        return self.output(self.intermediate(x), x)


class _Encoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layer = nn.ModuleList(_Layer(c) for _ in range(c.num_hidden_layers))


class _Pooler(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.hidden_size)


class BertModel(nn.Module):
    """forward(input_ids, attention_mask [bs,T] or [bs,T,T], token_type_ids, position_ids)
    -> {"last_hidden_state": [bs,T,hidden]} (what GroundingDINO.forward consumes, reference :457)."""

    def __init__(self, config: BertConfig = None):
        super().__init__()
        self.config = config or BertConfig()
        self.embeddings = _Embeddings(self.config)
        self.encoder = _Encoder(self.config)
        self.pooler = _Pooler(self.config)

    def forward(self, input_ids, attention_mask=None, token_type_ids=None, position_ids=None, **_):
        B, T = input_ids.shape
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        if position_ids is None:
            position_ids = torch.arange(T, device=input_ids.device)[None].expand(B, T)
        x = self.embeddings(input_ids, token_type_ids, position_ids)
        if attention_mask is None:
            bias = None
        else:
            m = attention_mask.bool()
            m = m[:, None, :, :] if m.dim() == 3 else m[:, None, None, :]
            bias = torch.zeros(m.shape, dtype=x.dtype, device=x.device).masked_fill(~m, torch.finfo(x.dtype).min)
        for layer in self.encoder.layer:
            x = layer(x, bias)
        return {"last_hidden_state": x}


class SimpleTokenizer:
    """Offline stand-in for the BERT word-piece tokenizer (no vocabulary file in this image):
    whitespace / punctuation split, one id per word by CRC32, BERT's ids for the special tokens
    the mask generator keys on ([CLS] 101, [SEP] 102, '.' 1012, '?' 1029), ``padding='longest'``."""

    cls_id, sep_id, pad_id = 101, 102, 0
    specials = {"[CLS]": 101, "[SEP]": 102, ".": 1012, "?": 1029}

    def convert_tokens_to_ids(self, tokens):
        return [self.specials.get(t, self._word_id(t)) for t in tokens]

    @staticmethod
    def _word_id(w):
        return 1996 + zlib.crc32(w.lower().encode()) % 28000

    def _encode(self, text):
        ids = [self.cls_id]
        for chunk in text.replace(".", " . ").replace("?", " ? ").split():
            ids.append(self.specials.get(chunk, self._word_id(chunk)))
        return ids + [self.sep_id]

    def __call__(self, captions, padding="longest", return_tensors="pt"):
        rows = [self._encode(c) for c in captions]
        T = max(len(r) for r in rows)
        input_ids = torch.full((len(rows), T), self.pad_id, dtype=torch.long)
        attention_mask = torch.zeros((len(rows), T), dtype=torch.long)
        for i, r in enumerate(rows):
            input_ids[i, :len(r)] = torch.tensor(r)
            attention_mask[i, :len(r)] = 1
        return TokenBatch(input_ids=input_ids, attention_mask=attention_mask,
                          token_type_ids=torch.zeros_like(input_ids))


class TokenBatch(dict):
    """dict with attribute access and .to(device), like transformers' BatchEncoding."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def to(self, device):
        return TokenBatch({k: v.to(device) for k, v in self.items()})
