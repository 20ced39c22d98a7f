"""Oracle: BERT / XLM-R encoder, plain PyTorch fp32 on CPU.

Restates the third-party encoder the reference instantiates by name at
/root/reference/n_best_asr_bert.py:33-37,481-487 and calls at
/root/reference/models/model.py:43-45 (ASR n-best ids) and :54-56 (transcript).
Module / parameter names follow the HuggingFace layout so a ``state_dict`` is
interchangeable with the reference's ``model.pt`` (models/model.py:75-83).

Algorithm (published BERT, post-LN):
  emb   = LN(word[ids] + type[seg] + pos[position_ids])            eps 1e-12 (BERT) / 1e-5 (XLM-R)
  layer = x -> LN(x + drop(Wo . attn(x))) -> LN(x1 + drop(W2 . gelu_erf(W1 . x1)))
  attn  = softmax(Q K^T / sqrt(d) + keymask) -> drop -> . V          keymask: -inf on masked keys

TEST INFRASTRUCTURE - see oracle/__init__.py.
"""
import math
from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class EncoderConfig:
    vocab_size: int = 30522
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    layer_norm_eps: float = 1e-12
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1
    pad_token_id: int = 0
    family: str = "bert"  # "bert" | "xlm-roberta"

    @staticmethod
    def bert_base(**kw):
        return EncoderConfig(**kw)

    @staticmethod
    def xlmr_base(**kw):
        base = dict(vocab_size=250002, max_position_embeddings=514, type_vocab_size=1,
                    layer_norm_eps=1e-5, pad_token_id=1, family="xlm-roberta")
        base.update(kw)
        return EncoderConfig(**base)


def position_ids_for(cfg, input_ids):
    """BERT: arange(S).  XLM-R: cumsum(ids != pad) * (ids != pad) + pad  (pad-offset positions,
    installed transformers modeling_xlm_roberta.py:142-155)."""
    B, S = input_ids.shape
    if cfg.family == "xlm-roberta":
        nonpad = input_ids.ne(cfg.pad_token_id).long()
        return torch.cumsum(nonpad, dim=1) * nonpad + cfg.pad_token_id
    return torch.arange(S, dtype=torch.long, device=input_ids.device).unsqueeze(0).expand(B, S)


class _Embeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        # padding_idx: the row of pad_token_id never receives a gradient (it matters for XLM-R, whose pad
        # positions ARE attended under the reference's ids>0 mask)
        self.word_embeddings = nn.Embedding(cfg.vocab_size, cfg.hidden_size, padding_idx=cfg.pad_token_id)
        # RoBERTa-family position tables also carry padding_idx (pad positions map to row pad_token_id)
        self.position_embeddings = nn.Embedding(
            cfg.max_position_embeddings, cfg.hidden_size,
            padding_idx=cfg.pad_token_id if cfg.family in ("roberta", "xlm-roberta") else None)
        self.token_type_embeddings = nn.Embedding(cfg.type_vocab_size, cfg.hidden_size)
        self.LayerNorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.p = cfg.hidden_dropout_prob

    def forward(self, ids, seg, pos):
        e = self.word_embeddings(ids) + self.token_type_embeddings(seg)
        e = e + self.position_embeddings(pos)
        return F.dropout(self.LayerNorm(e), self.p, self.training)


class _SelfAttention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        H = cfg.hidden_size
        self.query, self.key, self.value = nn.Linear(H, H), nn.Linear(H, H), nn.Linear(H, H)
        self.nh = cfg.num_attention_heads
        self.p = cfg.attention_probs_dropout_prob

    def forward(self, x, key_mask):
        B, S, H = x.shape
        d = H // self.nh
        split = lambda t: t.view(B, S, self.nh, d).transpose(1, 2)        # [B,h,S,d]
        q, k, v = split(self.query(x)), split(self.key(x)), split(self.value(x))
        scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(d)       # [B,h,S,S]
        scores = scores.masked_fill(~key_mask[:, None, None, :], float("-inf"))
        probs = F.dropout(torch.softmax(scores, dim=-1), self.p, self.training)
        ctx = torch.matmul(probs, v)                                       # [B,h,S,d]
        return ctx.transpose(1, 2).reshape(B, S, H)


class _SelfOutput(nn.Module):
    def __init__(self, cfg, in_dim):
        super().__init__()
        self.dense = nn.Linear(in_dim, cfg.hidden_size)
        self.LayerNorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.p = cfg.hidden_dropout_prob

    def forward(self, h, residual):
        return self.LayerNorm(F.dropout(self.dense(h), self.p, self.training) + residual)


class _Attention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.self = _SelfAttention(cfg)
        self.output = _SelfOutput(cfg, cfg.hidden_size)

    def forward(self, x, key_mask):
        return self.output(self.self(x, key_mask), x)


class _Intermediate(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.intermediate_size)

    def forward(self, x):
        u = self.dense(x)
        return 0.5 * u * (1.0 + torch.erf(u * (1.0 / math.sqrt(2.0))))     # gelu, erf form


class _Layer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.attention = _Attention(cfg)
        self.intermediate = _Intermediate(cfg)
        self.output = _SelfOutput(cfg, cfg.intermediate_size)

    def forward(self, x, key_mask):
        x1 = self.attention(x, key_mask)
        return self.output(self.intermediate(x1), x1)


class _Stack(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer = nn.ModuleList([_Layer(cfg) for _ in range(cfg.num_hidden_layers)])


class _Pooler(nn.Module):
    """Present only for state_dict compatibility: the reference slices the raw CLS row
    (models/model.py:46-47) and never calls the pooler, so it never receives a gradient."""

    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.hidden_size)


class OracleEncoder(nn.Module):
    """Call contract of the object the reference injects at models/model.py:19:
    ``enc(input_ids=, attention_mask=[, token_type_ids=])[0] -> [B,S,H]``."""

    def __init__(self, cfg: EncoderConfig):
        super().__init__()
        self.cfg = cfg
        self.embeddings = _Embeddings(cfg)
        self.encoder = _Stack(cfg)
        self.pooler = _Pooler(cfg)

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None,
                return_all=False):
        cfg = self.cfg
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids, dtype=torch.bool)
        key_mask = attention_mask.bool()
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        if position_ids is None:
            position_ids = position_ids_for(cfg, input_ids)
        x = self.embeddings(input_ids, token_type_ids, position_ids)
        hs = [x]
        for lyr in self.encoder.layer:
            x = lyr(x, key_mask)
            hs.append(x)
        return (x, hs) if return_all else (x,)
