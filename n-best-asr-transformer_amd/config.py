"""Encoder and label-space configuration.

The reference builds its encoder by model NAME (/root/reference/n_best_asr_bert.py:33-37,481-487);
the shapes below are those models' published configurations (bert-base-uncased, xlm-roberta-base,
xlm-roberta-large).  ``fea_dim`` follows the encoder width instead of the reference's hard-coded
768 (/root/reference/models/model.py:30).
"""
from dataclasses import dataclass, field, asdict


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
    cls_token_id: int = 101
    sep_token_id: int = 102
    family: str = "bert"            # "bert" | "roberta" | "xlm-roberta"

    def to_dict(self):
        return asdict(self)

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads


def bert_base(**kw):
    return EncoderConfig(**kw)


def xlmr_base(**kw):
    base = dict(vocab_size=250002, max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5,
                pad_token_id=1, cls_token_id=0, sep_token_id=2, family="xlm-roberta")
    base.update(kw)
    return EncoderConfig(**base)


def xlmr_large(**kw):
    large = dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
    large.update(kw)
    return xlmr_base(**large)


NAMED = {"bert": bert_base, "bert-base-uncased": bert_base, "xlm-roberta": xlmr_base,
         "xlm-roberta-base": xlmr_base, "xlm-roberta-large": xlmr_large}


@dataclass
class LabelSpace:
    """The STC label hierarchy of memory.pt (top2bottom_dict, idx2label):
    /root/reference/n_best_asr_bert.py:489-496."""
    top2bottom: dict                      # int top -> list[int bottom]
    idx2label: list = field(default_factory=list)

    def __post_init__(self):
        self.top2bottom = {int(k): [int(b) for b in v] for k, v in self.top2bottom.items()}
        self.n_top = len(self.top2bottom)
        self.n_bottom = sum(len(v) for v in self.top2bottom.values())
        self.multi = [t for t in range(self.n_top) if len(self.top2bottom[t]) >= 2]
        if not self.idx2label:
            self.idx2label = ["lbl%d" % i for i in range(self.n_bottom)]
        seen = {}
        for t, bs in self.top2bottom.items():
            for b in bs:
                if b in seen:
                    raise ValueError("map from bottom to top should be unique")
                seen[b] = t
        self.bottom2top = [seen[b] for b in range(self.n_bottom)]

    @property
    def n_head_rows(self):
        """rows of the concatenated head matrix: n_top + sum over multi-value tops of n_k."""
        return self.n_top + sum(len(self.top2bottom[t]) for t in self.multi)

    @staticmethod
    def from_json(path):
        import json
        with open(path) as f:
            d = json.load(f)
        return LabelSpace(d["top2bottom"], d.get("idx2label", []))
