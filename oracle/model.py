"""Oracle: the wrapper module, restating /root/reference/models/model.py:11-73
(TOD_ASR_Transformer_STC): encoder on the ASR n-best ids, encoder on the transcript ids, raw CLS
rows, STC heads on the ASR CLS row.  attention_mask = input_ids > 0 for EVERY family (quirk Q1,
models/model.py:43,45,54,56); XLM-R never receives token_type_ids (:42-43,53-54).

TEST INFRASTRUCTURE - see oracle/__init__.py.
"""
import torch.nn as nn

from .encoder import OracleEncoder
from .stc import OracleHeads


class OracleModel(nn.Module):
    def __init__(self, cfg, top2bottom, n_bottom, dropout):
        super().__init__()
        self.family = cfg.family
        self.bert_encoder = OracleEncoder(cfg)
        self.clf = OracleHeads(top2bottom, cfg.hidden_size, n_bottom, dropout)

    def _encode(self, ids, seg):
        if self.family == "xlm-roberta":
            seg = None
        return self.bert_encoder(input_ids=ids, attention_mask=ids > 0, token_type_ids=seg)[0][:, 0, :]

    def forward(self, input_ids, trans_input_ids=None, seg_ids=None, trans_seg_ids=None,
                classifier_input_type="asr"):
        asr_cls = self._encode(input_ids, seg_ids)
        trans_cls = self._encode(trans_input_ids, trans_seg_ids) if trans_input_ids is not None else None
        feats = trans_cls if classifier_input_type == "transcript" else asr_cls
        top, bottoms, final = self.clf(feats)
        return top, bottoms, final, asr_cls, trans_cls
