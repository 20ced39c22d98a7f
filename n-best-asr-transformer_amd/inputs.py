"""Host-side input builder: words of an n-best utterance -> padded id / segment tensors.

Same contract as /root/reference/utils/bert_xlnet_inputs.py:4-104
(``prepare_inputs_for_roberta(raw_in, tokenizer, opt, device) -> (ids[B,S], seg[B,S] | None, lens)``):
``[CLS] sys.. [SEP] hyp1 [SEP] hyp2 .. [SEP]``, segment 0 for ``[CLS] sys..``, 1 after, 0 on pads;
XLM-R uses the doubled separator string; ``--without_system_act`` drops the system part and the
segment ids; ``--tod_pre_trained_model`` keeps ``[SYS]`` / ``[USR]`` markers.  Rows are right-padded
with ``tokenizer.pad_token_id`` to the batch maximum (no truncation, as the reference).

The per-word tokenisation is memoised (the reference re-tokenises every word of every batch in a
Python loop, :46-53); an optional ``n_best`` cut keeps only the first n hypotheses.
"""
import torch


class WordPieceTokenizer:
    """Minimal BERT-style tokenizer over a local vocabulary (no network): lower-cases, splits
    punctuation, greedy longest-match WordPiece.  Exposes what the input builder and the reference
    loop need: cls/sep/pad/unk tokens, pad_token_id, vocab_size, tokenize(), convert_tokens_to_ids()."""

    def __init__(self, vocab, do_lower_case=True, cls_token="[CLS]", sep_token="[SEP]", pad_token="[PAD]", unk_token="[UNK]"):
        if not isinstance(vocab, dict):
            vocab = {w: i for i, w in enumerate(vocab)}
        self.vocab = vocab
        self.cls_token, self.sep_token, self.pad_token, self.unk_token = cls_token, sep_token, pad_token, unk_token
        self.do_lower_case = do_lower_case
        self.pad_token_id = vocab[pad_token]
        self.special = {cls_token, sep_token, pad_token, unk_token, "[MASK]"}
        self._cache = {}

    @property
    def vocab_size(self):
        return len(self.vocab)

    @staticmethod
    def _is_punct(ch):
        cp = ord(ch)
        return (33 <= cp <= 47) or (58 <= cp <= 64) or (91 <= cp <= 96) or (123 <= cp <= 126)

    def _basic(self, text):
        if self.do_lower_case:
            text = text.lower()
        out, cur = [], ""
        for ch in text:
            if ch.isspace():
                if cur:
                    out.append(cur)
                cur = ""
            elif self._is_punct(ch):
                if cur:
                    out.append(cur)
                out.append(ch)
                cur = ""
            else:
                cur += ch
        if cur:
            out.append(cur)
        return out

    def _wordpiece(self, tok):
        if len(tok) > 100:
            return [self.unk_token]
        pieces, start = [], 0
        while start < len(tok):
            end, cur = len(tok), None
            while start < end:
                sub = tok[start:end]
                if start > 0:
                    sub = "##" + sub
                if sub in self.vocab:
                    cur = sub
                    break
                end -= 1
            if cur is None:
                return [self.unk_token]
            pieces.append(cur)
            start = end
        return pieces

    def tokenize(self, word):
        hit = self._cache.get(word)
        if hit is None:
            if word in self.special:
                hit = [word]
            else:
                hit = [p for t in self._basic(word) for p in self._wordpiece(t)]
            self._cache[word] = hit
        return list(hit)

    def convert_tokens_to_ids(self, toks):
        unk = self.vocab[self.unk_token]
        return [self.vocab.get(t, unk) for t in toks]


def cut_n_best(seq, n_best):
    """keep the first n hypotheses of ``.. [USR] h1 [SEP] h2 [SEP] ..`` (build extension; reference data has <= 10)"""
    if not n_best:
        return seq
    out, seen = [], 0
    usr = seq.index("[USR]")
    for i, w in enumerate(seq):
        if i > usr and w == "[SEP]":
            seen += 1
            if seen >= n_best:
                break
        out.append(w)
    return out


def prepare_inputs_for_roberta(raw_in, tokenizer, opt, device, n_best=None):
    family = getattr(opt, "pre_trained_model", None)
    tod = getattr(opt, "tod_pre_trained_model", None)
    no_sys = getattr(opt, "without_system_act", False)
    sep = tokenizer.sep_token
    first_sep = sep + sep if family == "xlm-roberta" else sep
    rows, segs = [], []
    for seq in raw_in:
        seq = cut_n_best(list(seq), n_best)
        usr = seq.index("[USR]")
        a_words, b_words = seq[2:usr], seq[usr + 1:]            # drops "[CLS] [SYS]" (:24-28)
        if tod:
            a_words, b_words = ["[SYS]"] + a_words, ["[USR]"] + b_words
        b_words = [first_sep if w == "[SEP]" else w for w in b_words]
        a = [t for w in a_words for t in tokenizer.tokenize(w)]
        b = [t for w in b_words for t in tokenizer.tokenize(w)]
        if tod:
            left, right = [tokenizer.cls_token] + a, b + [sep]
        elif no_sys:
            rows.append([tokenizer.cls_token] + b + [sep])
            continue
        else:
            left, right = [tokenizer.cls_token] + a, [first_sep] + b + [sep]
        rows.append(left + right)
        segs.append([0] * len(left) + [1] * len(right))
    lens = [len(r) for r in rows]
    width = max(lens)
    pad = tokenizer.pad_token_id
    ids = torch.tensor([tokenizer.convert_tokens_to_ids(r) + [pad] * (width - len(r)) for r in rows], dtype=torch.long, device=device)
    seg = None
    if segs:
        seg = torch.tensor([s + [0] * (width - len(s)) for s in segs], dtype=torch.long, device=device)
    return ids, seg, lens
