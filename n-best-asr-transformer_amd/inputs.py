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


class SentencePieceTokenizer:
    """XLM-R style tokenizer over a LOCAL sentencepiece model (no network): the id convention of
    ``XLMRobertaTokenizer`` — ``<s>``=0, ``<pad>``=1, ``</s>``=2, ``<unk>``=3, every sentencepiece id shifted by +1
    (fairseq offset), ``<mask>`` last — behind the interface the input builder uses."""

    def __init__(self, model_file):
        import re
        import sentencepiece as spm
        self.sp = spm.SentencePieceProcessor(model_file=model_file)
        self.cls_token, self.sep_token, self.pad_token, self.unk_token = "<s>", "</s>", "<pad>", "<unk>"
        self.special_ids = {"<s>": 0, "<pad>": 1, "</s>": 2, "<unk>": 3}
        self.pad_token_id = 1
        self._mask_id = self.sp.get_piece_size() + 1
        self._split = re.compile("(<s>|</s>|<pad>|<unk>|<mask>)")
        self._cache = {}

    @property
    def vocab_size(self):
        return self.sp.get_piece_size() + 2

    def tokenize(self, word):
        hit = self._cache.get(word)
        if hit is None:
            hit = []
            for part in self._split.split(word):           # "</s></s>" -> two separator tokens
                if not part:
                    continue
                hit += [part] if self._split.fullmatch(part) else self.sp.encode(part, out_type=str)
            self._cache[word] = hit
        return list(hit)

    def convert_tokens_to_ids(self, toks):
        out = []
        for t in toks:
            if t in self.special_ids:
                out.append(self.special_ids[t])
            elif t == "<mask>":
                out.append(self._mask_id)
            else:
                i = self.sp.piece_to_id(t)
                out.append(3 if i == self.sp.unk_id() else i + 1)
        return out


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


def encode_utterance(seq, tokenizer, opt, n_best=None, max_seq_len=None):
    """one utterance -> (ids, seg | None) as python int lists, un-padded (bert_xlnet_inputs.py:20-85).

    ``max_seq_len`` (build extension; the reference never truncates) cuts the tail hypotheses and re-closes the
    sequence with the separator, so over-long n-best lists degrade to fewer hypotheses instead of failing."""
    family = getattr(opt, "pre_trained_model", None)
    tod = getattr(opt, "tod_pre_trained_model", None)
    no_sys = getattr(opt, "without_system_act", False)
    sep = tokenizer.sep_token
    first_sep = sep + sep if family == "xlm-roberta" else sep
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
        left, right = [], [tokenizer.cls_token] + b + [sep]
    else:
        left, right = [tokenizer.cls_token] + a, [first_sep] + b + [sep]
    toks = left + right
    seg = None if (no_sys and not tod) else [0] * len(left) + [1] * len(right)
    if max_seq_len and len(toks) > max_seq_len:
        toks = toks[:max_seq_len - 1] + [sep]
        seg = None if seg is None else seg[:max_seq_len - 1] + [1]
    return tokenizer.convert_tokens_to_ids(toks), seg


def collate(rows, pad_id, pin=False, alloc=None):
    """[(ids, seg | None)] -> right-padded int64 host tensors ``ids [B,S]``, ``seg [B,S] | None`` and the lengths
    (bert_xlnet_inputs.py:87-102: pad id for ids, 0 for segments, width = batch maximum).  One masked numpy assignment per
    tensor (round 3 copied row by row through torch.as_tensor: 18 ms per 256-utterance batch - with the real-data loop's
    prefetch thread sharing the interpreter lock with the training loop, that was most of a step).
    ``alloc(which, shape)`` -> an int64 host tensor of that shape to fill (``which`` 0 ids, 1 segments): the prefetcher hands out
    slices of long-lived PINNED staging buffers instead of a fresh pin_memory() per tensor per batch (trainer.PinnedStage)."""
    import numpy as np
    lens = [len(r[0]) for r in rows]
    width = max(lens)
    inside = np.arange(width)[None, :] < np.asarray(lens)[:, None]

    def filled(which, value, col):
        if alloc is None:
            a = np.full((len(rows), width), value, dtype=np.int64)
            t = None
        else:
            t = alloc(which, (len(rows), width))
            a = t.numpy()
            a.fill(value)
        a[inside] = np.concatenate([np.asarray(r[col], dtype=np.int64) for r in rows])
        if t is None:
            t = torch.from_numpy(a)
            t = t.pin_memory() if pin else t
        return t

    ids = filled(0, pad_id, 0)
    seg = filled(1, 0, 1) if rows[0][1] is not None else None
    return ids, seg, lens


def prepare_inputs_for_roberta(raw_in, tokenizer, opt, device, n_best=None, max_seq_len=None):
    rows = [encode_utterance(seq, tokenizer, opt, n_best, max_seq_len) for seq in raw_in]
    ids, seg, lens = collate(rows, tokenizer.pad_token_id)
    return ids.to(device), (None if seg is None else seg.to(device)), lens
