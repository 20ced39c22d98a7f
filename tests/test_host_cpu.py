"""CPU tests (no GPU): the C-ABI library loads and exports every symbol include/nbest_hip.h declares,
host-side logic (input builder, fscore, schedule, arena layout, synthetic generator), and the product
path fails loudly instead of falling back when the HIP side is unavailable."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT

import nbest_amd  # noqa: F401
from nbest_amd import config as ncfg, fscore, hipabi, inputs, synth
from nbest_amd.optim import warmup_linear


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "nbest_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(nbest_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    assert os.path.exists(hipabi.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(hipabi.LIB_PATH)
    for sym in declared:
        assert hasattr(lib, sym), "libnbest_hip.so does not export %s" % sym
    assert sorted(hipabi.EXPORTS) == declared
    assert lib.nbest_version() == 1
    lib.nbest_rowred_ws_bytes.restype = ctypes.c_size_t
    lib.nbest_rowred_ws_bytes.argtypes = [ctypes.c_int64, ctypes.c_int64]
    assert lib.nbest_rowred_ws_bytes(32768, 768) > 0
    assert ctypes.sizeof(hipabi.TensorDesc) == 32 and ctypes.sizeof(hipabi.LayerOffsets) == 96


def test_no_cpu_fallback_in_product_package():
    """nothing under the product package may import the oracle"""
    pkg = os.path.join(ROOT, "n-best-asr-transformer_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_product_fails_loudly_without_gpu(labels):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from nbest_amd.model import NBestSTCModel
    with pytest.raises(Exception):
        NBestSTCModel(ncfg.bert_base(num_hidden_layers=1, vocab_size=200), labels, device="cuda")


def test_fscore_known_answers():
    kat = json.load(open(os.path.join(GOLDEN, "kat.json")))
    for c in kat["update_f1"]:
        base = (0, 0, 0) if c["pred"] else (2, 3, 4)
        assert list(fscore.update_f1(c["pred"], c["gold"], *base)) == c["out"]
    for c in kat["compute_f1"]:
        assert list(fscore.compute_f1(*c["inp"])) == pytest.approx(c["out"])
    assert fscore.update_f1(["a", "b"], ["b", "c"], 0, 0, 0) == (1, 1, 1)
    assert fscore.compute_f1(1, 1, 1) == (50, 50, 50) and fscore.compute_f1(0, 3, 9) == (0, 0, 0)


def test_warmup_linear_schedule():
    assert warmup_linear(0, 100, 0.1) == 0.0
    assert warmup_linear(5, 100, 0.1) == pytest.approx(0.5)
    assert warmup_linear(10, 100, 0.1) == pytest.approx(1.0)
    assert warmup_linear(55, 100, 0.1) == pytest.approx(0.5)
    assert warmup_linear(100, 100, 0.1) == 0.0 and warmup_linear(150, 100, 0.1) == 0.0
    assert warmup_linear(7, -1, 0.1) == 1.0


VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "hello", "how", "may", "i", "help", "you", "?", ",", "want", "cheap",
         "food", "chi", "##nese", "##s", "uh"]


class _Opt:
    pre_trained_model = "bert"
    tod_pre_trained_model = None
    without_system_act = False


def test_input_builder_layout():
    tok = inputs.WordPieceTokenizer(VOCAB)
    raw = [["[CLS]", "[SYS]", "Hello", ",", "how", "may", "I", "help", "you?", "[USR]", "i", "want", "cheap", "chinese", "[SEP]",
            "uh", "foods", "zzz"],
           ["[CLS]", "[SYS]", "hello", "[USR]", "cheap", "[SEP]", "food"]]
    ids, seg, lens = inputs.prepare_inputs_for_roberta(raw, tok, _Opt(), "cpu")
    v = {w: i for i, w in enumerate(VOCAB)}
    row0 = [v["[CLS]"], v["hello"], v[","], v["how"], v["may"], v["i"], v["help"], v["you"], v["?"], v["[SEP]"],
            v["i"], v["want"], v["cheap"], v["chi"], v["##nese"], v["[SEP]"], v["uh"], v["food"], v["##s"], v["[UNK]"], v["[SEP]"]]
    assert lens == [len(row0), 7] and ids.shape == (2, len(row0))
    assert ids[0].tolist() == row0
    assert ids[1].tolist() == [v["[CLS]"], v["hello"], v["[SEP]"], v["cheap"], v["[SEP]"], v["food"], v["[SEP]"]] + [0] * (len(row0) - 7)
    assert seg[0].tolist() == [0] * 9 + [1] * 12
    assert seg[1].tolist() == [0, 0, 1, 1, 1, 1, 1] + [0] * (len(row0) - 7)          # pads carry segment 0
    # --without_system_act: [CLS] user.. [SEP], no segment ids
    o = _Opt()
    o.without_system_act = True
    ids2, seg2, _ = inputs.prepare_inputs_for_roberta(raw[1:], tok, o, "cpu")
    assert seg2 is None and ids2[0].tolist() == [v["[CLS]"], v["cheap"], v["[SEP]"], v["food"], v["[SEP]"]]
    # n_best cut keeps the first hypothesis only
    ids3, _, _ = inputs.prepare_inputs_for_roberta(raw[1:], tok, _Opt(), "cpu", n_best=1)
    assert ids3[0].tolist() == [v["[CLS]"], v["hello"], v["[SEP]"], v["cheap"], v["[SEP]"]]


def test_synthetic_batch_layout(labels):
    cfg = ncfg.bert_base()
    b = synth.nbest_batch(cfg, labels, 16, 128, n_best=5, seed=3, ragged=True, trans_len=32)
    ids, seg, y = b["ids"], b["seg"], b["labels"]
    assert ids.shape == (16, 128) and (ids[:, 0] == cfg.cls_token_id).all()
    for r in range(16):
        n = int((ids[r] != 0).sum())
        assert (ids[r, :n] != 0).all() and (ids[r, n:] == 0).all()              # right padded
        assert int((ids[r] == cfg.sep_token_id).sum()) == 6                      # [SEP] after sys + 5 hypotheses
        first_sep = int(np.argmax(ids[r] == cfg.sep_token_id))
        assert (seg[r, :first_sep] == 0).all() and (seg[r, first_sep:n] == 1).all() and (seg[r, n:] == 0).all()
        assert ids[r, n - 1] == cfg.sep_token_id
    assert (ids[0] != 0).all()                                                  # at least one full-length row
    for t in labels.multi:                                                       # <= 1 active bottom per multi-value top
        assert (y[:, labels.top2bottom[t]].sum(1) <= 1).all()
    assert ((y.sum(1) >= 1) & (y.sum(1) <= 3)).all()
    b2 = synth.nbest_batch(cfg, labels, 16, 128, n_best=5, seed=3, ragged=True, trans_len=32)
    assert all(np.array_equal(b[k], b2[k]) for k in b)                           # deterministic


def test_arena_layout(labels):
    """layout only (no device allocation): contiguity of the fused QKV / head matrices, alignment, HF names"""
    from nbest_amd import arena as ar
    cfg = ncfg.bert_base(num_hidden_layers=2, vocab_size=1000)
    a = ar.ParamArena(cfg, labels, "cpu", compute_dtype=torch.float32)
    H = cfg.hidden_size
    for i in range(2):
        p = "bert_encoder.encoder.layer.%d.attention.self." % i
        q, k, v = (a.by_name[p + n + ".weight"] for n in ("query", "key", "value"))
        assert k.offset == q.offset + H * H and v.offset == k.offset + H * H and q.offset % 64 == 0
        qb, kb, vb = (a.by_name[p + n + ".bias"] for n in ("query", "key", "value"))
        assert kb.offset == qb.offset + H and vb.offset == kb.offset + H
        assert a.layer_offsets[i].wqkv == q.offset and a.layer_offsets[i].bqkv == qb.offset
    Wh, bh = a.heads_wb()
    assert Wh.shape == (171, H) and bh.shape == (171,)
    names = [s.name for s in a.slots]
    sd_names = ["bert_encoder." + n for n, _ in synth.encoder_param_shapes(cfg)] + ["clf." + n for n, _ in synth.head_param_shapes(labels, H)]
    assert sorted(names) == sorted(sd_names)
    spans = sorted((s.offset, s.offset + s.numel) for s in a.slots)
    assert all(spans[i][1] <= spans[i + 1][0] for i in range(len(spans) - 1)) and spans[-1][1] <= a.total
    sd = synth.model_state(cfg, labels, seed=1)
    a.w16 = None
    a.load_state(sd)
    np.testing.assert_array_equal(a.view(a.p, "clf.linear_layers.lin_2.weight").numpy(), sd["clf.linear_layers.lin_2.weight"])
    np.testing.assert_array_equal(Wh[:30].numpy(), sd["clf.top_linear_layer.weight"])
    np.testing.assert_array_equal(Wh[30:105].numpy(), sd["clf.linear_layers.lin_2.weight"])
    lo, hi = a.layer_range[1]
    assert lo == a.layer_offsets[1].wqkv and hi > a.layer_offsets[1].ln2_b


def test_observability_files_match_reference(tmp_path):
    """per-epoch CSV + per-label report (tod_asr_util.py:150-223) byte-for-byte against files the reference wrote for the
    same cases; ontology filter (n_best_asr_bert.py:218-229) against the reference's output"""
    import json
    from nbest_amd import observe, trainer
    g = os.path.join(os.path.dirname(__file__), "golden", "observe")
    d = json.load(open(os.path.join(g, "inputs.json")))
    cases = [(r.split(" "), p, gd) for r, p, gd in zip(d["raw"], d["pred"], d["gold"])]
    eic = observe.EpochInfoCollector.from_cases(cases, d["mean_loss"], d["prf"], d["acc"])
    observe.observability_lens(eic, 3, "valid", str(tmp_path), "tod_asr_bert_stc")
    for fn in ("classification_report_epoch_3_for_valid.txt", "epoch_3_for_valid_observe_tod_asr_bert_stc.csv"):
        assert open(os.path.join(g, fn), "rb").read() == open(os.path.join(str(tmp_path), fn), "rb").read(), fn
    assert trainer.filter_informative(d["onto_labels"], d["ontology"]) == d["onto_filtered"]


def test_encoded_split_and_prefetcher_match_per_batch_builder(labels):
    """tokenise-once split + prefetching iterator produce exactly the tensors of the per-batch
    builder, in order, for every rank's slice; max_seq_len truncation re-closes the sequence"""
    from nbest_amd import trainer
    vocab = json.load(open(os.path.join(GOLDEN, "text_vocab.json")))
    tok = inputs.WordPieceTokenizer(vocab)
    data = trainer.read_wcn_data(os.path.join(GOLDEN, "valid_head.txt"))
    memory = dict(label2idx={l: i for i, l in enumerate(labels.idx2label)})
    opt = type("O", (), dict(tokenizer=tok, pre_trained_model="bert", n_best=4, max_seq_len=None))()
    split = trainer.EncodedSplit(data, opt, memory)
    lists = trainer.batch_indices(len(split), 7, shuffle=True, seed=3)
    assert sorted(j for l in lists for j in l) == list(range(24))
    for world in (1, 2):
        for rank in range(world):
            seen = 0
            for bi, mine, b in trainer.Prefetcher(split, lists, "cpu", rank, world):
                lo, hi = trainer.shard_bounds(len(lists[bi]), rank, world)
                assert mine == lists[bi][lo:hi]
                ids, seg, _ = inputs.prepare_inputs_for_roberta([data[0][j] for j in mine], tok, opt, "cpu", n_best=4)
                tids, _, _ = inputs.prepare_inputs_for_roberta([data[1][j] for j in mine], tok, opt, "cpu")
                y = trainer.labels_to_multihot([data[2][j] for j in mine], memory["label2idx"], "cpu")
                assert torch.equal(b["ids"], ids) and torch.equal(b["seg"], seg) and torch.equal(b["tids"], tids)
                assert torch.equal(b["labels"], y)
                seen += 1
            assert seen == len(lists)
    # the GPU loop's staging path: every tensor of a batch written into long-lived (there: pinned) buffers, reused batch after batch -
    # the same tensors as the allocating path, also when a LARGER batch follows a smaller one in the same stage
    stage = trainer.PinnedStage(pin=False)
    for idx in ([3], lists[0], lists[1][:2], list(range(24))):
        a, b = split.host_batch(idx), split.host_batch(idx, stage=stage)
        assert a.keys() == b.keys()
        for k in a:
            assert (a[k] is None and b[k] is None) or (a[k].dtype == b[k].dtype and torch.equal(a[k], b[k])), k
    # a rank with an empty slice still sees the batch (it must join that step's collectives)
    got = [(bi, mine) for bi, mine, _ in trainer.Prefetcher(split, [[0], [1, 2]], "cpu", 1, 2)]
    assert got == [(0, []), (1, [2])]
    full, _ = inputs.encode_utterance(data[0][0], tok, opt)
    cut, seg = inputs.encode_utterance(data[0][0], tok, opt, max_seq_len=20)
    assert len(full) > 20 and len(cut) == 20 and cut[:19] == full[:19] and cut[-1] == vocab.index("[SEP]") and len(seg) == 20


def test_sentencepiece_tokenizer_xlmr_layout(tmp_path):
    """XLM-R input layout from a LOCAL sentencepiece model: <s> sys.. </s></s> hyp1 </s></s> hyp2 </s>, fairseq id offset,
    pad id 1, unknown pieces -> 3 (bert_xlnet_inputs.py:37-43 doubled separator; XLMRobertaTokenizer id convention)"""
    import sentencepiece as spm
    text = tmp_path / "corpus.txt"
    lines = open(os.path.join(GOLDEN, "valid_head.txt")).read().replace("\t<=>\t", " ").replace("[", " ").replace("]", " ")
    text.write_text(lines)
    spm.SentencePieceTrainer.train(input=str(text), model_prefix=str(tmp_path / "sp"), vocab_size=120, model_type="unigram",
                                   minloglevel=2)
    tok = inputs.SentencePieceTokenizer(str(tmp_path / "sp.model"))
    assert (tok.cls_token, tok.sep_token, tok.pad_token_id) == ("<s>", "</s>", 1) and tok.vocab_size == 122
    assert tok.tokenize("</s></s>") == ["</s>", "</s>"]
    pieces = tok.tokenize("restaurant")
    ids = tok.convert_tokens_to_ids(pieces)
    assert pieces[0].startswith("▁") and all(i >= 4 for i in ids)
    assert ids == [tok.sp.piece_to_id(p) + 1 for p in pieces]
    assert tok.convert_tokens_to_ids(["<s>", "<pad>", "</s>", "<unk>", "▁zzzzqq"]) == [0, 1, 2, 3, 3]
    opt = type("O", (), dict(pre_trained_model="xlm-roberta"))()
    seq = "[CLS] [SYS] hello there [USR] cheap food [SEP] cheap foot".split(" ")
    ids, seg = inputs.encode_utterance(seq, tok, opt)
    # reference quirk (bert_xlnet_inputs.py:40,78): the FIRST separator enters the token list as the single string
    # "</s></s>", which is not a vocabulary entry -> <unk>; the separators between hypotheses are tokenised -> </s> </s>
    toks = ["<s>"] + tok.tokenize("hello") + tok.tokenize("there") + ["</s></s>"] + tok.tokenize("cheap") + tok.tokenize("food") + \
        ["</s>", "</s>"] + tok.tokenize("cheap") + tok.tokenize("foot") + ["</s>"]
    assert ids == tok.convert_tokens_to_ids(toks) and ids[0] == 0 and ids[-1] == 2
    assert ids[1 + len(tok.tokenize("hello")) + len(tok.tokenize("there"))] == 3
    n_a = 1 + len(tok.tokenize("hello")) + len(tok.tokenize("there"))
    assert seg == [0] * n_a + [1] * (len(ids) - n_a)
    batch_ids, _, lens = inputs.collate([(ids, seg), (ids[:5], seg[:5])], tok.pad_token_id)
    assert batch_ids[1, 5:].eq(1).all() and lens == [len(ids), 5]


def test_xlmr_input_builder_matches_reference_fixture():
    """ids / segment ids / lengths the REFERENCE's prepare_inputs_for_roberta produced (make_golden.py xlmr) for the first 8
    valid lines in xlm-roberta mode, with and without --without_system_act, over the committed tiny sentencepiece model"""
    d = json.load(open(os.path.join(GOLDEN, "xlmr_inputs.json")))
    tok = inputs.SentencePieceTokenizer(os.path.join(GOLDEN, "sp_tiny.model"))
    raw = [r.split(" ") for r in d["raw"]]
    for name, no_sys in (("default", False), ("without_system_act", True)):
        opt = type("O", (), dict(pre_trained_model="xlm-roberta", tod_pre_trained_model=None, without_system_act=no_sys))()
        ids, seg, lens = inputs.prepare_inputs_for_roberta(raw, tok, opt, "cpu")
        assert ids.tolist() == d[name]["ids"] and lens == d[name]["lens"]
        assert (seg is None and d[name]["seg"] is None) or seg.tolist() == d[name]["seg"]


def test_tod_mode_input_builder_matches_reference_fixture():
    """--tod_pre_trained_model keeps the [SYS] / [USR] markers (bert_xlnet_inputs.py:30-35, 55-65): ids / segment ids /
    lengths the reference's builder produced (make_golden.py tod)"""
    from nbest_amd import trainer
    d = json.load(open(os.path.join(GOLDEN, "tod_inputs.json")))
    tok = inputs.WordPieceTokenizer(json.load(open(os.path.join(GOLDEN, "text_vocab.json"))))
    data = trainer.read_wcn_data(os.path.join(GOLDEN, "valid_head.txt"))
    opt = type("O", (), dict(pre_trained_model="bert", tod_pre_trained_model="tod-bert", without_system_act=False))()
    for name, side in (("asr", data[0][:8]), ("trans", data[1][:8])):
        ids, seg, lens = inputs.prepare_inputs_for_roberta(side, tok, opt, "cpu")
        assert ids.tolist() == d[name]["ids"] and seg.tolist() == d[name]["seg"] and lens == d[name]["lens"]


def test_cli_flag_semantics_follow_the_reference():
    """--deviceId: 0 = automatic choice (first visible GPU), k > 0 = GPU k-1 (/root/reference/n_best_asr_bert.py:116-126);
    n_accum_steps = 4 iff --n_layers 12 (:522); choices the reference cannot run, or this build does not build, are refused
    with a message instead of failing later"""
    from nbest_amd import cli
    base = ["--dataset", "dstc2", "--dataroot", "x"]
    assert cli.parse_arguments(base + ["--deviceId", "0"]).gpu_index == 0
    assert cli.parse_arguments(base + ["--deviceId", "1"]).gpu_index == 0
    assert cli.parse_arguments(base + ["--deviceId", "3"]).gpu_index == 2
    assert cli.parse_arguments(base + ["--deviceId", "0"]).n_accum_steps == 1
    assert cli.parse_arguments(base + ["--deviceId", "0", "--n_layers", "12"]).n_accum_steps == 4
    for bad in (["--deviceId", "-1"], ["--deviceId", "0", "--pre_trained_model", "roberta"], ["--deviceId", "0", "--optim_choice", "adamw"],
                ["--deviceId", "0", "--pre_trained_model", "gpt2"]):
        with pytest.raises(SystemExit):
            cli.parse_arguments(base + bad)
