"""is trainer.train_epoch (+ eval_epoch) reproducible run to run?  Trains the 2-layer F1-trajectory model of tests/test_text_pipeline.py for two
epochs, three times from the same initialisation, in several configurations of the loop, and compares the parameters bit for bit.
python tools/loop_determinism.py"""
import json, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import nbest_amd  # noqa: F401
from nbest_amd import config as ncfg, inputs, synth, trainer
from nbest_amd.model import NBestSTCModel
from nbest_amd.optim import HipBertAdam

GOLDEN = os.path.join(ROOT, "tests", "golden")
labels = ncfg.LabelSpace.from_json(os.path.join(GOLDEN, "label_space.json"))
z = np.load(os.path.join(GOLDEN, "case_traj.npz"))
meta = json.loads(str(z["meta"]))
vocab = json.load(open(os.path.join(GOLDEN, "text_vocab.json")))
data = trainer.read_wcn_data(os.path.join(GOLDEN, "valid_512.txt"))
nt, nh = meta["n_train"], meta["n_held"]
tr = tuple(list(x[:nt]) for x in data)
he = tuple(list(x[nt:nt + nh]) for x in data)
cfg = ncfg.bert_base(num_hidden_layers=meta["L"], vocab_size=len(vocab), hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
label2idx = json.loads(str(np.load(os.path.join(GOLDEN, "case_text.npz"))["label2idx"]))
memory = dict(label2idx=label2idx, idx2label=labels.idx2label)


def run(dtype, with_eval, host_perm, epochs=2):
    m = NBestSTCModel(cfg, labels, device="cuda", compute_dtype=dtype, dropout=0.0)
    m.load_reference_state(synth.model_state(cfg, labels, seed=meta["seeds"][0]))
    opt = types.SimpleNamespace(batchSize=meta["batch"], tokenizer=inputs.WordPieceTokenizer(vocab), pre_trained_model="bert",
                                tod_pre_trained_model=None, without_system_act=False, add_l2_loss=False, add_segment_ids=True)
    opt.optimizer = HipBertAdam(m, lr=meta["lr"], bert_lr=meta["bert_lr"], warmup=0.1, t_total=meta["t_total"])
    split_tr, split_he = trainer.EncodedSplit(tr, opt, memory), trainer.EncodedSplit(he, opt, memory)
    if not host_perm:
        hb0 = split_tr.host_batch
        def hb_(idx, pin=False, stage=None):
            d = hb0(idx, pin, stage)
            d["tok_perm"] = d["ttok_perm"] = None
            return d
        split_tr.host_batch = hb_
    recs = []
    for ep in range(epochs):
        l, prf, a = trainer.train_epoch(m, split_tr, opt, memory, shuffle=False)
        recs.append(l)
        if with_eval:
            trainer.eval_epoch(m, split_he, opt, memory)
    torch.cuda.synchronize()
    return m.arena.p.clone(), recs


for dtype in (torch.float32, torch.bfloat16):
    for with_eval, host_perm in ((True, True), (False, True), (False, False)):
        runs = [run(dtype, with_eval, host_perm) for _ in range(3)]
        eq = [torch.equal(runs[0][0], r[0]) for r in runs[1:]]
        print("%s eval between epochs %-5s host-built token permutation %-5s: parameters bit-equal across 3 runs: %s   epoch losses %s" % (
            str(dtype).split(".")[-1], with_eval, host_perm, eq, [["%.6f" % x for x in r[1]] for r in runs]), flush=True)
