"""which gradient tensors of one training step differ between four runs of the same batch (fp32 and bf16, two shapes): python tools/determinism_check.py"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nbest_amd
from nbest_amd import config as ncfg, synth
from nbest_amd.model import NBestSTCModel
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
labels = ncfg.LabelSpace.from_json(os.path.join(ROOT, "tests", "golden", "label_space.json"))
for dt in (torch.float32, torch.bfloat16):
  for (B,S,L) in ((24,48,2),(64,128,2)):
    cfg = ncfg.bert_base(num_hidden_layers=L, vocab_size=3000, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd = synth.model_state(cfg, labels, seed=5)
    b = synth.nbest_batch(cfg, labels, B, S, n_best=5, seed=3, ragged=True, trans_len=16)
    t = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
    m = NBestSTCModel(cfg, labels, device="cuda", compute_dtype=dt, dropout=0.0)
    m.load_reference_state(sd); m.train()
    gs=[]; outs=[]
    for r in range(4):
        out = m.forward_backward(t["ids"], t["labels"], seg_ids=t["seg"])
        torch.cuda.synchronize()
        gs.append(m.arena.g.clone()); outs.append((out["final"].clone(), out["loss_parts"].clone()))
    bad=set()
    for r in range(1,4):
        if not torch.equal(outs[0][0], outs[r][0]): bad.add("FORWARD final")
        if not torch.equal(outs[0][1], outs[r][1]): bad.add("loss_parts")
        for s_ in m.arena.slots:
            x,y = m.arena.view(gs[0], s_.name), m.arena.view(gs[r], s_.name)
            if not torch.equal(x,y): bad.add(s_.name)
    print(dt, (B,S,L), "non-deterministic tensors:", sorted(bad)[:12], len(bad))

# ---- several optimizer steps over batches of VARYING shape (as the real-data loop sees them), twice from the same initialisation
from nbest_amd.optim import HipBertAdam
for dt in (torch.float32, torch.bfloat16):
    cfg = ncfg.bert_base(num_hidden_layers=2, vocab_size=3000, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd = synth.model_state(cfg, labels, seed=5)
    shapes = [(16, 40), (16, 52), (16, 33), (16, 61), (16, 47), (16, 52), (9, 38), (16, 64)]
    batches = []
    for i, (B, S) in enumerate(shapes):
        b = synth.nbest_batch(cfg, labels, B, S, n_best=5, seed=10 + i, ragged=True, trans_len=12)
        batches.append({k: torch.from_numpy(v).cuda() for k, v in b.items()})
    finals = []
    for rep in range(3):
        m = NBestSTCModel(cfg, labels, device="cuda", compute_dtype=dt, dropout=0.0)
        m.load_reference_state(sd); m.train()
        opt = HipBertAdam(m, lr=1e-3, bert_lr=1e-4, warmup=0.1, t_total=100)
        hist = []
        for ep in range(3):
            for t in batches:
                out = m.forward_backward(t["ids"], t["labels"], seg_ids=t["seg"])
                opt.step()
                hist.append(out["loss_parts"].clone())
        torch.cuda.synchronize()
        finals.append((m.arena.p.clone(), torch.stack(hist).cpu()))
    for rep in (1, 2):
        same_p = torch.equal(finals[0][0], finals[rep][0])
        first = next((i for i in range(finals[0][1].shape[0]) if not torch.equal(finals[0][1][i], finals[rep][1][i])), None)
        print(dt, "24 steps over varying shapes, run %d vs run 0: parameters bit-equal %s, first step whose loss differs: %s" % (rep, same_p, first))
