#!/usr/bin/env python3
"""Where does the HIP bf16 path differ from the bf16-storage oracle?  For one golden case: fp32 oracle gradients (CPU),
bf16-storage oracle gradients (CPU) and HIP bf16 gradients; per tensor the noise-to-signal ||g - g_ref|| / ||g_ref|| of
both legs and the signed norm error.  (tests/ infrastructure: uses oracle/.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_case, case_inputs
import nbest_amd  # noqa
from nbest_amd.config import LabelSpace
from nbest_amd.model import NBestSTCModel
from oracle.encoder import EncoderConfig
from oracle.model import OracleModel
from oracle import stc, bf16sim
labels = LabelSpace.from_json(os.path.join(ROOT, "tests", "golden", "label_space.json"))
name = sys.argv[1]
meta, z = load_case(name)
cfg, sd, batch = case_inputs(meta, labels)
ocfg = EncoderConfig(**{k: v for k, v in cfg.to_dict().items() if k in EncoderConfig.__dataclass_fields__})
om = OracleModel(ocfg, labels.top2bottom, labels.n_bottom, 0.0)
om.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); om.train()
t = {k: torch.from_numpy(v) for k, v in batch.items()}
seg = t["seg"] if meta["seg"] else None
b2t = stc.bottom2top_matrix(labels.top2bottom)
torch.set_num_threads(16)
top, bott, fin, asr, tr = om(t["ids"], t["tids"], seg_ids=seg, trans_seg_ids=t["tseg"])
_, total, _ = stc.total_loss(top, bott, fin, t["labels"], labels.top2bottom, b2t, asr, tr, meta["add_l2"])
total.backward()
gref = {n: p.grad.clone() for n, p in om.named_parameters() if p.grad is not None}
for p in om.parameters(): p.grad = None
stop, sbot, sfin, sasr, str_ = bf16sim.forward(om, t["ids"], t["tids"], seg_ids=seg, trans_seg_ids=t["tseg"])
_, stotal, _ = stc.total_loss(stop, sbot, sfin, t["labels"], labels.top2bottom, b2t, sasr, str_, meta["add_l2"])
stotal.backward()
gsim = {n: p.grad.clone() for n, p in om.named_parameters() if p.grad is not None}
m = NBestSTCModel(cfg, labels, device="cuda", compute_dtype=torch.bfloat16, dropout=0.0, seed=1)
m.load_reference_state(sd); m.train()
b = {k: v.cuda() for k, v in t.items()}
out = m.forward_backward(b["ids"], b["labels"], seg_ids=b["seg"] if meta["seg"] else None, trans_input_ids=b["tids"],
                         trans_seg_ids=b["tseg"], add_l2_loss=meta["add_l2"])
named = dict(m.named_parameters())
print("%-58s %10s %10s | %10s %10s" % ("tensor", "hip n/s", "sim n/s", "hip dnorm", "sim dnorm"))
for n, g in gref.items():
    if n.endswith("key.bias") or "word_emb" in n:
        continue
    gh = named[n].grad.float().cpu()
    ns = lambda a: ((a - g).norm() / g.norm()).item()
    dn = lambda a: ((a.norm() - g.norm()) / g.norm()).item()
    print("%-58s %10.3e %10.3e | %+10.3e %+10.3e" % (n[13:] if n.startswith("bert") else n, ns(gh), ns(gsim[n]), dn(gh), dn(gsim[n])))
