"""Where does the real-data epoch loop lose time?  Runs the same 16-step epoch with the per-step prediction hand-off in several forms
(0 none, 1 decode only, 2 + D2H copy and event, 3 + event wait, 4 + host counting, 5 event polled, 6 trainer.MetricsPipe as shipped)
and prints wall ms per step, the host segments and the GPU busy / idle time per step of each (tools/README.md).  Usage: python tools/real_variants.py [n_utterances]"""
import json, os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import nbest_amd  # noqa: F401
from nbest_amd import config as ncfg, inputs, synth, trainer
from nbest_amd.model import NBestSTCModel
from nbest_amd.optim import HipBertAdam

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
labels = ncfg.LabelSpace.from_json(os.path.join(ROOT, "tests", "golden", "label_space.json"))
cfg = ncfg.bert_base()
m = NBestSTCModel(cfg, labels, device="cuda", compute_dtype=torch.bfloat16, dropout=0.3, seed=999)
m.load_reference_state(synth.model_state(cfg, labels, seed=999))
m.train()
optim = HipBertAdam(m, lr=3e-5, bert_lr=3e-5, warmup=0.1, t_total=100000)
vocab = json.load(open(os.path.join(ROOT, "tests", "golden", "text_vocab.json")))
z = np.load(os.path.join(ROOT, "tests", "golden", "case_text.npz"))
memory = dict(label2idx=json.loads(str(z["label2idx"])), idx2label=labels.idx2label)
data = trainer.read_wcn_data(os.path.join(ROOT, "tests", "golden", "valid_512.txt"))
reps = (n + len(data[0]) - 1) // len(data[0])
data = tuple(list(x) * reps for x in data)
opt = types.SimpleNamespace(batchSize=256, tokenizer=inputs.WordPieceTokenizer(vocab), pre_trained_model="bert", tod_pre_trained_model=None,
                            without_system_act=False, add_l2_loss=False, add_segment_ids=True, n_best=5, max_seq_len=None, random_seed=999,
                            optimizer=optim)
split = trainer.EncodedSplit(data, opt, memory)
lists = trainer.batch_indices(len(split), 256, shuffle=True, seed=5)


def loop(variant):
    pf = trainer.Prefetcher(split, lists, m.device, 0, 1)
    pipe = trainer.MetricsPipe(m, memory["idx2label"])
    pending = None
    marks, gev, old_bufs = [], [], [None, None]
    torch.cuda.synchronize()
    t0 = time.time()
    for bi, mine, b in pf:
        ta = time.time()
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g0.record()
        out = trainer.train_step(m, optim, b, add_l2_loss=False, add_segment_ids=True, global_batch=len(lists[bi]))
        g1.record()
        gev.append((g0, g1))
        tb = time.time()
        if variant == 6:                      # the shipped pipeline: decode into pinned host rows + stamp, host polls one step later
            pipe.push(out, [split.labels[j] for j in mine])
        elif variant >= 1:
            pred = m.decode(out["top"], out["bott"])
        if 2 <= variant <= 5:                 # the rejected forms: D2H copy + event
            if old_bufs[bi & 1] is None or old_bufs[bi & 1].shape[0] < pred.shape[0]:
                old_bufs[bi & 1] = torch.empty(pred.shape, dtype=pred.dtype).pin_memory()
            host = old_bufs[bi & 1][:pred.shape[0]]
            host.copy_(pred, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
            prev, pending = pending, (host, done)
        tc = time.time()
        if variant < 2 or variant == 6:
            prev = None
        if variant in (3, 4) and prev is not None:
            prev[1].synchronize()
        if variant == 5 and prev is not None:
            while not prev[1].query():
                time.sleep(0.0003)
        td = time.time()
        if variant >= 4 and prev is not None:
            pipe.counts, _ = trainer._count_metrics(prev[0].tolist(), [split.labels[j] for j in mine], memory["idx2label"], pipe.counts)
        te = time.time()
        marks.append((ta, tb, tc, td, te))
    pipe.finish()
    torch.cuda.synchronize()
    wall = (time.time() - t0) * 1e3 / len(lists)
    seg = [sum((mk[i + 1] - mk[i]) for mk in marks) * 1e3 / len(marks) for i in range(4)]
    gap = sum(marks[i + 1][0] - marks[i][4] for i in range(len(marks) - 1)) * 1e3 / (len(marks) - 1)
    busy = sum(a.elapsed_time(b) for a, b in gev) / len(gev)
    idle = sum(gev[i][1].elapsed_time(gev[i + 1][0]) for i in range(len(gev) - 1)) / (len(gev) - 1)
    print("           prefetch worker per batch: collate %.1f ms, H2D enqueue %.1f ms, blocked on a full queue %.1f ms" % tuple(
        1e3 * x / len(lists) for x in pf.seconds))
    print("           GPU: step %.1f ms, idle between steps %.1f ms" % (busy, idle))
    print("variant %d: %.1f ms/step wall | host: train_step %.1f, decode+copy %.1f, event wait %.1f, count %.1f, next batch %.1f" % (
        variant, wall, seg[0], seg[1], seg[2], seg[3], gap), flush=True)


loop(0)
for v in (0, 6, 4, 6, 0, 6):
    print('--')
    loop(v)
