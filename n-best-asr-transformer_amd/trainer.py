"""Training / evaluation loops of the fine-tuning path, data-parallel over RCCL.

Host-side mirror of /root/reference/n_best_asr_bert.py:
  train_epoch (:232-294), eval_epoch (:297-388), pred_one_sample (:198-215),
  read_wcn_data / collate_fn labels (/root/reference/utils/dataset/tod_asr_util.py:43-71,114-123).

Data parallelism (new functionality; the reference is single-process): one process per GPU, every rank
holds the full arenas, the minibatch is sharded across ranks.  The reference's BCE / NLL terms are SUM
reductions (n_best_asr_bert.py:572-574), so gradients are all-reduced with SUM (not mean) and the MSE
term (a mean over B x H) is pre-scaled by 1/world.  The backward runs in layer chunks; as soon as a
chunk's gradients are complete their slice of the flat gradient arena is all-reduced asynchronously
(RCCL on its own stream over xGMI; 6 chunks of 2 layers for bert-base) while the remaining backward - ending with the embedding backward -
keeps the compute stream busy.  Per-tensor clipping + BertAdam run after the last reduce.
"""
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from .fscore import compute_f1, update_f1
from .inputs import prepare_inputs_for_roberta


# ------------------------------------------------------------------------------------------------
# distributed helpers
# ------------------------------------------------------------------------------------------------
def dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_distributed(backend=None):
    """read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun); no-op for a single process"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "RANK" not in os.environ:          # plain `python bench.py`: single process, no process group
        return 0, 1, 0
    # under torchrun a process group is created even for one rank, so the RCCL path (communicator, async
    # bucketed all-reduce, barrier) is the same code at every N and can be exercised on a 1-GPU box
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return rank, world, local


def broadcast_parameters(model):
    """identical replicas: rank 0's master arena wins"""
    if dist.is_available() and dist.is_initialized():
        dist.broadcast(model.arena.p, src=0)
        model.arena.refresh_compute_copy()


def shard_bounds(n, rank, world):
    """contiguous shard of n samples for this rank (sizes differ by at most one)"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradReducer:
    """bucketed asynchronous SUM all-reduce of slices of the flat gradient arena"""

    def __init__(self, arena, n_chunks=6):
        self.arena = arena
        self.rank, self.world = dist_info()
        L = len(arena.layer_range)
        n_chunks = max(1, min(n_chunks, L))
        edges = [round(i * L / n_chunks) for i in range(n_chunks + 1)]
        self.chunks = [(edges[i], edges[i + 1]) for i in range(n_chunks) if edges[i + 1] > edges[i]]
        self.pending = []

    def _launch(self, lo, hi):
        if dist.is_available() and dist.is_initialized() and hi > lo:
            self.pending.append(dist.all_reduce(self.arena.g[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def layers_ready(self, l_lo, l_hi):
        if not self.pending:
            self._launch(*self.arena.heads_range)    # head gradients were complete before the encoder backward began
        self._launch(self.arena.layer_range[l_lo][0], self.arena.layer_range[l_hi - 1][1])
        if l_lo == 0:
            self._launch(*self.arena.emb_range)      # embedding backward is the last kernel of the chunk

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []


def train_step(model, optimizer, batch, add_l2_loss=False, add_segment_ids=True, reducer=None):
    """One optimisation step on this rank's shard.  batch: dict(ids, seg, labels[, tids, tseg]) device tensors.
    Returns the step outputs (device tensors; no host synchronisation)."""
    _, world = dist_info()
    seg = batch.get("seg") if add_segment_ids else None          # n_best_asr_bert.py:252
    chunks = reducer.chunks if reducer is not None else None
    out = model.forward_backward(batch["ids"], batch["labels"], seg_ids=seg, trans_input_ids=batch.get("tids"),
                                 trans_seg_ids=batch.get("tseg"), add_l2_loss=add_l2_loss, mse_grad_scale=1.0 / world,
                                 chunks=chunks, on_chunk_done=reducer.layers_ready if reducer is not None else None)
    if reducer is not None:
        reducer.wait()
    optimizer.step()
    return out


# ------------------------------------------------------------------------------------------------
# data: "ASR \t<=>\t TRANSCRIPT \t<=>\t label1;label2" lines
# ------------------------------------------------------------------------------------------------
def coverage_sample(labels, coverage, seed=42):
    """Row order of the --coverage stratified subsample, /root/reference/utils/dataset/tod_asr_util.py:12-39:
    the first utterance of every distinct label set (file order), then
    rem = round(|coverage * N - n_unique|) of the remaining utterances drawn without replacement by
    pandas' DataFrame.sample(n=rem, random_state=42) = numpy RandomState(42).choice(len(rest), rem, False)."""
    seen, uniq = set(), []
    for i, l in enumerate(labels):
        t = tuple(l)
        if t not in seen:
            seen.add(t)
            uniq.append(i)
    uset = set(uniq)
    rest = [i for i in range(len(labels)) if i not in uset]
    rem = int(np.round(abs(float(coverage) * len(labels) - len(uniq))))
    pick = np.random.RandomState(seed).choice(len(rest), size=rem, replace=False)
    return uniq + [rest[int(j)] for j in pick]


def read_wcn_data(fn, coverage=None):
    """``ASR \\t<=>\\t TRANSCRIPT \\t<=>\\t label1;label2`` lines (tod_asr_util.py:43-71); coverage: see coverage_sample"""
    asr, trans, labels = [], [], []
    with open(fn) as fp:
        for line in fp:
            a, t, lbl = line.strip("\n\r").split("\t<=>\t")
            asr.append(a.strip().split(" "))
            trans.append(t.strip().split(" "))
            labels.append(lbl.strip().split(";") if lbl else [])
    if coverage:
        order = coverage_sample(labels, coverage)
        asr, trans, labels = [asr[i] for i in order], [trans[i] for i in order], [labels[i] for i in order]
    return asr, trans, labels


def labels_to_multihot(label_lists, label2idx, device, unk=1):
    y = torch.zeros(len(label_lists), len(label2idx))
    for i, ls in enumerate(label_lists):
        for l in ls:
            y[i, label2idx.get(l, unk)] = 1
    return y.to(device)


def batches(data, batch_size, shuffle=False, seed=0):
    asr, trans, labels = data
    order = np.arange(len(asr))
    if shuffle:
        np.random.default_rng(seed).shuffle(order)
    for i in range(0, len(order), batch_size):
        idx = order[i:i + batch_size]
        yield [asr[j] for j in idx], [trans[j] for j in idx], [labels[j] for j in idx]


def pred_labels_from_indices(pred_row, idx2label):
    """pred_one_sample (:198-215) from the device decode: labels in top-label order"""
    return [idx2label[j] for j in pred_row if j >= 0]


def _host_metrics(model, out, raw_labels, idx2label, counts):
    pred = model.decode(out["top"], out["bott"]).cpu().tolist()
    TP, FP, FN, corr, tot = counts
    all_preds = []
    for row, gold in zip(pred, raw_labels):
        pc = pred_labels_from_indices(row, idx2label)
        TP, FP, FN = update_f1(pc, gold, TP, FP, FN)
        tot += 1
        corr += int(set(pc) == set(gold))
        all_preds.append(pc)
    return (TP, FP, FN, corr, tot), all_preds


def _prep(raw_in, raw_trans, raw_labels, opt, memory, device):
    ids, seg, _ = prepare_inputs_for_roberta(raw_in, opt.tokenizer, opt, device, n_best=getattr(opt, "n_best", None))
    tids, tseg, _ = prepare_inputs_for_roberta(raw_trans, opt.tokenizer, opt, device)
    y = labels_to_multihot(raw_labels, memory["label2idx"], device)
    return dict(ids=ids, seg=seg, tids=tids, tseg=tseg, labels=y)


def train_epoch(model, data, opt, memory, epoch=0, shuffle=True):
    """n_best_asr_bert.py:232-294 -> (mean_loss, (p, r, f), acc).  ``data`` = (asr, trans, labels) lists."""
    model.train()
    rank, world = dist_info()
    reducer = GradReducer(model.arena) if (dist.is_available() and dist.is_initialized()) else None
    counts, losses = (0, 0, 0, 0, 0), []
    for raw_in, raw_trans, raw_labels in batches(data, opt.batchSize, shuffle=shuffle, seed=getattr(opt, "random_seed", 999) + epoch):
        lo, hi = shard_bounds(len(raw_in), rank, world)
        if hi <= lo:
            continue
        b = _prep(raw_in[lo:hi], raw_trans[lo:hi], raw_labels[lo:hi], opt, memory, model.device)
        out = train_step(model, opt.optimizer, b, add_l2_loss=opt.add_l2_loss, add_segment_ids=opt.add_segment_ids, reducer=reducer)
        losses.append((out["loss_parts"], hi - lo))
        counts, _ = _host_metrics(model, out, raw_labels[lo:hi], memory["idx2label"], counts)
    return _finish(losses, counts, model.device)


@torch.no_grad()
def eval_epoch(model, data, opt, memory, fp=None, efp=None):
    """n_best_asr_bert.py:297-388 -> (mean_loss, (p, r, f), acc, cases); writes ``raw <=> pred <=> gold`` lines."""
    model.eval()
    rank, world = dist_info()
    counts, losses, cases = (0, 0, 0, 0, 0), [], []
    for raw_in, raw_trans, raw_labels in batches(data, opt.batchSize):
        lo, hi = shard_bounds(len(raw_in), rank, world)
        if hi <= lo:
            continue
        b = _prep(raw_in[lo:hi], raw_trans[lo:hi], raw_labels[lo:hi], opt, memory, model.device)
        seg = b["seg"] if opt.add_segment_ids else None
        out = model.forward_backward(b["ids"], b["labels"], seg_ids=seg, need_grad=False)     # no MSE in eval (:331)
        losses.append((out["loss_parts"], hi - lo))
        counts, preds = _host_metrics(model, out, raw_labels[lo:hi], memory["idx2label"], counts)
        for raw, pc, gold in zip(raw_in[lo:hi], preds, raw_labels[lo:hi]):
            line = "%s\t<=>\t%s\t<=>\t%s\n" % (" ".join(raw), ";".join(pc), ";".join(gold))
            if fp is not None:
                fp.write(line)
            if efp is not None and set(pc) != set(gold):
                efp.write(line)
            cases.append((raw, pc, gold))
    return _finish(losses, counts, model.device) + (cases,)


def _finish(losses, counts, device):
    # loss_record of a batch = sum(parts) / batch_size (n_best_asr_bert.py:168-192); mean over batches (:290)
    recs = [float(lp.sum().item()) / n for lp, n in losses] if losses else [0.0]
    stat = torch.tensor(list(counts) + [sum(recs), len(recs)], dtype=torch.float64, device=device)
    _, world = dist_info()
    if world > 1:
        dist.all_reduce(stat, op=dist.ReduceOp.SUM)
    TP, FP, FN, corr, tot, lsum, ln = stat.tolist()
    p, r, f = compute_f1(int(TP), int(FP), int(FN))
    acc = corr / tot * 100 if tot else 0
    return lsum / max(ln, 1), (p, r, f), acc
