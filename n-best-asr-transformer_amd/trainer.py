"""Training / evaluation loops of the fine-tuning path, data-parallel over RCCL.

Host-side mirror of /root/reference/n_best_asr_bert.py:
  train_epoch (:232-294), eval_epoch (:297-388), pred_one_sample (:198-215),
  read_wcn_data / collate_fn labels (/root/reference/utils/dataset/tod_asr_util.py:43-71,114-123).

Data parallelism (new functionality; the reference is single-process): one process per GPU, every rank
holds the full arenas, the minibatch is sharded across ranks.  The reference's BCE / NLL terms are SUM
reductions (n_best_asr_bert.py:572-574), so gradients are all-reduced with SUM (not mean) and the MSE
term (a mean over B_global x H) is pre-scaled by B_local / B_global on each rank (shards may differ by one utterance).  The backward runs in layer chunks; as soon as a
chunk's gradients are complete their slice of the flat gradient arena is all-reduced asynchronously
(RCCL on its own stream over xGMI; 6 chunks of 2 layers for bert-base) while the remaining backward - ending with the embedding backward -
keeps the compute stream busy.  Per-tensor clipping + BertAdam run after the last reduce.
"""
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import hipabi as hb
from .fscore import compute_f1, update_f1
from .inputs import collate, encode_utterance, prepare_inputs_for_roberta


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
    if backend is None and os.environ.get("NBEST_DP_REHEARSAL") == "1":
        # one-GPU rehearsal of the multi-rank code path (tests): every rank on cuda:0, gloo as the transport
        backend, local = "gloo", 0
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


def _all_gather_flat(out, inp):
    """async all-gather of equal-sized tensors into one flat buffer (RCCL: the fused collective; gloo - CPU tests and the
    one-GPU rehearsal - gathers into views of it)"""
    if dist.get_backend() == "nccl":
        return dist.all_gather_into_tensor(out, inp, async_op=True)
    n = inp.shape[0]
    return dist.all_gather([out[r * n:(r + 1) * n] for r in range(dist.get_world_size())], inp, async_op=True)


class GradReducer:
    """bucketed asynchronous SUM all-reduce of slices of the flat gradient arena.

    Word-embedding table: with a 250 002-row vocabulary (XLM-R: 768 MB of fp32 gradient, 1 GB for the large model) only the
    <= B_local * (S + S_t) rows of tokens in the rank's shard are non-zero, and the table is the LAST bucket (the embedding
    backward is the last kernel), so nothing hides it.  ``sparse_word_grad`` (default: tables over 256 MB, i.e. XLM-R but not
    BERT's 94 MB) exchanges ``(row ids, row values)`` instead: every rank all-gathers its touched rows (32 x 128 tokens:
    <= 12.6 MB per rank instead of 768 MB through the all-reduce) and rebuilds the summed table locally - own rows zeroed,
    then every rank's rows added in rank order, so all replicas perform the same additions in the same order and stay
    bit-identical.  Position / token-type tables and the embedding LayerNorm stay in a (small) dense bucket."""

    def __init__(self, arena, n_chunks=6, sparse_word_grad=None, owner_ranges=None):
        """``owner_ranges`` (optim.HipBertAdam(shard=True).owner_ranges): per rank, the arena ranges it owns under the sharded
        optimizer - every bucket then travels as a REDUCE of its intersection with each owner's range to that owner (half the bytes
        of the all-reduce; the non-owners never read those gradients) instead of a SUM all-reduce."""
        self.arena = arena
        self.owner_ranges = owner_ranges
        self.rank, self.world = dist_info()
        L = len(arena.layer_range)
        n_chunks = max(1, min(n_chunks, L))
        edges = [round(i * L / n_chunks) for i in range(n_chunks + 1)]
        self.chunks = [(edges[i], edges[i + 1]) for i in range(n_chunks) if edges[i + 1] > edges[i]]
        self.pending = []
        self.pending_emb = []
        ws = arena.by_name["bert_encoder.embeddings.word_embeddings.weight"]
        self.word = (ws.offset, ws.offset + ws.numel, ws.shape)
        self.sparse = (ws.numel * 4 > (256 << 20)) if sparse_word_grad is None else bool(sparse_word_grad)
        self._tok = None          # (unique row ids of this rank's shard, async count exchange) of the running step
        self._sparse = None
        self._side = None         # side stream + pinned buffer that bring the per-rank row counts to the host
        self._cnt_host = None

    # ---- sparse word-embedding rows ------------------------------------------------------------------------------------
    def set_step_tokens(self, *id_tensors, rows=None):
        """call BEFORE the step's forward: ``rows`` = sorted unique token ids of this rank's shard (a device tensor whose
        size the host already knows - EncodedSplit / bench.py build it on the host, so nothing synchronises), or the
        token-id tensors themselves (then the unique rows are computed on the device, which costs one host
        synchronisation at the start of the step).  The per-rank row counts are exchanged asynchronously right away."""
        if not (self.sparse and self.world > 1):
            return
        dev = self.arena.device
        if rows is None:
            toks = [t.reshape(-1) for t in id_tensors if t is not None and t.numel()]
            rows = torch.unique(torch.cat(toks)) if toks else torch.empty(0, dtype=torch.long, device=dev)
        n = torch.tensor([rows.numel()], dtype=torch.long, device=dev)
        counts = torch.zeros(self.world, dtype=torch.long, device=dev)
        work = _all_gather_flat(counts, n)
        # The host needs the counts (they size the row exchange) - but must not wait for the COMPUTE stream to get them: that
        # would drain the whole backward before the row all-gathers, the optimizer and the next step are enqueued (round 2 did a
        # counts.tolist() there).  So the counts go to pinned host memory on a side stream, behind the collective only, with an
        # event; by the time the embedding backward has been enqueued that event completed long ago.
        ev = None
        if dev.type == "cuda":
            if self._side is None:
                self._side = torch.cuda.Stream(device=dev)
                self._cnt_host = torch.empty(self.world, dtype=torch.long).pin_memory()
            with torch.cuda.stream(self._side):
                work.wait()                                   # orders the side stream behind the collective, not the host
                self._cnt_host.copy_(counts, non_blocking=True)
                counts.record_stream(self._side)
                ev = torch.cuda.Event()
                ev.record(self._side)
        self._tok = (rows, counts, work, ev)

    def _start_sparse(self):
        rows, counts, work, ev = self._tok
        if ev is not None:
            ev.synchronize()                                  # the side-stream copy only; the compute stream keeps running
            cnt = self._cnt_host.tolist()
        else:                                                 # CPU tensors (gloo tests): the collective completes on the host
            work.wait()
            cnt = counts.tolist()
        cap = max(max(cnt), 1)
        lo, hi, shape = self.word
        G = self.arena.g[lo:hi].view(shape)
        dev = G.device
        if dev.type == "cuda":                                # library kernels (nbest_rows_gather): ids / values with padding in one launch
            ids, vals = hb.rows_gather(G, rows, cap)
        else:                                                 # gloo rehearsal on CPU tensors (tests/test_dp_gloo.py)
            ids = torch.full((cap,), -1, dtype=torch.long, device=dev)
            vals = torch.zeros(cap, shape[1], dtype=G.dtype, device=dev)
            if rows.numel():
                ids[:rows.numel()] = rows
                vals[:rows.numel()] = G.index_select(0, rows)
        all_ids = torch.empty(self.world * cap, dtype=torch.long, device=dev)
        all_vals = torch.empty(self.world * cap, shape[1], dtype=G.dtype, device=dev)
        w1 = _all_gather_flat(all_ids, ids)
        w2 = _all_gather_flat(all_vals, vals)
        self._sparse = (w1, w2, rows, cnt, cap, all_ids, all_vals, G)
        self._tok = None

    def _finish_sparse(self):
        w1, w2, rows, cnt, cap, all_ids, all_vals, G = self._sparse
        w1.wait()
        w2.wait()
        on_gpu = G.device.type == "cuda"
        if rows.numel():
            hb.rows_zero(G, rows) if on_gpu else G.index_fill_(0, rows, 0.0)
        for r in range(self.world):                                     # the same additions in the same order on every rank
            if cnt[r]:
                i, v = all_ids[r * cap:r * cap + cnt[r]], all_vals[r * cap:r * cap + cnt[r]]
                hb.rows_add(G, i, v) if on_gpu else G.index_add_(0, i, v)      # ids unique within a rank's block: plain += , no atomics
        self._sparse = None

    # ---- dense buckets -------------------------------------------------------------------------------------------------
    def _launch(self, lo, hi, last=False):
        if not (dist.is_available() and dist.is_initialized() and hi > lo):
            return
        todo = self.pending_emb if last else self.pending
        if self.owner_ranges is None:
            todo.append(dist.all_reduce(self.arena.g[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
            return
        for r, ranges in enumerate(self.owner_ranges):          # the same sequence of collectives on every rank
            for olo, ohi in ranges:
                a, b = max(lo, olo), min(hi, ohi)
                if b > a:
                    todo.append(dist.reduce(self.arena.g[a:b], dst=r, op=dist.ReduceOp.SUM, async_op=True))

    def layers_ready(self, l_lo, l_hi):
        if not self.pending and not self.pending_emb:
            self._launch(*self.arena.heads_range)    # head gradients were complete before the encoder backward began
        self._launch(self.arena.layer_range[l_lo][0], self.arena.layer_range[l_hi - 1][1])
        if l_lo == 0:                                # embedding backward is the last kernel of the chunk
            if self.sparse and self.world > 1:
                if self._tok is None:
                    raise RuntimeError("GradReducer: sparse word-embedding exchange needs set_step_tokens() before the step")
                self._launch(self.word[1], self.arena.emb_range[1], last=True)
                self._start_sparse()
            else:
                self._launch(*self.arena.emb_range, last=True)

    def wait_layers(self):
        """heads + encoder layers exchanged (the embedding tables may still be in flight)"""
        for w in self.pending:
            w.wait()
        self.pending = []

    def wait(self):
        self.wait_layers()
        for w in self.pending_emb:
            w.wait()
        self.pending_emb = []
        if self._sparse is not None:
            self._finish_sparse()

    def reduce_all(self):
        """exchange the whole gradient arena as it stands, in the bucket order of an overlapped step (used where no backward
        pass is running to overlap with: gradient-accumulation groups, ranks with an empty slice)"""
        for lo, hi in reversed(self.chunks):
            self.layers_ready(lo, hi)
        self.wait()

    def contribute_nothing(self):
        """a rank whose slice of a (short, final) batch is empty still joins every collective of the step, in the same
        order as the ranks that ran a backward pass, with zero gradients"""
        self.arena.g.zero_()
        self.set_step_tokens()
        self.reduce_all()


def train_step(model, optimizer, batch, add_l2_loss=False, add_segment_ids=True, reducer=None, global_batch=None):
    """One optimisation step on this rank's shard.  batch: dict(ids, seg, labels[, tids, tseg]) device tensors.
    ``global_batch`` = utterances of the whole minibatch over all ranks (default: world x this shard).
    Returns the step outputs (device tensors; no host synchronisation)."""
    _, world = dist_info()
    seg = batch.get("seg") if add_segment_ids else None          # n_best_asr_bert.py:252
    chunks = reducer.chunks if reducer is not None else None
    b_local = batch["ids"].shape[0]
    # MSE is a MEAN over B_global x H (n_best_asr_bert.py:574): the local kernel differentiates the mean over its own
    # B_local rows, so after the SUM all-reduce the term needs the weight B_local / B_global (= 1/world for equal shards)
    mse_scale = b_local / float(global_batch) if global_batch else 1.0 / world
    if reducer is not None:
        reducer.set_step_tokens(batch["ids"], batch.get("tids") if add_l2_loss else None, rows=batch.get("word_rows"))
    out = model.forward_backward(batch["ids"], batch["labels"], seg_ids=seg, trans_input_ids=batch.get("tids"),
                                 trans_seg_ids=batch.get("tseg"), add_l2_loss=add_l2_loss, mse_grad_scale=mse_scale,
                                 chunks=chunks, on_chunk_done=reducer.layers_ready if reducer is not None else None,
                                 tok_perm=batch.get("tok_perm"), trans_tok_perm=batch.get("ttok_perm"))
    if reducer is not None:
        reducer.wait_layers()
        optimizer.step_main()             # runs while the embedding tables' all-reduce is still in flight
        reducer.wait()
        optimizer.step_embeddings()
    else:
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


def filter_informative(labels, ontology):
    """n_best_asr_bert.py:218-229: keep act-slot-value labels only when the slot is ``this`` or an informable slot
    with more than one value; labels of any other arity pass through"""
    keep = []
    for lbl in labels:
        tup = lbl.split("-")
        if len(tup) != 3 or tup[1] == "this" or (tup[1] in ontology["informable"] and len(ontology["informable"][tup[1]]) > 1):
            keep.append(lbl)
    return keep


def _count_metrics(pred, raw_labels, idx2label, counts, ontology=None):
    """pred_one_sample + update_f1 + exact-match bookkeeping (n_best_asr_bert.py:198-215,283-288) from decoded index rows"""
    TP, FP, FN, corr, tot = counts
    all_preds = []
    for row, gold in zip(pred, raw_labels):
        pc = pred_labels_from_indices(row, idx2label)
        if ontology is not None:                                          # eval only (:341-344)
            pc, gold = filter_informative(pc, ontology), filter_informative(gold, ontology)
        TP, FP, FN = update_f1(pc, gold, TP, FP, FN)
        tot += 1
        corr += int(set(pc) == set(gold))
        all_preds.append(pc)
    return (TP, FP, FN, corr, tot), all_preds


def _host_metrics(model, out, raw_labels, idx2label, counts, ontology=None):
    """synchronous form (drains the compute stream): kept for callers outside the epoch loops"""
    return _count_metrics(model.decode(out["top"], out["bott"]).cpu().tolist(), raw_labels, idx2label, counts, ontology)


class MetricsPipe:
    """The per-sample decode of the reference (n_best_asr_bert.py:283-288) WITHOUT a host synchronisation per step.  Round 3 did
    ``model.decode(...).cpu().tolist()`` after every step, which drains the compute stream - the GPU then idles while Python
    prepares the next step.  Here the decode kernel of step i writes its int32 [B, 30] rows STRAIGHT into one of two pinned host
    buffers (mapped host memory: no copy command), a one-thread kernel behind it stamps the step number into a pinned word
    (nbest_stream_stamp), and the host turns the rows into labels / F1 counts one step LATER, after polling that word - while
    step i + 1 runs.  No hipMemcpy, no event, no stream synchronisation.  (tools/real_variants.py times this form against a
    non_blocking D2H copy + event wait: once the host thread pool is capped - limit_host_threads, the actual cause of the 37 ms
    steps first measured - both keep the GPU busy, 20.4 vs 20.5 ms per step; this one issues no runtime call besides the two
    launches.)  ``finish()`` drains the last one.  Same counts, same order."""

    def __init__(self, model, idx2label, ontology=None):
        self.model, self.idx2label, self.ontology = model, idx2label, ontology
        self.counts, self.preds = (0, 0, 0, 0, 0), []
        self.cuda = model.device.type == "cuda"
        self.bufs, self.turn, self.seq = [None, None], 0, 0
        self.flags = torch.zeros(2, dtype=torch.int32).pin_memory() if self.cuda else None
        self.pending = None

    def _host_rows(self, B, n_top):
        """one of two alternating pinned buffers, grown when a larger batch comes (never reallocated per step)"""
        t = self.turn = self.turn ^ 1
        if self.bufs[t] is None or self.bufs[t].shape[0] < B or self.bufs[t].shape[1] != n_top:
            self.bufs[t] = torch.empty(B, n_top, dtype=torch.int32).pin_memory()
        return t, self.bufs[t]

    def push(self, out, raw_labels, tag=None):
        if not self.cuda:
            self._consume((self.model.decode(out["top"], out["bott"]), None, raw_labels, tag))
            return
        # (buffer t held step i-2, whose rows were counted at the previous push)
        t, rows = self._host_rows(out["top"].shape[0], out["top"].shape[1])
        pred = self.model.decode(out["top"], out["bott"], out=rows)
        self.seq += 1
        hb.stream_stamp(self.flags[t:t + 1], self.seq)
        prev, self.pending = self.pending, (pred, (t, self.seq), raw_labels, tag)
        if prev is not None:
            self._consume(prev)

    def _consume(self, item):
        host, stamp, raw_labels, tag = item
        if stamp is not None:
            t, seq = stamp
            flag, t0 = self.flags.numpy(), time.time()
            while flag[t] != seq:              # step i-1's rows: the compute stream already holds step i
                time.sleep(0.0002)
                if time.time() - t0 > 600.0:
                    raise RuntimeError("nbest_amd: the decode of a step did not arrive within 10 minutes")
        self.counts, preds = _count_metrics(host.tolist(), raw_labels, self.idx2label, self.counts, self.ontology)
        self.preds.append((tag, preds))

    def finish(self):
        if self.pending is not None:
            self._consume(self.pending)
            self.pending = None
        return self.counts, self.preds


class EncodedSplit:
    """A data split tokenised ONCE.

    The reference re-tokenises every word of every batch of every epoch in a Python loop
    (bert_xlnet_inputs.py:46-53); at the step rates of the HIP path that loop is the bottleneck of real-data runs.
    Here every utterance is converted to id / segment lists when the split is first used; a batch is then only a
    gather + right-pad into pinned host buffers (inputs.collate).  Unpacks like the raw ``(asr, trans, labels)`` tuple.
    """

    def __init__(self, data, opt, memory):
        self.asr, self.trans, self.labels = data
        tok = opt.tokenizer
        nb, msl = getattr(opt, "n_best", None), getattr(opt, "max_seq_len", None)
        as_np = lambda r: (np.asarray(r[0], dtype=np.int64), None if r[1] is None else np.asarray(r[1], dtype=np.int64))
        self.rows = [as_np(encode_utterance(s, tok, opt, nb, msl)) for s in self.asr]
        self.trows = [as_np(encode_utterance(s, tok, opt, None, msl)) for s in self.trans]
        l2i = memory["label2idx"]
        self.y = torch.zeros(len(self.labels), len(l2i))
        for i, ls in enumerate(self.labels):
            for l in ls:
                self.y[i, l2i.get(l, 1)] = 1                      # unknown label -> index 1, as labels_to_multihot
        self.pad = tok.pad_token_id

    def __iter__(self):
        return iter((self.asr, self.trans, self.labels))

    def __len__(self):
        return len(self.asr)

    def host_batch(self, idx, pin=False, stage=None):
        """``stage``: a PinnedStage whose buffers receive every tensor of the batch (no allocation per batch)"""
        take = (lambda name: (lambda which, shape: stage.take(name + str(which), shape, torch.int64))) if stage is not None else (lambda name: None)
        ids, seg, _ = collate([self.rows[j] for j in idx], self.pad, pin and stage is None, alloc=take("a"))
        tids, tseg, _ = collate([self.trows[j] for j in idx], self.pad, pin and stage is None, alloc=take("t"))
        # rows of the word-embedding table this batch touches (sparse gradient exchange under data parallelism)
        rows = np.unique(np.concatenate([ids.numpy().ravel(), tids.numpy().ravel()]))
        # tokens sorted by word id, ties in token order: what the deterministic embedding backward reduces over (nbest_embed_ln_bwd)
        perm, tperm = token_perm(ids, as_numpy=True), token_perm(tids, as_numpy=True)
        if stage is None:
            p_ = (lambda t: t.pin_memory()) if pin else (lambda t: t)
            y = p_(self.y[torch.as_tensor(idx, dtype=torch.long)])
            rows, perm, tperm = p_(torch.from_numpy(rows)), p_(torch.from_numpy(perm)), p_(torch.from_numpy(tperm))
        else:
            y = stage.take("y", (len(idx), self.y.shape[1]), self.y.dtype)
            np.take(self.y.numpy(), np.asarray(idx, dtype=np.int64), axis=0, out=y.numpy())   # (numpy: no torch CPU thread pool in the worker)
            rows, perm, tperm = stage.put("rows", rows, torch.int64), stage.put("perm", perm, torch.int32), stage.put("tperm", tperm, torch.int32)
        return dict(ids=ids, seg=seg, tids=tids, tseg=tseg, labels=y, word_rows=rows, tok_perm=perm, ttok_perm=tperm)


_ITEMSIZE = {torch.int64: 8, torch.int32: 4, torch.float32: 4, torch.uint8: 1}


def host_cpu_budget():
    """CPUs this process may really use: the smaller of its affinity mask and its cgroup quota (cpu.max / cfs_quota_us)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        if os.path.exists("/sys/fs/cgroup/cpu.max"):
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(quota) // int(period)))
        elif os.path.exists("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
    except (OSError, ValueError):
        pass
    return n


def limit_host_threads():
    """torch sizes its CPU thread pool by the MACHINE's cores (128 on the GPU box) while the container's quota is 16: every small
    CPU op of the loop (an index_select of the label rows: 5.6 ms instead of 0.05) wakes 128 spinning threads, the cgroup is
    throttled and the whole process - the thread that launches kernels included - is frozen for the rest of the scheduling
    period.  Measured on the box (tools/host_batch_time.py): a prefetched batch cost 6.8 ms of host time at 128 threads, 0.9 ms
    at one; inside the training loop 23-47 ms, more than a step.  The HIP path needs no CPU parallelism: cap the pool at the
    smaller of 4 and the quota.  Called by the epoch loops' Prefetcher (both of its threads), cli.py and bench.py."""
    want = max(1, min(4, host_cpu_budget()))
    if torch.get_num_threads() > want:
        torch.set_num_threads(want)


class PinnedStage:
    """Long-lived pinned host buffers for ONE batch in flight, by name, grown geometrically when a larger batch comes.
    ``take(name, shape, dtype)`` -> a tensor view to fill; ``put(name, array, dtype)`` copies a numpy array in."""

    def __init__(self, pin=True):
        self.pin, self.buf = pin, {}

    def take(self, name, shape, dtype):
        n = int(np.prod(shape)) * _ITEMSIZE[dtype]
        t = self.buf.get(name)
        if t is None or t.numel() < n:
            t = torch.empty(max(n, 2 * (0 if t is None else t.numel()), 4096), dtype=torch.uint8)
            t = self.buf[name] = t.pin_memory() if self.pin else t
        return t[:n].view(dtype).view(*shape)

    def put(self, name, array, dtype):
        t = self.take(name, array.shape, dtype)
        t.numpy()[...] = array
        return t


def token_perm(ids, as_numpy=False):
    """host: int32 [B*S] token indices sorted by word id, ties in ascending token index (numpy's stable argsort)"""
    p = np.argsort(ids.reshape(-1).numpy(), kind="stable").astype(np.int32)
    return p if as_numpy else torch.from_numpy(p)


def encoded(data, opt, memory):
    return data if isinstance(data, EncodedSplit) else EncodedSplit(data, opt, memory)


def batch_indices(n, batch_size, shuffle=False, seed=0):
    order = np.arange(n)
    if shuffle:
        np.random.default_rng(seed).shuffle(order)
    return [order[i:i + batch_size].tolist() for i in range(0, n, batch_size)]


class Prefetcher:
    """Iterates ``(batch_index, dataset indices of this rank's slice, device batch)`` one batch ahead of the consumer.

    A worker thread collates batch i+1 into pinned host memory and enqueues its H2D copies on a side stream while the
    GPU runs step i; the consumer's stream waits on the copy event only (no host synchronisation)."""

    def __init__(self, split, index_lists, device, rank=0, world=1, depth=2):
        import queue
        import threading
        self.split, self.lists, self.device, self.rank, self.world = split, index_lists, torch.device(device), rank, world
        self.cuda = self.device.type == "cuda"
        self.q = queue.Queue(maxsize=depth)
        self.stream = torch.cuda.Stream(self.device) if self.cuda else None
        self.seconds = [0.0, 0.0, 0.0]
        limit_host_threads()
        # pinned staging: one PinnedStage per batch that can be alive at once (depth in the queue + the one the consumer holds + the one
        # being built + one spare), reused round-robin once the H2D copies that read it have completed
        self.stages = [(PinnedStage(), [None]) for _ in range(depth + 3)] if self.cuda else None
        self.thread = threading.Thread(target=self._work, daemon=True)
        self.thread.start()

    def _work(self):
        try:
            limit_host_threads()                                     # (OpenMP's thread count is a per-thread setting)
            for bi, idx in enumerate(self.lists):
                lo, hi = shard_bounds(len(idx), self.rank, self.world)
                if hi <= lo:
                    self.q.put((bi, [], None, None))                 # nothing for this rank in this batch
                    continue
                mine = idx[lo:hi]
                t0 = time.time()
                if not self.cuda:
                    self.q.put((bi, mine, self.split.host_batch(mine), None))
                    continue
                stage, last = self.stages[bi % len(self.stages)]
                if last[0] is not None:
                    last[0].synchronize()                             # the copies out of this stage, depth + 3 batches ago
                host = self.split.host_batch(mine, stage=stage)
                t1 = time.time()
                with torch.cuda.stream(self.stream):
                    dev = {k: (None if v is None else v.to(self.device, non_blocking=True)) for k, v in host.items()}
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
                last[0] = ev
                t2 = time.time()
                self.q.put((bi, mine, dev, (ev, host)))
                self.seconds[0] += t1 - t0                           # (worker-side clock: collate | H2D enqueue | queue full)
                self.seconds[1] += t2 - t1
                self.seconds[2] += time.time() - t2
            self.q.put(None)
        except BaseException as e:                                   # surface worker failures in the consumer
            self.q.put(e)

    def __iter__(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            bi, mine, dev, sync = item
            if sync is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(sync[0])
                for v in dev.values():
                    if v is not None:
                        v.record_stream(cur)
            yield bi, mine, dev


def _prep(raw_in, raw_trans, raw_labels, opt, memory, device):
    ids, seg, _ = prepare_inputs_for_roberta(raw_in, opt.tokenizer, opt, device, n_best=getattr(opt, "n_best", None),
                                             max_seq_len=getattr(opt, "max_seq_len", None))
    tids, tseg, _ = prepare_inputs_for_roberta(raw_trans, opt.tokenizer, opt, device, max_seq_len=getattr(opt, "max_seq_len", None))
    y = labels_to_multihot(raw_labels, memory["label2idx"], device)
    return dict(ids=ids, seg=seg, tids=tids, tseg=tseg, labels=y)


def train_epoch(model, data, opt, memory, epoch=0, shuffle=True):
    """n_best_asr_bert.py:232-294 -> (mean_loss, (p, r, f), acc).  ``data`` = (asr, trans, labels) lists or an
    EncodedSplit (tokenised once; pass the same object every epoch to reuse it).

    ``opt.n_accum_steps`` > 1 (n_best_asr_bert.py:522,526,266-280): the loader batch is batchSize / n_accum_steps, gradients
    of n_accum_steps consecutive micro-batches are summed and the optimizer steps after every n_accum_steps-th one; micro-
    batches left over at the end of the epoch are dropped by the next epoch's zero_grad, as there.  Under data parallelism
    only the micro-batch that closes a group exchanges gradients."""
    model.train()
    rank, world = dist_info()
    reducer = (GradReducer(model.arena, owner_ranges=getattr(opt.optimizer, "owner_ranges", None))
               if (dist.is_available() and dist.is_initialized()) else None)
    losses = []
    pipe = MetricsPipe(model, memory["idx2label"])
    split = encoded(data, opt, memory)
    n_accum = max(1, int(getattr(opt, "n_accum_steps", 1) or 1))
    lists = batch_indices(len(split), max(1, int(opt.batchSize / n_accum)), shuffle=shuffle, seed=getattr(opt, "random_seed", 999) + epoch)
    group_rows = []                                                  # word-table rows touched by the running accumulation group
    for bi, mine, b in Prefetcher(split, lists, model.device, rank, world):
        first, last = (bi % n_accum == 0), ((bi + 1) % n_accum == 0)
        if first:
            group_rows = []
        if not mine:
            if first:
                model.arena.g.zero_()
            if last:
                if reducer is not None:
                    reducer.set_step_tokens(*group_rows)
                    reducer.reduce_all()
                opt.optimizer.step()                                 # keeps replicas and schedule positions identical
            model.step_counter += 1
            continue
        if n_accum == 1:
            out = train_step(model, opt.optimizer, b, add_l2_loss=opt.add_l2_loss, add_segment_ids=opt.add_segment_ids, reducer=reducer,
                             global_batch=len(lists[bi]))
        else:
            seg = b.get("seg") if opt.add_segment_ids else None
            out = model.forward_backward(b["ids"], b["labels"], seg_ids=seg, trans_input_ids=b.get("tids"), trans_seg_ids=b.get("tseg"),
                                         add_l2_loss=opt.add_l2_loss, mse_grad_scale=len(mine) / float(len(lists[bi])),
                                         accumulate=not first, tok_perm=b.get("tok_perm"), trans_tok_perm=b.get("ttok_perm"))
            group_rows.append(b["word_rows"])
            if last:
                if reducer is not None:
                    reducer.set_step_tokens(*group_rows)
                    reducer.reduce_all()
                opt.optimizer.step()
        losses.append((out["loss_parts"], len(mine), len(lists[bi])))
        pipe.push(out, [split.labels[j] for j in mine])
    counts, _ = pipe.finish()
    return _finish(losses, counts, model.device, len(lists))


def merge_cases(chunks):
    """[(batch_index, rank, cases)] of this rank -> all ranks' cases in dataset order (batch, then rank slice)"""
    _, world = dist_info()
    if world > 1:
        every = [None] * world
        dist.all_gather_object(every, chunks)
        chunks = sorted((c for part in every for c in part), key=lambda c: (c[0], c[1]))
    return [case for _, _, part in chunks for case in part]


@torch.no_grad()
def eval_epoch(model, data, opt, memory, fp=None, efp=None):
    """n_best_asr_bert.py:297-388 -> (mean_loss, (p, r, f), acc, cases); writes ``raw <=> pred <=> gold`` lines.

    Under data parallelism every rank evaluates its slice of each batch; the per-utterance cases are gathered and put
    back into dataset order, so ``cases`` and the lines written are the same on every rank as in a single process.
    """
    model.eval()
    rank, world = dist_info()
    onto = getattr(opt, "ontology", None)
    losses, chunks = [], []
    pipe = MetricsPipe(model, memory["idx2label"], onto)
    split = encoded(data, opt, memory)
    # the reference builds its valid / test loaders with int(batchSize / n_accum_steps) too (n_best_asr_bert.py:529-531), and the
    # record is the mean over batches of sum / batch size: the partition (and the weight of a short last batch) must be the same
    n_accum = max(1, int(getattr(opt, "n_accum_steps", 1) or 1))
    lists = batch_indices(len(split), max(1, int(opt.batchSize / n_accum)))
    for bi, mine, b in Prefetcher(split, lists, model.device, rank, world):
        if not mine:
            continue
        seg = b["seg"] if opt.add_segment_ids else None
        out = model.forward_backward(b["ids"], b["labels"], seg_ids=seg, need_grad=False)     # no MSE in eval (:331)
        losses.append((out["loss_parts"], len(mine), len(lists[bi])))
        pipe.push(out, [split.labels[j] for j in mine], tag=(bi, mine))
    counts, tagged = pipe.finish()
    for (bi, mine), preds in tagged:
        raw_labels = [split.labels[j] for j in mine]
        golds = [filter_informative(g, onto) if onto is not None else g for g in raw_labels]
        chunks.append((bi, rank, list(zip([split.asr[j] for j in mine], preds, golds))))
    cases = merge_cases(chunks)
    for raw, pc, gold in cases:
        line = "%s\t<=>\t%s\t<=>\t%s\n" % (" ".join(raw), ";".join(pc), ";".join(gold))
        if fp is not None:
            fp.write(line)
        if efp is not None and set(pc) != set(gold):
            efp.write(line)
    return _finish(losses, counts, model.device, len(lists)) + (cases,)


def _finish(losses, counts, device, n_batches=None):
    """losses: [(loss_parts[4] = BCE(final), BCE(top), mean-CE, MSE of this rank's shard, B_local, B_global)] per batch.
    loss_record of a batch = sum(parts) / batch_size (n_best_asr_bert.py:168-192), mean over batches (:290).  The BCE / CE
    parts are sums over utterances, so a rank contributes its sums / B_global; the MSE part is a mean over the rank's own
    rows and enters with the weight B_local / B_global - summed over ranks this is the single-process record even when the
    shards are uneven.  ``n_batches``: batches of the epoch (the same on every rank; a rank whose shard of a batch was
    empty has no entry for it)."""
    rec = 0.0
    for lp, b_local, b_global in losses:
        lp = lp.double().cpu()
        rec += (float(lp[:3].sum()) + float(lp[3]) * b_local / b_global) / b_global
    if n_batches is None:
        n_batches = len(losses)
    stat = torch.tensor(list(counts) + [rec], dtype=torch.float64, device=device)
    _, world = dist_info()
    if world > 1:
        dist.all_reduce(stat, op=dist.ReduceOp.SUM)
    TP, FP, FN, corr, tot, lsum = stat.tolist()
    p, r, f = compute_f1(int(TP), int(FP), int(FN))
    acc = corr / tot * 100 if tot else 0
    return lsum / max(n_batches, 1), (p, r, f), acc
