"""BertAdam over the flat parameter arena, one multi-tensor HIP launch set per step.

Interface and semantics of /root/reference/models/optimization.py:183-302 as driven by
/root/reference/n_best_asr_bert.py:540-561: each parameter tensor is its own group (lr = bert_lr for
``bert_encoder.*`` else lr; weight_decay 0.01 except bias / LayerNorm), gradient clipped PER TENSOR to
L2 norm 1.0, no bias correction, eps 1e-6, ``warmup_linear`` schedule evaluated at the step count
BEFORE the increment.
"""
import torch
import torch.distributed as dist

from . import hipabi as hb


def warmup_linear(step, t_total, warmup):
    """optimization.py:162-171 (+ :60-61: t_total < 0 disables the schedule)"""
    if t_total < 0:
        return 1.0
    x = float(step) / float(t_total)
    if x < warmup:
        return x / warmup
    return max((x - 1.0) / (warmup - 1.0), 0.0)


class HipBertAdam:
    """``shard`` (data parallel, new functionality - the reference is single-process): the optimizer SHARDED over the ranks instead
    of replicated.  Every rank owns a contiguous range of the optimizer's blocks (16 384 elements of one tensor each; about 1/world of
    the elements, separately for the encoder-layer / head tensors and for the embedding tables, which are updated after their own, last,
    gradient exchange) and with it that range of the arenas:
      * gradients travel as REDUCE to the owner (trainer.GradReducer(owner_ranges=...)) - half the bytes of the all-reduce;
      * per-tensor clip: each rank computes the block sums of squares of its own blocks into a zeroed vector, which is
        SUM-all-reduced (n_blocks floats, 27 KB for bert-base): x + 0 is exact, so every rank holds exactly the numbers one process
        computes, and the clip coefficients - hence the updates - are bit-identical to the replicated optimizer's;
      * each rank updates p, m, v and the bf16 compute copy of its own range only (1/world of the optimizer's HBM traffic);
      * the owners then broadcast what the next step reads: the bf16 compute copy of their range (2 B per parameter) and, gathered
        into one small buffer, the fp32 values of the tensors the kernels read from the master arena (biases, LayerNorm parameters,
        STC heads: 0.26 M of bert-base's 109.6 M parameters).  Bytes per parameter and step: 4 (reduce) + 2 (bf16 copy) against 8 for
        the all-reduce: 0.75 x.  With fp32 compute (the parity path) the fp32 range itself is broadcast.
    A rank's fp32 master and moments are then current only inside its own range: ``gather_master()`` (called by ``state_dict`` and
    before a checkpoint is written) re-assembles them everywhere.  Not combined with the fp8 mode (the e4m3 weight copies are
    quantised from the full fp32 master after every step): ``shard`` is ignored there."""

    def __init__(self, model, lr, bert_lr=None, warmup=-1, t_total=-1, b1=0.9, b2=0.999, e=1e-6, max_grad_norm=1.0, shard=False):
        self.model, self.arena = model, model.arena
        self.lr, self.bert_lr = lr, lr if bert_lr is None else bert_lr
        self.warmup, self.t_total = max(warmup, 0.0), t_total
        self.b1, self.b2, self.e, self.max_grad_norm = b1, b2, e, max_grad_norm
        self.step_count = 0
        a = self.arena
        if a.m is None:
            a.m = torch.zeros_like(a.p)
            a.v = torch.zeros_like(a.p)
        # two launch sets: everything but the embedding tables, and the embedding tables.  Under data parallelism the
        # tables' gradients are the last to be exchanged (the embedding backward is the last kernel); updating the other
        # 85 M parameters meanwhile hides most of that exchange.
        is_emb = lambda name: name.startswith("bert_encoder.embeddings.")
        self.parts = []
        self._selects = (lambda n: not is_emb(n), is_emb)
        for sel in self._selects:
            descs, n_t, n_b = a.build_descs(self.lr, self.bert_lr, select=sel)
            ws = torch.empty((n_b + n_t + 16) * 4, dtype=torch.uint8, device=a.device)
            self.parts.append((descs, n_t, n_b, ws))
        self.rank, self.world = 0, 1
        if dist.is_available() and dist.is_initialized():
            self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.sharded = bool(shard) and self.world > 1 and getattr(a, "w8", None) is None
        self.owner_ranges = None
        if self.sharded:
            self._plan_shards()

    # ---- sharding plan (identical on every rank: pure geometry) -------------------------------------------------------------
    def _plan_shards(self):
        a, W = self.arena, self.world
        self.blk_bounds, self.elem_ranges, self.small_idx, self.partials, self.coefs = [], [], [], [], []
        for part, sel in enumerate(self._selects):
            blocks = a.block_table(sel)
            n_t, n_b = self.parts[part][1], self.parts[part][2]
            assert len(blocks) == n_b, (len(blocks), n_b)
            weight = [n if act else 0 for _, n, act in blocks]
            total, acc, cuts = sum(weight), 0, [0]
            for i, w in enumerate(weight):                      # rank r's range ends where the running total passes (r + 1) / W of the elements
                acc += w
                while len(cuts) < W and acc * W >= total * len(cuts):
                    cuts.append(i + 1)
            while len(cuts) < W:
                cuts.append(n_b)
            cuts.append(n_b)
            bb = [(cuts[r], max(cuts[r], cuts[r + 1])) for r in range(W)]
            er = [(blocks[lo][0], blocks[hi - 1][0] + blocks[hi - 1][1]) if hi > lo else (0, 0) for lo, hi in bb]
            self.blk_bounds.append(bb)
            self.elem_ranges.append(er)
            idx = []
            smalls = a.fp32_read_slots(sel)
            for r, (lo, hi) in enumerate(er):
                pieces = [torch.arange(max(s.offset, lo), min(s.offset + s.numel, hi), dtype=torch.long) for s in smalls
                          if min(s.offset + s.numel, hi) > max(s.offset, lo)]
                idx.append((torch.cat(pieces) if pieces else torch.empty(0, dtype=torch.long)).to(a.device))
            self.small_idx.append(idx)
            self.partials.append(torch.zeros(max(n_b, 1), dtype=torch.float32, device=a.device))
            self.coefs.append(torch.zeros(max(n_t, 1), dtype=torch.float32, device=a.device))
        # what trainer.GradReducer needs: per rank, the arena ranges whose reduced gradient it must receive
        self.owner_ranges = [[self.elem_ranges[p_][r] for p_ in range(len(self.parts)) if self.elem_ranges[p_][r][1] > self.elem_ranges[p_][r][0]]
                             for r in range(W)]

    def get_lr_mult(self):
        return warmup_linear(self.step_count, self.t_total, self.warmup)

    def zero_grad(self):
        self.arena.g.zero_()

    def _launch(self, part):
        a = self.arena
        descs, n_t, n_b, ws = self.parts[part]
        if n_t == 0:
            return
        if self.sharded:
            return self._launch_sharded(part)
        hb.check(hb.lib().nbest_bertadam_step(hb.ptr(a.p), hb.ptr(a.g), hb.ptr(a.m), hb.ptr(a.v), hb.ptr(a.w16),
                                              hb.ptr(descs), n_t, n_b, self.get_lr_mult(),
                                              self.b1, self.b2, self.e, self.max_grad_norm, hb.ptr(ws),
                                              ws.numel(), hb.stream_ptr()), "bertadam_step")

    def _launch_sharded(self, part):
        a = self.arena
        descs, n_t, n_b, _ = self.parts[part]
        lo, hi = self.blk_bounds[part][self.rank]
        partial, coef = self.partials[part], self.coefs[part]
        partial.zero_()
        hb.check(hb.lib().nbest_bertadam_norms(hb.ptr(a.g), hb.ptr(descs), n_t, n_b, lo, hi, hb.ptr(partial), hb.stream_ptr()), "bertadam_norms")
        dist.all_reduce(partial, op=dist.ReduceOp.SUM)          # every block has ONE owner: the sum only fills in the other ranks' blocks
        hb.check(hb.lib().nbest_bertadam_update(hb.ptr(a.p), hb.ptr(a.g), hb.ptr(a.m), hb.ptr(a.v), hb.ptr(a.w16), hb.ptr(descs), n_t, n_b,
                                                lo, hi, hb.ptr(partial), hb.ptr(coef), self.get_lr_mult(), self.b1, self.b2, self.e,
                                                self.max_grad_norm, hb.stream_ptr()), "bertadam_update")
        # the owners hand out what the next step reads
        for r, (elo, ehi) in enumerate(self.elem_ranges[part]):
            if ehi <= elo:
                continue
            if a.w16 is None:                                   # fp32 compute: the kernels read the master arena itself
                dist.broadcast(a.p[elo:ehi], src=r)
                continue
            dist.broadcast(a.w16[elo:ehi], src=r)
            idx = self.small_idx[part][r]
            if idx.numel():
                buf = a.p.index_select(0, idx) if r == self.rank else torch.empty(idx.numel(), dtype=torch.float32, device=a.device)
                dist.broadcast(buf, src=r)
                if r != self.rank:
                    a.p.index_copy_(0, idx, buf)

    def gather_master(self, moments=True):
        """sharded mode: make the fp32 master (and the moments) current on every rank - before a checkpoint, a state_dict,
        or a switch back to replicated updates"""
        if not self.sharded:
            return
        a = self.arena
        for part in range(len(self.parts)):
            for r, (elo, ehi) in enumerate(self.elem_ranges[part]):
                if ehi > elo:
                    dist.broadcast(a.p[elo:ehi], src=r)
                    if moments:
                        dist.broadcast(a.m[elo:ehi], src=r)
                        dist.broadcast(a.v[elo:ehi], src=r)

    def step_main(self):
        """every tensor except the embedding tables (+ the k-contiguous weight copy the next forward / dgrad reads)"""
        self._launch(0)
        self.arena.refresh_transposed()

    def step_embeddings(self):
        self._launch(1)
        self.step_count += 1

    def step(self):
        self.step_main()
        self.step_embeddings()

    def state_dict(self, gather=True):
        """per-parameter ``next_m`` / ``next_v`` keyed by parameter name plus the shared step count — the content of
        the reference optimizer's ``state[p]`` (optimization.py:256-262,300), independent of the arena layout.
        Sharded mode: ``gather_master`` is a COLLECTIVE - either every rank calls ``state_dict()``, or every rank calls
        ``gather_master()`` and the one rank that writes the checkpoint calls ``state_dict(gather=False)``."""
        if gather:
            self.gather_master()
        a = self.arena
        return dict(step=self.step_count, t_total=self.t_total, warmup=self.warmup,
                    state={s.name: dict(next_m=a.view(a.m, s.name).detach().cpu().clone(),
                                        next_v=a.view(a.v, s.name).detach().cpu().clone()) for s in a.slots})

    def load_state_dict(self, sd):
        a = self.arena
        self.step_count = int(sd["step"])
        for s in a.slots:
            st = sd["state"][s.name]
            a.view(a.m, s.name).copy_(st["next_m"])
            a.view(a.v, s.name).copy_(st["next_v"])
