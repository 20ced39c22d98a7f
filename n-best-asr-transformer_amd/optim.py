"""BertAdam over the flat parameter arena, one multi-tensor HIP launch set per step.

Interface and semantics of /root/reference/models/optimization.py:183-302 as driven by
/root/reference/n_best_asr_bert.py:540-561: each parameter tensor is its own group (lr = bert_lr for
``bert_encoder.*`` else lr; weight_decay 0.01 except bias / LayerNorm), gradient clipped PER TENSOR to
L2 norm 1.0, no bias correction, eps 1e-6, ``warmup_linear`` schedule evaluated at the step count
BEFORE the increment.
"""
import torch

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
    def __init__(self, model, lr, bert_lr=None, warmup=-1, t_total=-1, b1=0.9, b2=0.999, e=1e-6, max_grad_norm=1.0):
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
        for sel in (lambda n: not is_emb(n), is_emb):
            descs, n_t, n_b = a.build_descs(self.lr, self.bert_lr, select=sel)
            ws = torch.empty((n_b + n_t + 16) * 4, dtype=torch.uint8, device=a.device)
            self.parts.append((descs, n_t, n_b, ws))

    def get_lr_mult(self):
        return warmup_linear(self.step_count, self.t_total, self.warmup)

    def zero_grad(self):
        self.arena.g.zero_()

    def _launch(self, part):
        a = self.arena
        descs, n_t, n_b, ws = self.parts[part]
        if n_t == 0:
            return
        hb.check(hb.lib().nbest_bertadam_step(hb.ptr(a.p), hb.ptr(a.g), hb.ptr(a.m), hb.ptr(a.v), hb.ptr(a.w16),
                                              hb.ptr(descs), n_t, n_b, self.get_lr_mult(),
                                              self.b1, self.b2, self.e, self.max_grad_norm, hb.ptr(ws),
                                              ws.numel(), hb.stream_ptr()), "bertadam_step")

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

    def state_dict(self):
        """per-parameter ``next_m`` / ``next_v`` keyed by parameter name plus the shared step count — the content of
        the reference optimizer's ``state[p]`` (optimization.py:256-262,300), independent of the arena layout"""
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
