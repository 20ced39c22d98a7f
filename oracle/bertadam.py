"""Oracle: BertAdam step, restating /root/reference/models/optimization.py:237-302 with the
warmup-linear schedule of :162-171 and the per-parameter grouping of
/root/reference/n_best_asr_bert.py:540-561 (every parameter is its own group, so the gradient clip
of :270-271 is PER TENSOR to L2 norm 1.0; no bias correction; eps 1e-6 added to sqrt(v);
weight decay added to the update before the learning rate; schedule multiplier uses state['step']
BEFORE the increment).

TEST INFRASTRUCTURE - see oracle/__init__.py.
"""
import torch


def warmup_linear(step, t_total, warmup):
    if t_total < 0:
        return 1.0
    x = float(step) / float(t_total)
    if x < warmup:
        return x / warmup
    return max((x - 1.0) / (warmup - 1.0), 0.0)


NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")


def group_for(name, lr, bert_lr):
    """n_best_asr_bert.py:540-550."""
    return dict(weight_decay=0.0 if any(nd in name for nd in NO_DECAY) else 0.01,
                lr=bert_lr if "bert_encoder" in name else lr)


class OracleBertAdam:
    def __init__(self, named_params, lr, bert_lr, warmup, t_total, b1=0.9, b2=0.999, e=1e-6, max_grad_norm=1.0):
        self.items = [(n, p, group_for(n, lr, bert_lr)) for n, p in named_params if p.requires_grad]
        self.state = {}
        self.warmup, self.t_total, self.b1, self.b2, self.e, self.max_grad_norm = warmup, t_total, b1, b2, e, max_grad_norm

    def zero_grad(self):
        for _, p, _ in self.items:
            p.grad = None

    @torch.no_grad()
    def step(self):
        for n, p, g in self.items:
            if p.grad is None:
                continue
            grad = p.grad
            st = self.state.setdefault(n, dict(step=0, m=torch.zeros_like(p), v=torch.zeros_like(p)))
            if self.max_grad_norm > 0:
                # torch.nn.utils.clip_grad_norm_(p, max_norm): coef = max_norm/(norm+1e-6), clamped to 1
                norm = grad.norm(2)
                coef = torch.clamp(self.max_grad_norm / (norm + 1e-6), max=1.0)
                grad.mul_(coef)
            st["m"].mul_(self.b1).add_(grad, alpha=1 - self.b1)
            st["v"].mul_(self.b2).addcmul_(grad, grad, value=1 - self.b2)
            upd = st["m"] / (st["v"].sqrt() + self.e)
            if g["weight_decay"] > 0.0:
                upd = upd + g["weight_decay"] * p
            lr = g["lr"] * warmup_linear(st["step"], self.t_total, self.warmup)
            p.add_(upd, alpha=-lr)
            st["step"] += 1
