"""Oracle: one fine-tuning step on CPU = forward + losses + backward + BertAdam, restating the
body of train_epoch, /root/reference/n_best_asr_bert.py:242-280 (n_accum_steps = 1, :522).
Also the timed leg of bench.py's ``cpu_baseline`` (kind "port").

TEST INFRASTRUCTURE - see oracle/__init__.py.
"""
from .stc import total_loss


def train_step(model, optim, batch, top2bottom, b2t, add_l2_loss=False, skip_unused_transcript=False):
    """batch: dict(ids, seg, tids, tseg, labels).  Returns (loss_record, outputs)."""
    tids, tseg = batch.get("tids"), batch.get("tseg")
    if skip_unused_transcript and not add_l2_loss:
        tids = tseg = None      # the reference computes and discards this pass
    top, bottoms, final, asr_cls, trans_cls = model(batch["ids"], tids, seg_ids=batch.get("seg"), trans_seg_ids=tseg)
    record, total, parts = total_loss(top, bottoms, final, batch["labels"], top2bottom, b2t,
                                      asr_cls, trans_cls, add_l2_loss)
    total.backward()
    optim.step()
    optim.zero_grad()
    return record, dict(top=top, bottoms=bottoms, final=final, asr_cls=asr_cls, trans_cls=trans_cls, parts=parts)
