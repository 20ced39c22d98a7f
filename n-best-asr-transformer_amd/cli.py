"""Command-line front end with the flag surface of the reference's training script.

    python -m nbest_amd.cli --dataset dstc2 --dataroot dstc2_data/processed_data/raw --pre_trained_model bert \\
        --deviceId 0 --random_seed 999 --dropout 0.3 --bert_dropout 0.1 --optim_choice bertadam --lr 3e-5 --bert_lr 3e-5 \\
        --warmup_proportion 0.1 --batchSize 16 --max_epoch 50 --experiment exp/ --coverage 1.0 --add_segment_ids

Flag names, defaults and meaning follow /root/reference/n_best_asr_bert.py:39-112 as driven by
/root/reference/run/train_eval_N_Best_ASR_Transformer_STC.sh:62-75 (live flags: --pre_trained_model,
--add_l2_loss, --add_segment_ids, --coverage, --without_system_act, --dropout, --bert_dropout, --lr, --bert_lr,
--warmup_proportion, --batchSize, --max_epoch, --random_seed, --testing, --experiment, --ontology_path; flags the
reference parses but never uses are accepted and only enter the experiment-directory name, as there).
Differences forced by the environment (no network, no CUDA): the encoder is built from its published shape
(nbest_amd.config.NAMED) and initialised from ``--init_checkpoint`` (a state dict with the reference's keys, e.g. a
converted HuggingFace checkpoint or a model.pt written by either implementation) or randomly; the tokenizer is a
local WordPiece vocabulary (``--vocab``; default: the words of memory.pt).  Additive flags: --dtype, --n_best,
--label_space, --synthetic.  Under torchrun the minibatch is sharded over the ranks (RCCL gradient all-reduce).
"""
import argparse
import json
import os
import random
import sys
import time
from datetime import timedelta

import numpy as np
import torch

from . import config as ncfg, observe, synth, trainer
from .inputs import SentencePieceTokenizer, WordPieceTokenizer
from .model import NBestSTCModel
from .optim import HipBertAdam


def parse_arguments(argv=None):
    ap = argparse.ArgumentParser(description="N-best ASR transformer STC fine-tuning on MI355X (HIP)")
    g = ap.add_argument_group("model structure (accepted for compatibility; only used in the experiment directory name)")
    g.add_argument("--emb_size", type=int, default=256)
    g.add_argument("--hidden_size", type=int, default=512)
    g.add_argument("--max_seq_len", type=int, default=None)
    g.add_argument("--n_layers", type=int, default=6)
    g.add_argument("--n_head", type=int, default=4)
    g.add_argument("--d_k", type=int, default=64)
    g.add_argument("--d_v", type=int, default=64)
    g.add_argument("--score_util", default="pp", choices=["none", "np", "pp", "mul"])
    g.add_argument("--sent_repr", default="bin_sa_cls",
                   choices=["cls", "maxpool", "attn", "bin_lstm", "bin_sa", "bin_sa_cls", "tok_sa_cls"])
    g.add_argument("--cls_type", default="stc", choices=["nc", "tf_hd", "stc"])
    g = ap.add_argument_group("data")
    g.add_argument("--dataset", required=True)
    g.add_argument("--dataroot", required=True)
    g.add_argument("--train_file", default="train")
    g.add_argument("--valid_file", default="valid")
    g.add_argument("--test_file", default="test")
    g.add_argument("--ontology_path", default=None, help="ontology JSON: evaluation keeps informative act-slot-value labels only")
    g = ap.add_argument_group("encoder")
    g.add_argument("--bert_model_name", default="bert-base-uncased")
    g.add_argument("--fix_bert_model", action="store_true")
    g.add_argument("--pre_trained_model", help="bert | xlm-roberta (| xlm-roberta-large).  'roberta' is refused: the reference hands "
                   "segment ids to RoBERTa's one-row token-type table and dies with an IndexError (models/model.py:56)")
    g.add_argument("--tod_pre_trained_model", help="ToD-BERT style checkpoint: keeps [SYS]/[USR] markers")
    g = ap.add_argument_group("training / testing")
    g.add_argument("--testing", action="store_true")
    g.add_argument("--deviceId", type=int, default=-1,
                   help="as the reference (n_best_asr_bert.py:116-126): 0 = pick a GPU automatically (here: the first visible one; "
                        "the reference asks gpustat / NVML for the least loaded), k > 0 = GPU k-1, -1 = CPU (refused: the path is "
                        "HIP-only).  Under torchrun every rank uses its LOCAL_RANK GPU instead")
    g.add_argument("--random_seed", type=int, default=999)
    g.add_argument("--l2", type=float, default=0)
    g.add_argument("--dropout", type=float, default=0.0)
    g.add_argument("--bert_dropout", type=float, default=0.1)
    g.add_argument("--batchSize", type=int, default=16)
    g.add_argument("--max_norm", type=float, default=5.0)
    g.add_argument("--max_epoch", type=int, default=50)
    g.add_argument("--experiment", default="exp")
    g.add_argument("--optim_choice", default="bertadam", choices=["adam", "adamw", "bertadam"],
                   help="only bertadam (the shipped script's choice) is built as a fused HIP optimizer; adam / adamw "
                        "(n_best_asr_bert.py:552-569: torch Adam, HF AdamW + linear schedule, global-norm clip) are refused")
    g.add_argument("--lr", type=float, default=5e-4)
    g.add_argument("--bert_lr", type=float, default=1e-5)
    g.add_argument("--warmup_proportion", type=float, default=0.1)
    g.add_argument("--init_type", default="uf", choices=["uf", "xuf", "normal"])
    g.add_argument("--init_range", type=float, default=0.2)
    g.add_argument("--with_system_act", action="store_true")
    g.add_argument("--coverage", type=float)
    g.add_argument("--add_l2_loss", action="store_true")
    g.add_argument("--without_system_act", action="store_true")
    g.add_argument("--add_segment_ids", action="store_true")
    g = ap.add_argument_group("additive flags of this build")
    g.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "fp8w"],
                   help="bf16 (default) | f32 (parity path) | fp8w: the bf16 path with every GEMM of the encoder layers (forward, input "
                        "gradient, weight gradient) on the CDNA4 block-scaled fp8 MFMA, from per-matrix-scaled e4m3 copies of the weights "
                        "and e4m3 copies of activations / gradients (BASELINE configs[4]: 'fp8 weights'); fp32 master weights, attention, "
                        "LayerNorm, heads and the optimizer are unchanged")
    g.add_argument("--n_best", type=int, default=None, help="keep only the first n hypotheses of every utterance")
    g.add_argument("--init_checkpoint", default=None, help="state dict (reference keys) to start from")
    g.add_argument("--pretrained_path", default=None,
                   help="LOCAL HF-format encoder checkpoint (directory or model.safetensors / pytorch_model.bin); its vocab.txt is "
                        "used when --vocab is not given.  Stands in for from_pretrained(name), which needs the network")
    g.add_argument("--stop_after_epoch", type=int, default=None, help="leave after this epoch (preemption drills; use with --resume)")
    g.add_argument("--resume", action="store_true", help="continue from <exp_dir>/last.pt (model + BertAdam state + epoch)")
    g.add_argument("--vocab", default=None, help="WordPiece vocabulary: vocab.txt (one token per line) or a JSON list")
    g.add_argument("--label_space", default=None, help="JSON with top2bottom / idx2label instead of memory.pt")
    g.add_argument("--encoder_layers", type=int, default=None, help="override the number of encoder layers (smoke runs)")
    g.add_argument("--shard_optimizer", default="off", choices=["on", "off"],
                   help="data parallel only: BertAdam sharded over the ranks (reduce-to-owner + owner broadcasts, DESIGN 6) instead of "
                        "replicated behind the all-reduce.  Opt-in: the path has not run over RCCL on more than one GPU yet")
    opt = ap.parse_args(argv)
    if opt.optim_choice != "bertadam":
        ap.error("only --optim_choice bertadam is built (the shipped script's choice)")
    if opt.deviceId < 0:
        ap.error("--deviceId -1 (CPU) is not available: the path is HIP-only")
    if opt.pre_trained_model == "roberta":
        ap.error("--pre_trained_model roberta: the reference passes segment ids (1 after the first separator) to RoBERTa's "
                 "one-row token-type table in both encoder passes (models/model.py:45,56; n_best_asr_bert.py:255) and fails "
                 "with an IndexError; use bert or xlm-roberta")
    if opt.pre_trained_model and opt.pre_trained_model not in ncfg.NAMED:
        ap.error("--pre_trained_model %s: known shapes are %s" % (opt.pre_trained_model, ", ".join(sorted(ncfg.NAMED))))
    opt.gpu_index = 0 if opt.deviceId == 0 else opt.deviceId - 1            # n_best_asr_bert.py:116-126 (0: auto -> first GPU)
    # gradient accumulation exactly as the reference derives it (n_best_asr_bert.py:522): 4 micro-batches of batchSize / 4
    # per optimizer step when --n_layers 12 is passed (the shipped script never passes it -> 1)
    opt.n_accum_steps = 4 if opt.n_layers == 12 else 1
    opt.ontology = None if opt.ontology_path is None else json.load(open(opt.ontology_path))       # n_best_asr_bert.py:138-140
    return opt


def exp_dir(opt):
    """experiment directory name, the scheme of /root/reference/utils/util.py:20-55"""
    parts = ["nl_%s" % opt.n_layers, "nh_%s" % opt.n_head, "dk_%s" % opt.d_k, "dv_%s" % opt.d_v, "bs_%s" % opt.batchSize,
             "dp_%s_%s" % (opt.dropout, opt.bert_dropout),
             "opt_%s_%s_%s_%s" % (opt.optim_choice, opt.warmup_proportion, opt.lr, opt.bert_lr), "mn_%s" % opt.max_norm,
             "me_%s" % opt.max_epoch, "seed_%s" % opt.random_seed, "score_%s" % opt.score_util, "repr_%s" % opt.sent_repr,
             "cls_%s" % opt.cls_type]
    return os.path.join(opt.experiment, "data_%s" % opt.dataset, "__".join(parts))


def load_memory(opt):
    if opt.label_space:
        d = json.load(open(opt.label_space))
        idx2label = d["idx2label"]
        return dict(top2bottom_dict={int(k): v for k, v in d["top2bottom"].items()}, idx2label=idx2label,
                    label2idx={l: i for i, l in enumerate(idx2label)}, word2idx=d.get("word2idx", {}))
    # memory.pt is a plain dict of dicts / lists: the safe loader reads it
    m = torch.load(os.path.join(opt.dataroot, "memory.pt"), weights_only=True)
    m["idx2label"] = [m["idx2label"][i] for i in range(len(m["idx2label"]))]
    return m


def load_tokenizer(opt, memory):
    for fn in ("vocab.txt", "sentencepiece.bpe.model"):
        if not opt.vocab and opt.pretrained_path and os.path.exists(os.path.join(opt.pretrained_path, fn)):
            opt.vocab = os.path.join(opt.pretrained_path, fn)
    if opt.vocab and opt.vocab.endswith(".model"):              # sentencepiece model (XLM-R family)
        return SentencePieceTokenizer(opt.vocab)
    if opt.vocab:
        if opt.vocab.endswith(".json"):
            vocab = json.load(open(opt.vocab))
        else:
            vocab = [l.rstrip("\n") for l in open(opt.vocab, encoding="utf-8")]
    else:
        words = sorted({w.lower() for w in memory.get("word2idx", {}) if isinstance(w, str)})
        vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + [w for w in words if w and not w.startswith("[")]
        vocab = list(dict.fromkeys(vocab))
    return WordPieceTokenizer(vocab)


class _Log:
    def __init__(self, path, rank, append=False):
        self.fp = open(path, "a" if append else "w") if rank == 0 else None

    def info(self, msg):
        if self.fp:
            self.fp.write(msg + "\n")
            self.fp.flush()
            print(msg, flush=True)


def main(argv=None):
    opt = parse_arguments(argv)
    trainer.limit_host_threads()
    rank, world, local = trainer.init_distributed()
    dev = torch.device("cuda", local if world > 1 else opt.gpu_index)
    torch.cuda.set_device(dev)
    random.seed(opt.random_seed)
    np.random.seed(opt.random_seed)
    torch.manual_seed(opt.random_seed)

    memory = load_memory(opt)
    labels = ncfg.LabelSpace(memory["top2bottom_dict"], memory["idx2label"])
    opt.tokenizer = load_tokenizer(opt, memory)
    family = opt.pre_trained_model or "bert"
    cfg = ncfg.NAMED[family](hidden_dropout_prob=opt.bert_dropout, attention_probs_dropout_prob=opt.bert_dropout)
    if opt.encoder_layers:
        cfg.num_hidden_layers = opt.encoder_layers
    if (family == "bert" and not (opt.init_checkpoint or opt.pretrained_path)) or opt.vocab:
        cfg.vocab_size = max(opt.tokenizer.vocab_size, 8)          # embedding table sized for the local vocabulary
    model = NBestSTCModel(cfg, labels, device=dev, compute_dtype=torch.float32 if opt.dtype == "f32" else torch.bfloat16,
                          dropout=opt.dropout, seed=opt.random_seed, fp8_forward=(opt.dtype == "fp8w"))
    if opt.init_checkpoint:
        model.load_model(opt.init_checkpoint)
    else:
        model.load_reference_state(synth.model_state(cfg, labels, seed=opt.random_seed))
        if opt.pretrained_path:
            model.load_pretrained_encoder(opt.pretrained_path)
    trainer.broadcast_parameters(model)
    n_params = sum(s.numel for s in model.arena.slots)
    n_bert = sum(s.numel for s in model.arena.slots if "bert_encoder" in s.name)
    opt.exp_dir = exp_dir(opt)
    if rank == 0:
        os.makedirs(opt.exp_dir, exist_ok=True)
        print("word vocab size:", opt.tokenizer.vocab_size)
        print("#labels:", labels.n_bottom)
        print("#top-labels:", labels.n_top)
        print("num params: {}".format(n_params))
        print("num bert params: {}, {}%".format(n_bert, 100 * n_bert / n_params))

    def load(split, coverage=None):
        fn = os.path.join(opt.dataroot, split)
        if not os.path.exists(fn):
            return None
        return trainer.EncodedSplit(trainer.read_wcn_data(fn, coverage), opt, memory)       # tokenised once per run

    valid, test = load(opt.valid_file), load(opt.test_file)
    if opt.testing:
        model.load_model(os.path.join(opt.exp_dir, "model.pt"))
        log = _Log(os.path.join(opt.exp_dir, "log.test"), rank)
        for name, data in (("Train", load(opt.train_file)), ("Valid", valid), ("Test", test)):
            if data is None:
                continue
            with open(os.path.join(opt.exp_dir, "%s.eval" % name.lower()), "w") as fp, \
                    open(os.path.join(opt.exp_dir, "%s.eval.err" % name.lower()), "w") as efp:
                t0 = time.time()
                loss, (p, r, f), acc, _ = trainer.eval_epoch(model, data, opt, memory, fp, efp)
                log.info("[%s]\tTime: %.2f\tLoss: %.2f\t(p/r/f): (%.2f/%.2f/%.2f)\tAcc: %.2f" % (name, time.time() - t0, loss, p, r, f, acc))
        return 0

    train = load(opt.train_file, opt.coverage)
    if train is None:
        raise SystemExit("no training split at %s" % os.path.join(opt.dataroot, opt.train_file))
    t_total = (len(train) // opt.batchSize + 1) * opt.max_epoch            # n_best_asr_bert.py:556
    opt.optimizer = HipBertAdam(model, lr=opt.lr, bert_lr=opt.bert_lr, warmup=opt.warmup_proportion, t_total=t_total,
                                shard=opt.shard_optimizer == "on")
    log = _Log(os.path.join(opt.exp_dir, "log.train"), rank, append=opt.resume and os.path.exists(os.path.join(opt.exp_dir, "last.pt")))
    t_start = time.time()
    log.info("Training starts at %s" % time.asctime(time.localtime(t_start)))
    best = dict(epoch=0, vf=0.0, tef=0.0, v_acc=0.0, te_acc=0.0)
    first_epoch, last = 0, os.path.join(opt.exp_dir, "last.pt")
    if opt.resume and os.path.exists(last):
        ck = torch.load(last, map_location="cpu", weights_only=True)           # written by this program: tensors + numbers
        model.load_reference_state(ck["model"])
        opt.optimizer.load_state_dict(ck["optimizer"])
        best, first_epoch = ck["best"], ck["epoch"] + 1
        model.step_counter = int(ck["dropout_step"])                           # dropout streams continue where they stopped
        log.info("Resumed after epoch %02d (optimizer step %d)" % (ck["epoch"], opt.optimizer.step_count))
    for ep in range(first_epoch, opt.max_epoch):
        t0 = time.time()
        loss, (p, r, f), acc = trainer.train_epoch(model, train, opt, memory, epoch=ep)
        log.info("[Train]\tEpoch: %02d\tTime: %.2f\tLoss: %.2f\t(p/r/f): (%.2f/%.2f/%.2f)\tAcc: %.2f" % (ep, time.time() - t0, loss, p, r, f, acc))
        res = {}
        for name, data in (("valid", valid), ("test", test)):
            if data is None:
                continue
            fn = os.path.join(opt.exp_dir, "%s.iter%d" % (name, ep))
            with (open(fn, "w") if rank == 0 else open(os.devnull, "w")) as fp, \
                    (open(fn + ".err", "w") if rank == 0 else open(os.devnull, "w")) as efp:
                t0 = time.time()
                loss, (p, r, f), acc, cases = trainer.eval_epoch(model, data, opt, memory, fp, efp)
            log.info("[%s]\tEpoch: %02d\tTime: %.2f\tLoss: %.2f\t(p/r/f): (%.2f/%.2f/%.2f)\tAcc: %.2f" % (
                name.capitalize(), ep, time.time() - t0, loss, p, r, f, acc))
            if rank == 0:                                                           # n_best_asr_bert.py:416,426
                observe.observability_lens(observe.EpochInfoCollector.from_cases(cases, loss, (p, r, f), acc), ep, name,
                                           opt.exp_dir, "tod_asr_bert_stc")
            res[name] = (f, acc)
        vf, v_acc = res.get("valid", (0.0, 0.0))
        tef, te_acc = res.get("test", (0.0, 0.0))
        if vf > best["vf"]:
            best.update(epoch=ep, vf=vf, tef=tef, v_acc=v_acc, te_acc=te_acc)
            # sharded optimizer: the fp32 master is current only on each range's owner.  gather_master is a sequence of
            # collectives: EVERY rank runs it (vf is the all-reduced F1, so every rank takes this branch together); only
            # the file write is rank 0's
            opt.optimizer.gather_master(moments=False)
            if rank == 0:
                model.save_model(os.path.join(opt.exp_dir, "model.pt"))
            log.info("NEW BEST:\tEpoch: %02d\tvalid F1/Acc: %.2f/%.2f\ttest F1/Acc: %.2f/%.2f" % (ep, vf, v_acc, tef, te_acc))
        if opt.resume:
            opt.optimizer.gather_master()           # all ranks (collectives); master and moments are whole everywhere afterwards
            if rank == 0:                           # ... so the state dicts are built AFTER the gather, on rank 0 only
                torch.save(dict(model={k: v.detach().cpu() for k, v in model.state_dict().items()},
                                optimizer=opt.optimizer.state_dict(gather=False), best=best, epoch=ep, dropout_step=model.step_counter),
                           last + ".tmp")
                os.replace(last + ".tmp", last)
        if opt.stop_after_epoch is not None and ep >= opt.stop_after_epoch:
            log.info("Stopping after epoch %02d as requested" % ep)
            return 0
    log.info("Done training. Elapsed time: %s" % timedelta(seconds=time.time() - t_start))
    log.info("BEST RESULT:\tEpoch: %02d\tBest valid F1/Acc: %.2f/%.2f\ttest F1/Acc: %.2f/%.2f" % (
        best["epoch"], best["vf"], best["v_acc"], best["tef"], best["te_acc"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
