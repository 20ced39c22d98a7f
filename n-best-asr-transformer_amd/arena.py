"""Flat parameter arenas in HBM.

All parameters of the model (encoder + STC heads, 109.6 M for bert-base) live in ONE contiguous fp32
buffer ``p`` (the master copy); gradients ``g`` and the BertAdam moments ``m``/``v`` are buffers of the
same layout, and the bf16 *compute copy* ``w16`` read by the MFMA GEMMs mirrors it at the same element
offsets.  Consequences:

* the HIP encoder takes two base pointers plus a per-layer offset table (no per-tensor marshalling);
* the fused Q|K|V weight [3H,H] is simply the three HF tensors placed back to back, so HuggingFace
  checkpoints load with no repacking and ``state_dict()`` keeps the reference's key names
  (/root/reference/models/model.py:75-83);
* data-parallel gradient exchange is a few large all-reduces over slices of ``g`` (layer-ordered);
* BertAdam is one multi-tensor launch over the arenas (nbest_bertadam_step).

``nn.Parameter`` objects of the module tree are *views* into ``p`` and their ``.grad`` are views into
``g``.
"""
import ctypes as C

import torch

from . import hipabi as hb
from .synth import encoder_param_shapes, head_param_shapes

ALIGN = 64  # elements; keeps every tensor 128-byte aligned in bf16 and 256-byte aligned in fp32

NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")   # /root/reference/n_best_asr_bert.py:540


class Slot:
    __slots__ = ("name", "shape", "offset", "numel")

    def __init__(self, name, shape, offset):
        self.name, self.shape, self.offset = name, tuple(shape), offset
        n = 1
        for s in shape:
            n *= s
        self.numel = n


class ParamArena:
    def __init__(self, cfg, labels, device, compute_dtype=torch.bfloat16):
        self.cfg, self.labels, self.device, self.compute_dtype = cfg, labels, torch.device(device), compute_dtype
        self.slots, self.by_name = [], {}
        off = 0

        def add(name, shape, align=True):
            nonlocal off
            if align:
                off = (off + ALIGN - 1) // ALIGN * ALIGN
            s = Slot(name, shape, off)
            self.slots.append(s)
            self.by_name[name] = s
            off += s.numel
            return s

        enc = dict(encoder_param_shapes(cfg))
        pre = "bert_encoder."
        for n in ("embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight",
                  "embeddings.token_type_embeddings.weight", "embeddings.LayerNorm.weight", "embeddings.LayerNorm.bias"):
            add(pre + n, enc[n])
        self.layer_offsets = (hb.LayerOffsets * cfg.num_hidden_layers)()
        self.layer_range = []
        for i in range(cfg.num_hidden_layers):
            lp = "encoder.layer.%d." % i
            lo = None
            o = self.layer_offsets[i]
            for j, nm in enumerate(("query", "key", "value")):           # fused [3H,H]: contiguous, no padding
                s = add(pre + lp + "attention.self.%s.weight" % nm, enc[lp + "attention.self.%s.weight" % nm], align=(j == 0))
                if j == 0:
                    o.wqkv, lo = s.offset, s.offset
            for j, nm in enumerate(("query", "key", "value")):
                s = add(pre + lp + "attention.self.%s.bias" % nm, enc[lp + "attention.self.%s.bias" % nm], align=(j == 0))
                if j == 0:
                    o.bqkv = s.offset
            for field, nm in (("wo", "attention.output.dense.weight"), ("bo", "attention.output.dense.bias"),
                              ("ln1_g", "attention.output.LayerNorm.weight"), ("ln1_b", "attention.output.LayerNorm.bias"),
                              ("w1", "intermediate.dense.weight"), ("b1", "intermediate.dense.bias"),
                              ("w2", "output.dense.weight"), ("b2", "output.dense.bias"),
                              ("ln2_g", "output.LayerNorm.weight"), ("ln2_b", "output.LayerNorm.bias")):
                setattr(o, field, add(pre + lp + nm, enc[lp + nm]).offset)
            self.layer_range.append((lo, off))
        add(pre + "pooler.dense.weight", enc["pooler.dense.weight"])
        add(pre + "pooler.dense.bias", enc["pooler.dense.bias"])
        heads = head_param_shapes(labels, cfg.hidden_size)
        w_names = [n for n, _ in heads if n.endswith("weight")]
        b_names = [n for n, _ in heads if n.endswith("bias")]
        hs = dict(heads)
        for j, n in enumerate(w_names):                                    # Wh [R,H] contiguous
            s = add("clf." + n, hs[n], align=(j == 0))
            if j == 0:
                self.off_wh = s.offset
        for j, n in enumerate(b_names):                                    # bh [R] contiguous
            s = add("clf." + n, hs[n], align=(j == 0))
            if j == 0:
                self.off_bh = s.offset
        self.total = (off + ALIGN - 1) // ALIGN * ALIGN
        self.heads_range = (self.off_wh, off)
        self.emb_range = (0, self.layer_range[0][0]) if self.layer_range else (0, self.by_name[pre + "pooler.dense.weight"].offset)

        f32 = dict(dtype=torch.float32, device=self.device)
        self.p = torch.zeros(self.total, **f32)
        self.g = torch.zeros(self.total, **f32)
        self.m = None
        self.v = None
        self.w16 = torch.zeros(self.total, dtype=torch.bfloat16, device=self.device) if compute_dtype == torch.bfloat16 else None
        # transposed bf16 copy of the per-layer weight matrices (same offsets): the dgrad GEMMs read it so both of
        # their operands are k-contiguous.  Only allocated on a GPU (the kernel that fills it is HIP).
        self.w16t = None
        self.lazy_w16t = False      # set by the model while its backward runs in fp8: the bf16 transposed copy has no reader then
        self.w16t_stale = False
        self.w8 = self.w8t = self.w8_inv_scale = self._w8_ws = self.gamax = self.aamax = self.gamax_slots = self.aamax_slots = None
        self._tdescs = None
        self.wpk = self.wpkt = None     # packed copies for the bf16 GEMMs (dropped when the fp8 mode is enabled: its GEMMs read w8 / w8t)
        self.w8p = self.w8tp = None
        self._pdescs = None
        if self.w16 is not None and self.device.type == "cuda":
            self.w16t = torch.zeros(self.total, dtype=torch.bfloat16, device=self.device)
            H, F = cfg.hidden_size, cfg.intermediate_size
            mats = []
            for o in self.layer_offsets:
                mats += [(o.wqkv, 3 * H, H), (o.wo, H, H), (o.w1, F, H), (o.w2, H, F)]
            arr = (hb.MatrixDesc * len(mats))()
            t = 0
            for i, (off, r, c) in enumerate(mats):
                arr[i].offset, arr[i].rows, arr[i].cols, arr[i].tile_start = off, r, c, t
                t += ((r + 63) // 64) * ((c + 63) // 64)
            self._tdescs = (torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device), len(mats), t)
            # the same matrices PACKED for the k-contiguous GEMM kernels (nbest_pack_weights): wpk from w16 ([out][in], forward GEMMs),
            # wpkt from w16t ([in][out], dgrad GEMMs).  All or nothing: encoder.hip hands `wpk + offset` to EVERY GEMM of a layer, so as
            # soon as one matrix has no packed form (nbest_pack_bn == 0: e.g. H = 512, whose widths are neither multiples of 192 nor
            # >= 1024 multiples of 256) both tables are dropped and every GEMM reads w16 / w16t row by row - same results.
            self._pdescs = []
            for transposed in (False, True):
                sel = []
                for off, r, c in mats:
                    n, k = (c, r) if transposed else (r, c)
                    bn = hb.lib().nbest_pack_bn(n)
                    if not (bn > 0 and k % 32 == 0 and n % bn == 0):
                        sel = None
                        break
                    sel.append((off, n, k, bn))
                if sel is None:
                    self._pdescs = None
                    break
                parr = (hb.MatrixDesc * len(sel))()
                t = 0
                for i, (off, n, k, bn) in enumerate(sel):
                    parr[i].offset, parr[i].rows, parr[i].cols, parr[i].tile_start, parr[i].pad = off, n, k, t, bn
                    t += (n // bn) * (k // 32)
                self._pdescs.append((torch.frombuffer(bytearray(bytes(parr)), dtype=torch.uint8).to(self.device), len(sel), t))
            if self._pdescs is not None:
                self.wpk = torch.zeros(self.total, dtype=torch.bfloat16, device=self.device)
                self.wpkt = torch.zeros(self.total, dtype=torch.bfloat16, device=self.device)
        self._descs = None

    # ---- views ---------------------------------------------------------------------------------
    def view(self, buf, name):
        s = self.by_name[name]
        return buf[s.offset:s.offset + s.numel].view(s.shape)

    def heads_wb(self):
        R, H = self.labels.n_head_rows, self.cfg.hidden_size
        return self.p[self.off_wh:self.off_wh + R * H].view(R, H), self.p[self.off_bh:self.off_bh + R]

    def heads_grad_wb(self):
        R, H = self.labels.n_head_rows, self.cfg.hidden_size
        return self.g[self.off_wh:self.off_wh + R * H].view(R, H), self.g[self.off_bh:self.off_bh + R]

    @property
    def weights(self):
        """arena the GEMMs / gathers read: bf16 compute copy, or the fp32 master in fp32 mode"""
        return self.w16 if self.w16 is not None else self.p

    def refresh_compute_copy(self):
        if self.w16 is not None:
            hb.cast_bf16(self.p, self.w16)
        self.refresh_transposed()

    def refresh_transposed(self):
        """the derived copies of the weight matrices the next step reads: k-contiguous bf16 copy for the dgrads and, with
        the fp8 forward enabled, the e4m3 copy + per-matrix scales (called after every optimizer step)"""
        if self.w16t is not None:
            if self.lazy_w16t:
                self.w16t_stale = True          # model._backward_pass refreshes it should a bf16 backward run after all
            else:
                self.refresh_w16t()
        if self.w8 is not None:
            d, n, t = self._tdescs
            hb.check(hb.lib().nbest_quantize_weights_fp8(hb.ptr(self.p), hb.ptr(self.w8), hb.ptr(self.w8t), hb.ptr(d), n, t,
                                                         hb.ptr(self.w8_inv_scale), hb.ptr(self._w8_ws), self._w8_ws.numel(),
                                                         hb.stream_ptr()), "quantize_weights_fp8")
            if getattr(self, "w8p", None) is not None:
                for src, dst, (pd, pn, pt) in ((self.w8, self.w8p, self._p8descs[0]), (self.w8t, self.w8tp, self._p8descs[1])):
                    hb.check(hb.lib().nbest_pack_weights_fp8(hb.ptr(src), hb.ptr(dst), hb.ptr(pd), pn, pt, hb.stream_ptr()), "pack_weights_fp8")

    def refresh_w16t(self):
        d, n, t = self._tdescs
        hb.check(hb.lib().nbest_transpose_weights(hb.ptr(self.w16), hb.ptr(self.w16t), hb.ptr(d), n, t, hb.stream_ptr()),
                 "transpose_weights")
        self.w16t_stale = False
        if self.wpk is not None:
            for src, dst, (pd, pn, pt) in ((self.w16, self.wpk, self._pdescs[0]), (self.w16t, self.wpkt, self._pdescs[1])):
                hb.check(hb.lib().nbest_pack_weights(hb.ptr(src), hb.ptr(dst), hb.ptr(pd), pn, pt, hb.stream_ptr()), "pack_weights")

    def enable_fp8_forward(self):
        """allocate the e4m3 weight copy (one byte per element at the arena's element offsets) and its per-matrix inverse
        scales [4 L] (QKV, attention-out, FFN-up, FFN-down per layer) - BASELINE configs[4] "fp8 weights"."""
        if self.w16t is None:
            raise RuntimeError("nbest_amd: the fp8 forward needs the bf16 path on a GPU")
        self.wpk = self.wpkt = None          # the fp8 GEMMs read w8 / w8t; the bf16 fall-back GEMMs read w16 / w16t unpacked
        if self.w8 is None:
            n = self._tdescs[1]
            self.w8 = torch.zeros(self.total, dtype=torch.uint8, device=self.device)
            self.w8t = torch.zeros(self.total, dtype=torch.uint8, device=self.device)      # transposed: B operand of the fp8 dgrads
            # gradient amax history of the fp8 dgrads (float bits) [4 L]: what the next pass reads; and the slot array this pass
            # records into (include/nbest_hip.h nbest_fp8_amax_fold), folded into the history by the model after every backward
            self.gamax = torch.zeros(n, dtype=torch.int32, device=self.device)
            self.gamax_slots = torch.zeros(n * hb.AMAX_TENSOR_WORDS, dtype=torch.int32, device=self.device)
            # ... and the ACTIVATION amax history of the fp8 forward (x, ctx, x1, gelu(u) per layer), folded after every step
            self.aamax = torch.zeros(n, dtype=torch.int32, device=self.device)
            self.aamax_slots = torch.zeros(n * hb.AMAX_TENSOR_WORDS, dtype=torch.int32, device=self.device)
            self.w8_inv_scale = torch.ones(n, dtype=torch.float32, device=self.device)
            self._w8_ws = torch.zeros(4 * n + 16, dtype=torch.uint8, device=self.device)
            # w8 / w8t packed for gemm8_kernel's tiles (nbest_pack_weights_fp8), as wpk / wpkt for the bf16 kernels
            H, F = self.cfg.hidden_size, self.cfg.intermediate_size
            mats = []
            for o in self.layer_offsets:
                mats += [(o.wqkv, 3 * H, H), (o.wo, H, H), (o.w1, F, H), (o.w2, H, F)]
            self._p8descs = []
            for transposed in (False, True):
                parr = (hb.MatrixDesc * len(mats))()
                t = 0
                for i, (off, r, c) in enumerate(mats):
                    nn, kk = (c, r) if transposed else (r, c)
                    bn = hb.lib().nbest_pack_bn_fp8(nn, kk)
                    if not (bn > 0 and kk % 64 == 0 and nn % bn == 0):       # all or nothing, as for wpk / wpkt above
                        parr = None
                        break
                    parr[i].offset, parr[i].rows, parr[i].cols, parr[i].tile_start, parr[i].pad = off, nn, kk, t, bn
                    t += (nn // bn) * (kk // 64)
                if parr is None:
                    self._p8descs = None
                    break
                self._p8descs.append((torch.frombuffer(bytearray(bytes(parr)), dtype=torch.uint8).to(self.device), len(mats), t))
            self.w8p = self.w8tp = None
            if self._p8descs is not None:
                self.w8p = torch.zeros(self.total, dtype=torch.uint8, device=self.device)
                self.w8tp = torch.zeros(self.total, dtype=torch.uint8, device=self.device)
            self.refresh_transposed()

    def load_state(self, sd, strict=True):
        """copy a reference-keyed state dict (numpy arrays or tensors) into the master arena"""
        missing = []
        for s in self.slots:
            if s.name in sd:
                src = sd[s.name]
                src = torch.as_tensor(src)
                if tuple(src.shape) != s.shape:
                    raise RuntimeError("shape mismatch for %s: %s vs %s" % (s.name, tuple(src.shape), s.shape))
                self.view(self.p, s.name).copy_(src.to(torch.float32))
            else:
                missing.append(s.name)
        if strict and missing:
            raise RuntimeError("missing keys: %s" % missing[:5])
        self.refresh_compute_copy()
        return missing

    # ---- optimizer descriptors -----------------------------------------------------------------
    def block_table(self, select=None):
        """host view of the optimizer's block list for the tensors ``select`` picks (the order of build_descs): one
        (arena offset, elements, active) triple per block of nbest_bertadam_chunk() elements of one tensor"""
        chunk = hb.lib().nbest_bertadam_chunk()
        out = []
        for s in self.slots:
            if select is not None and not select(s.name):
                continue
            act = 0 if "pooler" in s.name else 1
            for c0 in range(0, s.numel, chunk):
                out.append((s.offset + c0, min(chunk, s.numel - c0), act))
        return out

    def fp32_read_slots(self, select=None):
        """tensors the kernels read from the fp32 MASTER arena even in the bf16 path (biases, LayerNorm parameters, the STC heads):
        a data-parallel rank that does not own them needs their updated fp32 values, not the bf16 compute copy"""
        return [s for s in self.slots if (select is None or select(s.name)) and "pooler" not in s.name and
                (s.name.startswith("clf.") or s.name.endswith("bias") or "LayerNorm" in s.name)]

    def build_descs(self, lr, bert_lr, active=None, select=None):
        """device array of nbest_tensor_desc, one per tensor (grouping of n_best_asr_bert.py:540-550);
        ``select(name)`` restricts the set to some tensors (the optimizer is split so the embedding tables can be
        updated after their own, last, gradient exchange)."""
        chunk = hb.lib().nbest_bertadam_chunk()
        slots = [s for s in self.slots if select is None or select(s.name)]
        n = len(slots)
        arr = (hb.TensorDesc * max(n, 1))()
        blk = 0
        for i, s in enumerate(slots):
            d = arr[i]
            d.offset, d.numel = s.offset, s.numel
            d.lr = bert_lr if "bert_encoder" in s.name else lr
            d.wd = 0.0 if any(nd in s.name for nd in NO_DECAY) else 0.01
            d.active = 0 if "pooler" in s.name else 1            # never receives a gradient
            if active is not None:
                d.active = int(bool(active(s.name)))
            d.block_start = blk
            blk += (s.numel + chunk - 1) // chunk
        raw = bytes(arr)
        dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        return dev, n, blk
