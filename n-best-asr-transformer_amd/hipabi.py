"""ctypes binding of the C-ABI library ``csrc/libnbest_hip.so`` (declared in include/nbest_hip.h).

PyTorch is used only for device memory and streams: every wrapper passes ``tensor.data_ptr()`` and
the current HIP stream handle.  There is NO fallback: if the library is missing or an entry point
fails, a RuntimeError is raised (the product path must never silently run on the CPU).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NBEST_LIB") or os.path.join(_HERE, "csrc", "libnbest_hip.so")   # NBEST_LIB: diagnostic builds only

F32, BF16 = 0, 1
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_DROP_RES, EPI_DGELU, EPI_RES, EPI_F32_SPLITK = range(7)

EXPORTS = [
    "nbest_version", "nbest_last_error", "nbest_embed_ln_fwd", "nbest_embed_ln_bwd", "nbest_embed_bwd_ws_bytes", "nbest_rows_gather", "nbest_rows_zero", "nbest_rows_add",
    "nbest_gemm_ws_bytes", "nbest_gemm", "nbest_wgrad_pair_ws_bytes", "nbest_wgrad_pair", "nbest_pack_bn", "nbest_pack_weights", "nbest_pack_bn_fp8", "nbest_pack_weights_fp8", "nbest_attention_fwd", "nbest_attention_bwd", "nbest_attention_bwd_ws_bytes", "nbest_attention_keep_bytes", "nbest_attention_fwd_keep", "nbest_attention_bwd_keep", "nbest_layernorm_fwd",
    "nbest_layernorm_bwd", "nbest_rowred_ws_bytes", "nbest_colsum", "nbest_heads_ws_bytes", "nbest_stc_heads",
    "nbest_stc_heads_vjp", "nbest_cls_mse", "nbest_cls_grad_scatter", "nbest_stc_decode", "nbest_stream_stamp", "nbest_fp8_amax_fold", "nbest_bertadam_chunk", "nbest_bertadam_step", "nbest_bertadam_norms", "nbest_bertadam_update",
    "nbest_cast_f32_to_bf16", "nbest_transpose_weights", "nbest_encoder_act_bytes", "nbest_encoder_ws_bytes", "nbest_encoder_wgrad_launches_per_layer", "nbest_encoder_forward",
    "nbest_encoder_backward", "nbest_gemm_fp8", "nbest_gemm_fp8_ws_bytes", "nbest_wgrad_fp8", "nbest_wgrad_fp8_ws_bytes", "nbest_wgrad_fp8_pair", "nbest_wgrad_fp8_pair_ws_bytes", "nbest_cast_bf16_to_fp8", "nbest_quantize_weights_fp8",
]


class GemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p), ("R", C.c_void_p),
                ("U", C.c_void_p), ("ws", C.c_void_p), ("ws_bytes", C.c_size_t),
                ("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64),
                ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64), ("ldr", C.c_int64), ("ldu", C.c_int64),
                ("trans_a", C.c_int32), ("trans_b", C.c_int32), ("epilogue", C.c_int32), ("dtype", C.c_int32),
                ("accumulate", C.c_int32), ("drop_p", C.c_float), ("drop_stream", C.c_uint32), ("seed", C.c_uint64),
                ("colsum_out", C.c_void_p), ("colsum_accumulate", C.c_int32), ("flags", C.c_int32),
                ("B_packed", C.c_void_p), ("b_pack_bn", C.c_int32), ("pad_", C.c_int32)]


class GemmFp8Args(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p), ("R", C.c_void_p), ("U", C.c_void_p),
                ("C8", C.c_void_p), ("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64),
                ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64), ("ldr", C.c_int64), ("ldu", C.c_int64), ("ldc8", C.c_int64),
                ("epilogue", C.c_int32), ("out_scale", C.c_float), ("drop_p", C.c_float), ("drop_stream", C.c_uint32), ("seed", C.c_uint64),
                ("out_scale_dev", C.c_void_p), ("a_amax", C.c_void_p), ("c8_amax_prev", C.c_void_p), ("c8_amax_new", C.c_void_p),
                ("colsum_out", C.c_void_p), ("colsum_accumulate", C.c_int32), ("pad", C.c_int32), ("ws", C.c_void_p), ("ws_bytes", C.c_size_t),
                ("B_packed", C.c_void_p), ("b_pack_bn", C.c_int32), ("pad3", C.c_int32)]


class LabelSpaceC(C.Structure):
    _fields_ = [("n_top", C.c_int32), ("n_bottom", C.c_int32), ("n_rows", C.c_int32),
                ("bottom_off", C.c_void_p), ("bottom_ids", C.c_void_p), ("head_row", C.c_void_p)]


class TensorDesc(C.Structure):
    _fields_ = [("offset", C.c_int64), ("numel", C.c_int64), ("lr", C.c_float), ("wd", C.c_float),
                ("active", C.c_int32), ("block_start", C.c_int32)]


class MatrixDesc(C.Structure):
    _fields_ = [("offset", C.c_int64), ("rows", C.c_int32), ("cols", C.c_int32), ("tile_start", C.c_int32), ("pad", C.c_int32)]


class LayerOffsets(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("wqkv", "bqkv", "wo", "bo", "ln1_g", "ln1_b", "w1", "b1", "w2", "b2", "ln2_g", "ln2_b")]


class EncoderDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("B", C.c_int32), ("S", C.c_int32), ("H", C.c_int32), ("L", C.c_int32),
                ("heads", C.c_int32), ("F", C.c_int32), ("vocab", C.c_int32), ("max_pos", C.c_int32), ("n_types", C.c_int32),
                ("ln_eps", C.c_float), ("hidden_drop", C.c_float), ("attn_drop", C.c_float),
                ("word_pad_id", C.c_int64), ("pos_pad_id", C.c_int64),
                ("off_word", C.c_int64), ("off_pos", C.c_int64), ("off_type", C.c_int64),
                ("off_emb_ln_g", C.c_int64), ("off_emb_ln_b", C.c_int64),
                ("layers_host", C.POINTER(LayerOffsets)), ("seed", C.c_uint64), ("drop_stream_base", C.c_uint32),
                ("wgrad_events_n", C.c_int32), ("wgrad_events", C.POINTER(C.c_void_p)),
                ("w8", C.c_void_p), ("w8_inv_scale", C.c_void_p), ("w8t", C.c_void_p), ("gamax_prev", C.c_void_p),
                ("gamax_new", C.c_void_p), ("fp8_bwd", C.c_int32), ("pad2", C.c_int32),
                ("wpk", C.c_void_p), ("wpkt", C.c_void_p), ("w8p", C.c_void_p), ("w8tp", C.c_void_p),
                ("word_perm", C.c_void_p), ("aamax_prev", C.c_void_p), ("aamax_new", C.c_void_p), ("fp8_act", C.c_int32), ("pad4", C.c_int32)]


_lib = None


def lib():
    """Load the library once; fail loudly when it has not been built (``__graft_entry__.build()``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("nbest_amd: %s is missing - build it with `make -C %s` (hipcc, gfx950). "
                               "There is no CPU fallback for the product path." % (LIB_PATH, os.path.dirname(LIB_PATH)))
        L = C.CDLL(LIB_PATH)
        for name in EXPORTS:
            if not hasattr(L, name):
                raise RuntimeError("nbest_amd: %s does not export %s" % (LIB_PATH, name))
        for name in ("nbest_embed_bwd_ws_bytes", "nbest_gemm_ws_bytes", "nbest_wgrad_pair_ws_bytes", "nbest_rowred_ws_bytes", "nbest_heads_ws_bytes",
                     "nbest_attention_bwd_ws_bytes", "nbest_attention_keep_bytes",
                     "nbest_encoder_act_bytes", "nbest_encoder_ws_bytes"):
            getattr(L, name).restype = C.c_size_t
        L.nbest_embed_bwd_ws_bytes.argtypes = [C.c_int64, C.c_int64]
        L.nbest_rows_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.nbest_rows_zero.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.nbest_rows_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.nbest_rowred_ws_bytes.argtypes = [C.c_int64, C.c_int64]
        L.nbest_heads_ws_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
        L.nbest_gemm_ws_bytes.argtypes = [C.POINTER(GemmArgs)]
        L.nbest_gemm.argtypes = [C.POINTER(GemmArgs), C.c_void_p]
        L.nbest_wgrad_pair_ws_bytes.argtypes = [C.POINTER(GemmArgs), C.POINTER(GemmArgs)]
        L.nbest_wgrad_pair.argtypes = [C.POINTER(GemmArgs), C.POINTER(GemmArgs), C.c_void_p]
        L.nbest_encoder_act_bytes.argtypes = [C.POINTER(EncoderDesc)]
        L.nbest_encoder_ws_bytes.argtypes = [C.POINTER(EncoderDesc)]
        L.nbest_encoder_wgrad_launches_per_layer.argtypes = [C.POINTER(EncoderDesc)]
        vp, i64, i32, f32, u64, u32, sz = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_uint64, C.c_uint32, C.c_size_t
        L.nbest_embed_ln_fwd.argtypes = [vp] * 10 + [i64, i32, f32, i32, f32, u64, u32, vp]
        L.nbest_embed_ln_bwd.argtypes = [vp] * 15 + [i32, i32, i32, i32, i32, i64, i64, i32, i32, f32, u64, u32, vp, sz, vp]
        L.nbest_attention_fwd.argtypes = [vp] * 4 + [i32] * 5 + [f32, u64, u32, vp]
        L.nbest_attention_bwd.argtypes = [vp] * 7 + [i32, vp, sz] + [i32] * 5 + [f32, u64, u32, vp]
        L.nbest_attention_bwd_ws_bytes.argtypes = [i32, i32, i32]
        L.nbest_attention_keep_bytes.argtypes = [i32, i32, i32]
        L.nbest_attention_fwd_keep.argtypes = [vp] * 4 + [i32] * 5 + [f32, u64, u32, vp, vp]
        L.nbest_attention_bwd_keep.argtypes = [vp] * 7 + [i32, vp, sz] + [i32] * 5 + [f32, u64, u32, vp, vp]
        L.nbest_layernorm_fwd.argtypes = [vp] * 5 + [i64, i32, f32, i32, vp]
        L.nbest_layernorm_bwd.argtypes = [vp] * 9 + [i64, i32, i32, i32, f32, u64, u32, vp, sz, vp]
        L.nbest_colsum.argtypes = [vp, vp, i64, i64, i64, i32, i32, vp, sz, vp]
        L.nbest_stc_heads.argtypes = [vp, i64, vp, vp, C.POINTER(LabelSpaceC)] + [vp] * 8 + [i32] * 5 + [f32, u64, u32, vp, sz, vp]
        L.nbest_stc_heads_vjp.argtypes = [vp, C.POINTER(LabelSpaceC)] + [vp] * 8 + [i32, i32, i32, f32, u64, u32, vp, sz, vp]
        L.nbest_cls_mse.argtypes = [vp, i64, vp, i64, vp, vp, vp, i32, i32, i32, f32, vp]
        L.nbest_cls_grad_scatter.argtypes = [vp, vp, i32, i32, i32, i32, vp]
        L.nbest_stc_decode.argtypes = [vp, vp, C.POINTER(LabelSpaceC), vp, vp, i32, vp]
        L.nbest_stream_stamp.argtypes = [vp, i32, vp]
        L.nbest_fp8_amax_fold.argtypes = [vp, vp, i32, vp]
        L.nbest_bertadam_step.argtypes = [vp] * 6 + [i32, i32, f32, f32, f32, f32, f32, vp, sz, vp]
        L.nbest_bertadam_norms.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
        L.nbest_bertadam_update.argtypes = [vp] * 6 + [i32, i32, i32, i32, vp, vp, f32, f32, f32, f32, f32, vp]
        L.nbest_cast_f32_to_bf16.argtypes = [vp, vp, i64, vp]
        L.nbest_encoder_forward.argtypes = [C.POINTER(EncoderDesc)] + [vp] * 7 + [sz, vp, sz, C.POINTER(C.c_void_p), vp]
        L.nbest_encoder_backward.argtypes = [C.POINTER(EncoderDesc)] + [vp] * 9 + [sz, vp, vp, sz, i32, i32, i32, i32, vp]
        L.nbest_transpose_weights.argtypes = [vp, vp, vp, i32, i32, vp]
        L.nbest_pack_bn.argtypes = [i64]
        L.nbest_pack_weights.argtypes = [vp, vp, vp, i32, i32, vp]
        L.nbest_pack_bn_fp8.argtypes = [i64, i64]
        L.nbest_pack_weights_fp8.argtypes = [vp, vp, vp, i32, i32, vp]
        L.nbest_gemm_fp8.argtypes = [C.POINTER(GemmFp8Args), vp]
        L.nbest_cast_bf16_to_fp8.argtypes = [vp, vp, i64, vp]
        L.nbest_wgrad_fp8_ws_bytes.restype = C.c_size_t
        L.nbest_wgrad_fp8_ws_bytes.argtypes = [i64, i64, i64]
        L.nbest_wgrad_fp8.argtypes = [vp, vp, vp, i64, i64, i64, i64, i64, i64, vp, vp, i32, vp, sz, vp]
        L.nbest_wgrad_fp8_pair_ws_bytes.restype = C.c_size_t
        L.nbest_wgrad_fp8_pair_ws_bytes.argtypes = [i64, i64, i64, i64]
        L.nbest_wgrad_fp8_pair.argtypes = [vp, vp, vp, i64, i64, i64, i64, vp, vp, vp, vp, vp, i64, i64, i64, i64, vp, vp, i64, i64, i32, vp, sz, vp]
        L.nbest_quantize_weights_fp8.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, sz, vp]
        L.nbest_gemm_fp8_ws_bytes.restype = C.c_size_t
        L.nbest_gemm_fp8_ws_bytes.argtypes = [C.POINTER(GemmFp8Args)]
        L.nbest_last_error.argtypes = [C.c_char_p, sz]
        _lib = L
    return _lib


def last_error():
    buf = C.create_string_buffer(512)
    lib().nbest_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def check(rc, what):
    if rc != 0:
        raise RuntimeError("nbest_hip %s failed (%d): %s" % (what, rc, last_error()))


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dtype_code(t):
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    raise RuntimeError("nbest_amd: unsupported activation dtype %s" % t)


def ptr(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------------------
# thin per-op wrappers (used by the kernel parity tests; training goes through encoder_forward/backward)
# ------------------------------------------------------------------------------------------------
def gemm_prepared(A, B, M, N, K, out, trans_a=False, trans_b=False, epilogue=EPI_NONE, defer_reduce=False):
    """build the argument block + workspace ONCE and return a zero-overhead launcher (benchmark loops)"""
    g = GemmArgs()
    g.A, g.B, g.C = A.data_ptr(), B.data_ptr(), out.data_ptr()
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = A.stride(0), B.stride(0), out.stride(0)
    g.trans_a, g.trans_b, g.epilogue, g.dtype = int(trans_a), int(trans_b), epilogue, dtype_code(A.dtype)
    g.flags = 1 if defer_reduce else 0
    ws = _ws(lib().nbest_gemm_ws_bytes(C.byref(g)), A.device)
    g.ws, g.ws_bytes = ws.data_ptr(), ws.numel()
    fn, ref, st = lib().nbest_gemm, C.byref(g), stream_ptr()

    def launch():
        rc = fn(ref, st)
        if rc:
            check(rc, "gemm")
    launch.keepalive = (g, ws, A, B, out)
    return launch


def pack_weight(W):
    """W [N, K] bf16 (k-contiguous) -> (packed copy, tile width) for nbest_gemm_args::B_packed, or (None, 0) when the shape has no packed form"""
    N, K = W.shape
    bn = lib().nbest_pack_bn(N)
    if bn == 0 or K % 32 or N % bn:
        return None, 0
    d = (MatrixDesc * 1)()
    d[0].offset, d[0].rows, d[0].cols, d[0].tile_start, d[0].pad = 0, N, K, 0, bn
    dd = torch.frombuffer(bytearray(bytes(d)), dtype=torch.uint8).to(W.device)
    out = torch.empty_like(W)
    check(lib().nbest_pack_weights(ptr(W), ptr(out), ptr(dd), 1, (N // bn) * (K // 32), stream_ptr()), "pack_weights")
    return out, bn


def pack_weight_fp8(W8):
    """W8 [N, K] e4m3 bytes -> (packed copy, tile width) for nbest_gemm_fp8_args::B_packed"""
    N, K = W8.shape
    bn = lib().nbest_pack_bn_fp8(N, K)
    if bn == 0 or K % 64 or N % bn:
        return None, 0
    d = (MatrixDesc * 1)()
    d[0].offset, d[0].rows, d[0].cols, d[0].tile_start, d[0].pad = 0, N, K, 0, bn
    dd = torch.frombuffer(bytearray(bytes(d)), dtype=torch.uint8).to(W8.device)
    out = torch.empty_like(W8)
    check(lib().nbest_pack_weights_fp8(ptr(W8), ptr(out), ptr(dd), 1, (N // bn) * (K // 64), stream_ptr()), "pack_weights_fp8")
    return out, bn


def gemm(A, B, M, N, K, trans_a=False, trans_b=False, epilogue=EPI_NONE, bias=None, R=None, U=None, out=None,
         accumulate=False, drop_p=0.0, seed=0, drop_stream=0, colsum_out=None, defer_reduce=False, B_packed=None, b_pack_bn=0):
    """C[M,N] = epi(op(A) . op(B)); returns C (and U for EPI_BIAS_GELU)."""
    dt = dtype_code(A.dtype)
    dev = A.device
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32 if epilogue == EPI_F32_SPLITK else A.dtype, device=dev)
    if epilogue == EPI_BIAS_GELU and U is None:        # gelu'(u): float in the fp32 path, 8-bit fixed point in the bf16 path
        U = torch.empty(M, N, dtype=torch.float32 if A.dtype == torch.float32 else torch.uint8, device=dev)
    g = GemmArgs()
    g.A, g.B, g.C = A.data_ptr(), B.data_ptr(), out.data_ptr()
    g.bias = bias.data_ptr() if bias is not None else None
    g.R = R.data_ptr() if R is not None else None
    g.U = U.data_ptr() if U is not None else None
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = A.stride(0), B.stride(0), out.stride(0)
    g.ldr = R.stride(0) if R is not None else 0
    g.ldu = U.stride(0) if U is not None else 0
    g.trans_a, g.trans_b, g.epilogue, g.dtype = int(trans_a), int(trans_b), epilogue, dt
    g.accumulate, g.drop_p, g.drop_stream, g.seed = int(accumulate), drop_p, drop_stream, seed
    g.colsum_out = colsum_out.data_ptr() if colsum_out is not None else None
    g.flags = 1 if defer_reduce else 0
    if B_packed is not None:
        g.B_packed, g.b_pack_bn = B_packed.data_ptr(), b_pack_bn
    nb = lib().nbest_gemm_ws_bytes(C.byref(g))
    ws = _ws(nb, dev)
    g.ws, g.ws_bytes = ws.data_ptr(), ws.numel()
    check(lib().nbest_gemm(C.byref(g), stream_ptr()), "gemm")
    return (out, U) if epilogue == EPI_BIAS_GELU else out


def wgrad_pair(dY1, X1, dY2, X2, out1=None, out2=None, accumulate=False):
    """(dY1^T . X1, dY2^T . X2) in fp32 by ONE launch (nbest_wgrad_pair): dY_i [K, M_i], X_i [K, N] bf16, token-major"""
    K, N = X1.shape
    outs, gs = [], []
    for dY, X, out in ((dY1, X1, out1), (dY2, X2, out2)):
        M = dY.shape[1]
        if out is None:
            out = torch.empty(M, N, dtype=torch.float32, device=X.device)
        g = GemmArgs()
        g.A, g.B, g.C = dY.data_ptr(), X.data_ptr(), out.data_ptr()
        g.M, g.N, g.K = M, N, K
        g.lda, g.ldb, g.ldc = dY.stride(0), X.stride(0), out.stride(0)
        g.trans_a = g.trans_b = 1
        g.epilogue, g.dtype, g.accumulate = EPI_F32_SPLITK, dtype_code(X.dtype), int(accumulate)
        outs.append(out)
        gs.append(g)
    nb = lib().nbest_wgrad_pair_ws_bytes(C.byref(gs[0]), C.byref(gs[1]))
    ws = _ws(nb, X1.device)
    gs[0].ws, gs[0].ws_bytes = ws.data_ptr(), ws.numel()
    check(lib().nbest_wgrad_pair(C.byref(gs[0]), C.byref(gs[1]), stream_ptr()), "wgrad_pair")
    return outs


def rows_gather(table, rows, cap):
    """(ids [cap] int64 with -1 padding, vals [cap, H] fp32 with zero padding) <- rows of the fp32 table (nbest_rows_gather)"""
    H = table.shape[1]
    ids = torch.empty(cap, dtype=torch.long, device=table.device)
    vals = torch.empty(cap, H, dtype=torch.float32, device=table.device)
    check(lib().nbest_rows_gather(ptr(table), ptr(rows), rows.numel(), cap, ptr(ids), ptr(vals), H, stream_ptr()), "rows_gather")
    return ids, vals


def rows_zero(table, ids):
    check(lib().nbest_rows_zero(ptr(table), ptr(ids), ids.numel(), table.shape[1], stream_ptr()), "rows_zero")


def rows_add(table, ids, vals):
    """table[ids[i]] += vals[i]; ids unique within the call, negative ids are padding"""
    check(lib().nbest_rows_add(ptr(table), ptr(ids), ptr(vals), ids.numel(), table.shape[1], stream_ptr()), "rows_add")


def gelu_d_decode(U):
    """gelu'(u) as float from what BIAS_GELU stored (bf16 path: q = round(200 g') + 26 in one byte)"""
    return U if U.dtype == torch.float32 else (U.float() - 26.0) / 200.0


def gelu_d_encode(g):
    return torch.clamp(torch.round(g.float() * 200.0 + 26.0), 0, 255).to(torch.uint8)      # round half to even, as the kernels


def cast_fp8(x):
    """bf16 -> e4m3 bytes (unit scale, saturating): the A operand of an fp8 forward GEMM"""
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    check(lib().nbest_cast_bf16_to_fp8(ptr(x), ptr(out), x.numel(), stream_ptr()), "cast_bf16_to_fp8")
    return out


def gemm_fp8(A8, W8, M, N, K, bias, out_scale=1.0, epilogue=EPI_BIAS, R=None, drop_p=0.0, seed=0, drop_stream=0, out=None, B_packed=None, b_pack_bn=0):
    """C[M,N] (bf16) = epi((A8 . W8^T) * out_scale + bias), A8 / W8 e4m3 bytes; BIAS_GELU also returns (U 8-bit gelu', C8 fp8 copy)"""
    dev = A8.device
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    g = GemmFp8Args()
    g.A, g.B, g.C, g.bias = A8.data_ptr(), W8.data_ptr(), out.data_ptr(), bias.data_ptr()
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc = M, N, K, A8.stride(0), W8.stride(0), out.stride(0)
    g.epilogue, g.out_scale, g.drop_p, g.drop_stream, g.seed = epilogue, out_scale, drop_p, drop_stream, seed
    U = C8 = None
    if epilogue == EPI_BIAS_GELU:
        U = torch.empty(M, N, dtype=torch.uint8, device=dev)
        C8 = torch.empty(M, N, dtype=torch.uint8, device=dev)
        g.U, g.C8, g.ldu, g.ldc8 = U.data_ptr(), C8.data_ptr(), N, N
    if R is not None:
        g.R, g.ldr = R.data_ptr(), R.stride(0)
    if B_packed is not None:
        g.B_packed, g.b_pack_bn = B_packed.data_ptr(), b_pack_bn
    check(lib().nbest_gemm_fp8(C.byref(g), stream_ptr()), "gemm_fp8")
    return (out, U, C8) if epilogue == EPI_BIAS_GELU else out


def wgrad_fp8(dY8, X8, M, N, K, a_amax=None, out=None, accumulate=False, x_amax=None):
    """dW[M,N] (fp32) = dY8^T . X8 over K token rows, both operands token-major e4m3 bytes; a_amax / x_amax: the amax words the
    delayed scales of dY8 / X8 were derived from (None = unit scale)"""
    if out is None:
        out = torch.zeros(M, N, dtype=torch.float32, device=dY8.device)
    ws = _ws(lib().nbest_wgrad_fp8_ws_bytes(M, N, K), dY8.device)
    check(lib().nbest_wgrad_fp8(ptr(dY8), ptr(X8), ptr(out), M, N, K, dY8.stride(0), X8.stride(0), out.stride(0), ptr(a_amax), ptr(x_amax),
                                int(accumulate), ptr(ws), ws.numel(), stream_ptr()), "wgrad_fp8")
    return out


def wgrad_fp8_pair(dY8a, X8a, dY8b, X8b, amax_a=None, amax_b=None, outs=None, accumulate=False, xamax_a=None, xamax_b=None):
    """(dY8a^T . X8a / s_a, dY8b^T . X8b / s_b) in fp32 by ONE launch; dY8_i [K, M_i], X8_i [K, N] token-major e4m3 bytes"""
    K, N = X8a.shape
    Ma, Mb = dY8a.shape[1], dY8b.shape[1]
    oa, ob = outs if outs is not None else (torch.zeros(Ma, N, dtype=torch.float32, device=X8a.device),
                                            torch.zeros(Mb, N, dtype=torch.float32, device=X8a.device))
    ws = _ws(lib().nbest_wgrad_fp8_pair_ws_bytes(Ma, Mb, N, K), X8a.device)
    check(lib().nbest_wgrad_fp8_pair(ptr(dY8a), ptr(X8a), ptr(oa), Ma, dY8a.stride(0), X8a.stride(0), oa.stride(0), ptr(amax_a), ptr(xamax_a),
                                     ptr(dY8b), ptr(X8b), ptr(ob), Mb, dY8b.stride(0), X8b.stride(0), ob.stride(0), ptr(amax_b), ptr(xamax_b),
                                     N, K, int(accumulate), ptr(ws), ws.numel(), stream_ptr()), "wgrad_fp8_pair")
    return oa, ob


def layernorm_fwd(x, gamma, beta, eps):
    M, H = x.shape
    y = torch.empty_like(x)
    stats = torch.empty(M, 2, dtype=torch.float32, device=x.device)
    check(lib().nbest_layernorm_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(stats), M, H, eps, dtype_code(x.dtype),
                                    stream_ptr()), "layernorm_fwd")
    return y, stats


def layernorm_bwd(dy, x, stats, gamma, want_dbias=True, drop_p=0.0, seed=0, drop_stream=0):
    M, H = x.shape
    dx = torch.empty_like(x)
    dxd = torch.empty_like(x) if drop_p > 0 else None
    dg = torch.zeros(H, dtype=torch.float32, device=x.device)
    db = torch.zeros_like(dg)
    dbias = torch.zeros_like(dg) if want_dbias else None
    ws = _ws(lib().nbest_rowred_ws_bytes(M, H), x.device)
    check(lib().nbest_layernorm_bwd(ptr(dy), ptr(x), ptr(stats), ptr(gamma), ptr(dx), ptr(dxd), ptr(dg), ptr(db), ptr(dbias),
                                    M, H, dtype_code(x.dtype), 0, drop_p, seed, drop_stream, ptr(ws), ws.numel(), stream_ptr()),
          "layernorm_bwd")
    return dx, dxd, dg, db, dbias


def colsum(X, accumulate=False, out=None):
    M, N = X.shape
    if out is None:
        out = torch.zeros(N, dtype=torch.float32, device=X.device)
    ws = _ws(lib().nbest_rowred_ws_bytes(M, N), X.device)
    check(lib().nbest_colsum(ptr(X), ptr(out), M, N, X.stride(0), dtype_code(X.dtype), int(accumulate), ptr(ws), ws.numel(),
                             stream_ptr()), "colsum")
    return out


def attention_fwd(qkv, key_mask, B, S, heads, drop_p=0.0, seed=0, drop_stream=0, want_keep=False):
    """``want_keep``: also return the dropout keep words for attention_bwd(keep=...) (None where the shape has no such path)"""
    H = heads * 64
    ctx = torch.empty(B * S, H, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(B, heads, S, dtype=torch.float32, device=qkv.device)
    if want_keep:
        nb = lib().nbest_attention_keep_bytes(B, S, heads) if qkv.dtype == torch.bfloat16 else 0
        keep = torch.zeros(nb // 4, dtype=torch.int32, device=qkv.device) if nb else None
        check(lib().nbest_attention_fwd_keep(ptr(qkv), ptr(key_mask), ptr(ctx), ptr(lse), B, S, heads, 64, dtype_code(qkv.dtype),
                                             drop_p, seed, drop_stream, ptr(keep), stream_ptr()), "attention_fwd_keep")
        return ctx, lse, keep
    check(lib().nbest_attention_fwd(ptr(qkv), ptr(key_mask), ptr(ctx), ptr(lse), B, S, heads, 64, dtype_code(qkv.dtype),
                                    drop_p, seed, drop_stream, stream_ptr()), "attention_fwd")
    return ctx, lse


def attention_bwd(qkv, key_mask, ctx, dctx, lse, B, S, heads, drop_p=0.0, seed=0, drop_stream=0, dbias=None, keep=None):
    dqkv = torch.empty_like(qkv)
    ws = _ws(lib().nbest_attention_bwd_ws_bytes(B, S, heads), qkv.device) if dbias is not None else None
    if keep is not None:
        check(lib().nbest_attention_bwd_keep(ptr(qkv), ptr(key_mask), ptr(ctx), ptr(dctx), ptr(lse), ptr(dqkv), ptr(dbias), 0, ptr(ws),
                                             ws.numel() if ws is not None else 0, B, S, heads, 64, dtype_code(qkv.dtype), drop_p, seed,
                                             drop_stream, ptr(keep), stream_ptr()), "attention_bwd_keep")
        return dqkv
    check(lib().nbest_attention_bwd(ptr(qkv), ptr(key_mask), ptr(ctx), ptr(dctx), ptr(lse), ptr(dqkv), ptr(dbias), 0, ptr(ws),
                                    ws.numel() if ws is not None else 0, B, S, heads, 64, dtype_code(qkv.dtype), drop_p, seed,
                                    drop_stream, stream_ptr()), "attention_bwd")
    return dqkv


def embed_ln_fwd(ids, seg, pos, word, type_tab, ptab, gamma, beta, eps, drop_p=0.0, seed=0, drop_stream=0):
    M, H = ids.numel(), word.shape[1]
    out = torch.empty(M, H, dtype=word.dtype, device=word.device)
    stats = torch.empty(M, 2, dtype=torch.float32, device=word.device)
    check(lib().nbest_embed_ln_fwd(ptr(ids), ptr(seg), ptr(pos), ptr(word), ptr(type_tab), ptr(ptab), ptr(gamma), ptr(beta),
                                   ptr(out), ptr(stats), M, H, eps, dtype_code(word.dtype), drop_p, seed, drop_stream,
                                   stream_ptr()), "embed_ln_fwd")
    return out, stats


def word_perm(ids):
    """token indices sorted by word id, ties in token order (int32 [B*S]): the index the deterministic embedding backward
    reduces over.  The data loaders build it on the host next to ids (trainer.EncodedSplit.host_batch, bench.py); this is the
    device-side stand-in for callers that hand over bare id tensors (a stable sort on the current stream, no host sync)."""
    return torch.sort(ids.reshape(-1), stable=True)[1].to(torch.int32)


def embed_ln_bwd(ids, seg, pos, word, type_tab, ptab, gamma, stats, dout, B, S, word_pad_id=-1, pos_pad_id=-1, drop_p=0.0,
                 seed=0, drop_stream=0, perm=None, accumulate_into=None):
    """accumulate_into = (dword, dtype_tab, dptab, dgamma, dbeta): add to these (tables_accumulate = accumulate = 1)"""
    H = word.shape[1]
    if perm is None:
        perm = word_perm(ids)
    if accumulate_into is not None:
        dword, dtype_tab, dptab, dg, db = accumulate_into
        ws = _ws(lib().nbest_embed_bwd_ws_bytes(B * S, H), word.device)
        check(lib().nbest_embed_ln_bwd(ptr(ids), ptr(seg), ptr(pos), ptr(perm), ptr(word), ptr(type_tab), ptr(ptab), ptr(gamma), ptr(stats),
                                       ptr(dout), ptr(dword), ptr(dtype_tab), ptr(dptab), ptr(dg), ptr(db), B, S, H,
                                       type_tab.shape[0], dtype_code(word.dtype), word_pad_id, pos_pad_id, 1, 1, drop_p, seed,
                                       drop_stream, ptr(ws), ws.numel(), stream_ptr()), "embed_ln_bwd")
        return dword, dtype_tab, dptab, dg, db
    dev = word.device
    dword = torch.zeros(word.shape, dtype=torch.float32, device=dev)
    dtype_tab = torch.zeros(type_tab.shape, dtype=torch.float32, device=dev)
    dptab = torch.zeros(ptab.shape, dtype=torch.float32, device=dev)
    dg = torch.zeros(H, dtype=torch.float32, device=dev)
    db = torch.zeros_like(dg)
    ws = _ws(lib().nbest_embed_bwd_ws_bytes(B * S, H), dev)
    check(lib().nbest_embed_ln_bwd(ptr(ids), ptr(seg), ptr(pos), ptr(perm), ptr(word), ptr(type_tab), ptr(ptab), ptr(gamma), ptr(stats),
                                   ptr(dout), ptr(dword), ptr(dtype_tab), ptr(dptab), ptr(dg), ptr(db), B, S, H,
                                   type_tab.shape[0], dtype_code(word.dtype), word_pad_id, pos_pad_id, 0, 0, drop_p, seed,
                                   drop_stream, ptr(ws), ws.numel(), stream_ptr()), "embed_ln_bwd")
    return dword, dtype_tab, dptab, dg, db


class DeviceLabelSpace:
    """Device copy of the STC label hierarchy in the layout nbest_label_space expects."""

    def __init__(self, labels, device):
        off, ids, head_row = [0], [], []
        row = labels.n_top
        for t in range(labels.n_top):
            bs = labels.top2bottom[t]
            ids += bs
            off.append(len(ids))
            if len(bs) >= 2:
                head_row.append(row)
                row += len(bs)
            else:
                head_row.append(-1)
        self.n_rows = row
        assert row == labels.n_head_rows
        self.labels = labels
        i32 = dict(dtype=torch.int32, device=device)
        self.bottom_off = torch.tensor(off, **i32)
        self.bottom_ids = torch.tensor(ids, **i32)
        self.head_row = torch.tensor(head_row, **i32)
        self.head_row_host = head_row
        self.none_flag = torch.tensor([1 if l.endswith("NONE") else 0 for l in labels.idx2label], dtype=torch.uint8, device=device)
        self.c = LabelSpaceC(labels.n_top, labels.n_bottom, row, self.bottom_off.data_ptr(), self.bottom_ids.data_ptr(),
                             self.head_row.data_ptr())


def heads_ws(B, R, H, device):
    """a PRIVATE workspace for stc_heads (the default is a shared scratch buffer): the autograd bridge keeps it until stc_heads_vjp"""
    return torch.empty(lib().nbest_heads_ws_bytes(B, R, H), dtype=torch.uint8, device=device)


def stc_heads_vjp(Wh, dls, top, bott, dtop, dbott, dfin, B, H, dWh, dbh, ws, accumulate=True, drop_p=0.0, seed=0, drop_stream=0):
    """d(CLS) [B, H] (and dWh / dbh, accumulated) for arbitrary upstream gradients of top / bottoms / final; ``ws``: the workspace of
    the stc_heads call that produced top / bott"""
    dcls = torch.empty(B, H, dtype=torch.float32, device=Wh.device)
    check(lib().nbest_stc_heads_vjp(ptr(Wh), C.byref(dls.c), ptr(top), ptr(bott), ptr(dtop), ptr(dbott), ptr(dfin), ptr(dcls), ptr(dWh),
                                    ptr(dbh), B, H, int(accumulate), drop_p, seed, drop_stream, ptr(ws), ws.numel(), stream_ptr()),
          "stc_heads_vjp")
    return dcls


def stc_heads(hidden, cls_stride, Wh, bh, dls, labels_f, B, H, need_grad=True, accumulate=False, drop_p=0.0, seed=0,
              drop_stream=0, dWh=None, dbh=None, ws=None):
    dev = Wh.device
    R, nt, nb = dls.n_rows, dls.labels.n_top, dls.labels.n_bottom
    f = dict(dtype=torch.float32, device=dev)
    top, bott, fin = torch.empty(B, nt, **f), torch.empty(B, R - nt, **f), torch.empty(B, nb, **f)
    loss = torch.empty(4, **f)
    dcls = torch.empty(B, H, **f) if need_grad else None
    if need_grad and dWh is None:
        dWh, dbh = torch.zeros(R, H, **f), torch.zeros(R, **f)
    if ws is None:
        ws = _ws(lib().nbest_heads_ws_bytes(B, R, H), dev)
    check(lib().nbest_stc_heads(ptr(hidden), cls_stride, ptr(Wh), ptr(bh), C.byref(dls.c), ptr(labels_f), ptr(top), ptr(bott),
                                ptr(fin), ptr(loss), ptr(dcls), ptr(dWh), ptr(dbh), B, H, dtype_code(hidden.dtype),
                                int(need_grad), int(accumulate), drop_p, seed, drop_stream, ptr(ws), ws.numel(), stream_ptr()),
          "stc_heads")
    return top, bott, fin, loss, dcls, dWh, dbh


def stc_decode(top, bott, dls, out=None):
    """``out``: int32 [>= B, n_top] rows to decode into instead of a fresh device tensor - a PINNED host tensor makes the kernel
    write the rows straight into host memory (include/nbest_hip.h nbest_stc_decode)"""
    B = top.shape[0]
    if out is None:
        pred = torch.empty(B, dls.labels.n_top, dtype=torch.int32, device=top.device)
    else:
        assert out.dtype == torch.int32 and out.is_contiguous() and out.shape[0] >= B and out.shape[1] == dls.labels.n_top
        assert out.is_cuda or out.is_pinned(), "stc_decode: out must be device memory or pinned host memory"
        pred = out[:B]
    check(lib().nbest_stc_decode(ptr(top), ptr(bott), C.byref(dls.c), ptr(dls.none_flag), ptr(pred), B, stream_ptr()), "stc_decode")
    return pred


def stream_stamp(flag, value):
    """*flag = value in stream order (flag: one int32 of device or PINNED host memory)"""
    assert flag.dtype == torch.int32 and flag.numel() == 1 and (flag.is_cuda or flag.is_pinned())
    check(lib().nbest_stream_stamp(ptr(flag), int(value), stream_ptr()), "stream_stamp")


AMAX_TENSOR_WORDS = 1024        # include/nbest_hip.h NBEST_AMAX_TENSOR_WORDS


def fp8_amax_fold(slots, out):
    """out[t] = max over the recording slots of tensor t; the slots are zeroed (nbest_fp8_amax_fold)"""
    assert slots.numel() == out.numel() * AMAX_TENSOR_WORDS and slots.dtype == out.dtype == torch.int32
    check(lib().nbest_fp8_amax_fold(ptr(slots), ptr(out), out.numel(), stream_ptr()), "fp8_amax_fold")


def cls_mse(hidden_a, stride_a, hidden_t, stride_t, B, H, da=None, dt=None, grad_scale=1.0):
    loss = torch.empty(1, dtype=torch.float32, device=hidden_a.device)
    check(lib().nbest_cls_mse(ptr(hidden_a), stride_a, ptr(hidden_t), stride_t, ptr(loss), ptr(da), ptr(dt), B, H,
                              dtype_code(hidden_a.dtype), grad_scale, stream_ptr()), "cls_mse")
    return loss


def cls_grad_scatter(dcls, B, S, H, dtype, out=None):
    dh = out if out is not None else torch.empty(B * S, H, dtype=dtype, device=dcls.device)
    check(lib().nbest_cls_grad_scatter(ptr(dcls), ptr(dh), B, S, H, dtype_code(dtype), stream_ptr()), "cls_grad_scatter")
    return dh


def cast_bf16(src, dst):
    check(lib().nbest_cast_f32_to_bf16(ptr(src), ptr(dst), src.numel(), stream_ptr()), "cast_f32_to_bf16")
