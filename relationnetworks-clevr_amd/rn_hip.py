"""ctypes binding of the C-ABI HIP library (include/rn_hip.h).

Thin by design: torch tensors only supply device pointers, strides and the
current HIP stream; every call is a plain `extern "C"` entry point.  Nothing
here falls back to PyTorch or the CPU: if librn_hip.so is missing or a call
fails, a RuntimeError is raised."""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librn_hip.so")
_PRODUCT_LIB_PATH = LIB_PATH                         # (tools may point LIB_PATH at a variant build)
RN_BF16, RN_F32, RN_F16, RN_FP8, RN_F32X3 = 0, 1, 2, 3, 4
RN_RELU, RN_ACCUMULATE = 1, 2
ABI_VERSION = 9

_lib = None

# name -> (restype, argtypes); mirrors include/rn_hip.h one to one
_P, _I, _L, _Z = C.c_void_p, C.c_int, C.c_long, C.c_size_t
SIGNATURES = {
    "rn_abi_version": (_I, []),
    "rn_workspace_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "rn_last_error": (C.c_char_p, []),
    "rn_stream_abandon_capture": (_I, [_P]),
    "rn_pair_build_fwd": (_I, [_P, _L, _L, _L, _P, _L, _P, _I, _I, _I, _I, _I, _I, _P]),
    "rn_qst_broadcast": (_I, [_P, _L, _P, _I, _I, _I, _I, _I, _I, _P]),
    "rn_pack_matrix": (_I, [_P, _L, _L, _I, _I, _P, _I, _I, _I, _P]),
    "rn_g_linear_fwd": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "rn_g_chain_rr_tile": (_I, []),
    "rn_pair_tables": (_I, [_P, _L, _L, _L, _P, _I, _P, _L, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "rn_g_chain_fwd_rr_f16s_alg0": (_I, [_P, _P, _I, _I, _P, _P, _I, _P, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P]),
    "rn_pair_sum_tiles": (_I, [_P, _P, _I, _I, _I, _P]),
    "rn_f_phi_fwd_from_partials": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "rn_g_chain_bwd_rr": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "rn_g_chain_bwd_rr_red_tpu": (_I, [_I, _I, _I]),
    "rn_g_chain_bwd_rr_red_whole": (_I, [_I, _I, _I, _I]),
    "rn_g_chain_bwd_rr_red": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _I, _P]),
    "rn_pair_reduce_parts": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "rn_pack_matrix_frag_many": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "rn_pair_sum_fwd": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P]),
    "rn_pair_sum_bwd": (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "rn_g_linear_bwd_dgrad": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "rn_g_linear_bwd_wgrad": (_I, [_P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "rn_wgrad_blocked_splits": (_I, [_I, _I, _I, _I]),
    "rn_wgrad_blocked_db_partials_offset": (_Z, [_I, _I, _I, _I, _I]),
    "rn_blocked_question_sums": (_I, [_P, _P, _I, _I, _P]),
    "rn_g_wgrad_blocked": (_I, [_P, _P, _P, _I, _P, _I, _I, _P, _P, _I, _P, _I, _P]),
    "rn_fp8_copy_health": (_I, [_P, _P, _P, _I, _P]),
    "rn_pair_reduce_bwd": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "rn_pair_dx_dq": (_I, [_P, _P, _P, _P, _P, _L, _L, _L, _I, _P, _I, _I, _I, _I, _I, _P]),
    "rn_wgrad0_from_reductions": (_I, [_P, _P, _P, _P, _L, _L, _L, _P, _I, _P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "rn_gemm_f32": (_I, [_P, _L, _L, _P, _L, _L, _P, _L, _I, _I, _I, _P, _P, _L, _P, _L, _I, _P]),
    "rn_pair_features": (_I, [_P, _I, _I, _P, _P, _P, _I, _I, _I, _P]),
    "rn_extract_features": (_I, [_P, _L, _L, _L, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _P]),
    "rn_f_phi_fwd": (_I, [_P] * 11 + [_I] * 6 + [_P]),
    "rn_f_phi_fwd_nll": (_I, [_P] * 14 + [_I] * 6 + [_P]),
    "rn_f_phi_fwd_bwd_from_partials": (_I, [_P, _I] + [_P] * 19 + [_I] * 5 + [_P]),
    "rn_f_phi_bwd_grads": (_I, [_P] * 10 + [_I] * 5 + [_P]),
    "rn_f_phi_split_ok": (_I, [_I] * 5),
    "rn_f_phi_split": (_I, [_P, _I] + [_P] * 18 + [_I] * 5 + [_P]),
    "rn_f_phi_split_status": (_I, [_P, _P]),
    "rn_f_phi_bwd_nll": (_I, [_P] * 18 + [_I] * 5 + [_P]),
    "rn_f_phi_bwd": (_I, [_P] * 17 + [_I] * 5 + [_P]),
    "rn_lstm_fwd": (_I, [_P] * 10 + [_I] * 5 + [_P]),
    "rn_lstm_bwd": (_I, [_P] * 5 + [_I] * 3 + [_P]),
    "rn_embedding_bwd": (_I, [_P] * 3 + [_I] * 4 + [_P]),
    "rn_nll_mean_fwd": (_I, [_P, _P, _P, _I, _I, _P]),
    "rn_nll_mean_bwd": (_I, [_P, _P, _P, _I, _I, _P]),
    "rn_clip_adam_chunk": (_I, []),
    "rn_lstm_bwd_tail": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "rn_clip_adam_step_dev": (_I, [_P, _I, _P, _P, _P, _L, _P, _P, _P, _P, _P]),
    "rn_copy_many": (_I, [_P, _P, _P, _I, _P]),
    "rn_dropout_mask": (_I, [_P, C.c_long, C.c_float, C.c_ulonglong, _P, _P]),
    "rn_clip_adam_step": (_I, [_P, _I, _P, _P, _P, _L, _P] + [C.c_float] * 7 + [_I, _P, _P]),
    "rn_conv3x3s2_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "rn_conv3x3s2_bwd_data": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "rn_conv3x3s2_bwd_weight": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "rn_bn_relu_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, _I, _I, _I, _P]),
    "rn_bn_relu_apply": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "rn_bn_relu_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "rn_bn_relu_bwd_conv_wgrad": (_I, [_P] * 13 + [_I, _I, _I, _I, _P]),
}
# diagnostics (include/rn_hip_debug.h): tests / tools only, not part of the product ABI
DEBUG_SIGNATURES = {
    "rn_relu_gate_image": (_I, [_P, _P, _I, _P]),
    "rn_rows_to_blocked": (_I, [_P, _P, _I, _I, _I, _P]),
    "rn_debug_stamp": (_I, [_P, _P]),
    "rn_probe_tr16": (_I, [_P, _P, _P]),
    "rn_probe_tr8": (_I, [_P, _P, _P]),
    "rn_probe_fp8_cvt": (_I, [_P, C.c_float, _P, _P, _P, _I, _P]),
    "rn_probe_mfma_stream": (_I, [_P, _I, _I, _I, _I, _P]),
    "rn_probe_mfma_stream_ops": (_I, [_P, _I, _I, _I, _I, _I, _P]),
    "rn_debug_wgrad_blocked_mix": (_I, [_I, _I, _I, _P, _P]),
    "rn_probe_red_schedule": (_I, [_I, _I, _I, _I, _I, _P, _I]),
    "rn_debug_f_phi_wide": (_I, [_I]),
    "rn_debug_gemm_small_below": (_I, [_I]),
}


# rn_workspace_bytes ops (include/rn_hip.h: RN_WS_*)
WS_RR_MASK = 0
WS_PAIR_SUM = 1
WS_WGRAD = 2
WS_WGRAD_BLOCKED = 3
WS_PAIR_REDUCE = 4
WS_WGRAD0 = 5
WS_PAIR_FEATURES = 6
WS_EXTRACT = 7
WS_F_PHI_BWD = 8
WS_F_PHI_NLL = 9
WS_CLIP_ADAM = 10
WS_CONV_BWD_WEIGHT = 11
WS_BN_RELU = 12
WS_F_PHI_SPLIT = 13


def workspace_bytes(op, a=0, b=0, c=0, d=0) -> int:
    """Bytes of scratch the entry point named by `op` needs for a shape (rn_workspace_bytes)."""
    return int(load().rn_workspace_bytes(op, int(a), int(b), int(c), int(d)))


def load(path: str | None = None):
    """dlopen librn_hip.so and attach the prototypes.  Raises loudly if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            "HIP extension %s not found: build it with `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % p)
    lib = C.CDLL(p)
    variant = os.path.abspath(p) != os.path.abspath(_PRODUCT_LIB_PATH)       # (tools/: a variant build for an A/B may predate a diagnostic entry point)
    for name, (res, args) in list(SIGNATURES.items()) + list(DEBUG_SIGNATURES.items()):
        if variant and name in DEBUG_SIGNATURES and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.restype, fn.argtypes = res, args
    v = lib.rn_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError("librn_hip.so ABI version %d != expected %d" % (v, ABI_VERSION))
    if path is None:
        _lib = lib
    return lib


def _check(rc: int, name: str):
    if rc != 0:
        msg = load().rn_last_error()
        raise RuntimeError("%s failed (rc=%d): %s" % (name, rc, msg.decode() if msg else "?"))


def stream_abandon_capture(stream) -> bool:
    """End (and throw away) whatever capture `stream` (a torch.cuda.Stream) is in -- see include/rn_hip.h.  -> True when the
    stream is not capturing afterwards."""
    return load().rn_stream_abandon_capture(C.c_void_p(stream.cuda_stream)) == 0


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


FP8_DTYPES = (torch.float8_e4m3fn, torch.uint8)      # the e4m3 copies of the stored activations (h_dtype / a_dtype = RN_FP8)


def _h_code(Hs):
    if Hs is None:
        return RN_BF16
    return RN_FP8 if any(h is not None and h.dtype in FP8_DTYPES for h in Hs) else RN_BF16


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def dtype_code(precision: str) -> int:
    """Storage dtype of activations / gradients.  "f16s" (fp16 tile x split fp16 weights forward) keeps
    bf16 storage for everything the backward pass touches."""
    return {"bf16": RN_BF16, "f16s": RN_BF16, "fp32": RN_F32, "bf16x3": RN_F32}[precision]      # ("bf16x3": fp32 storage, split-bf16 GEMMs)


def torch_dtype(code: int):
    return {RN_BF16: torch.bfloat16, RN_F32: torch.float32, RN_F16: torch.float16}[code]


def _dev(t, name):
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU (cuda:N = HIP device); got %s. The MI355X path has no CPU fallback."
                           % (name, t.device))


# ------------------------------------------------------------------ live kernel timing (bench.py)
class KernelTimer:
    """HIP-event brackets around selected entry points, recorded on the stream the kernels are
    launched on (torch's current stream).  Enabled only by bench.py; `summary()` synchronises."""

    def __init__(self):
        self.enabled = False
        self.pairs = {}

    def reset(self):
        self.pairs = {}

    def start(self, name):
        if not self.enabled:
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        self.pairs.setdefault(name, []).append((e0, e1))
        return e1

    def bracket_overhead_ms(self, n=50):
        """Duration an EMPTY event bracket reports (event processing on the stream): subtracted per bracket."""
        torch.cuda.synchronize()
        pairs = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in pairs)
        return ts[len(ts) // 2]

    def summary(self, steps=None):
        """{name: (brackets, total ms)} with the empty-bracket overhead removed from every bracket.  steps: the brackets cover
        that many identical steps -- the total is then steps x the MEDIAN per-step sum (one slow step, e.g. a clock dip
        right after the timed region, does not move it)."""
        torch.cuda.synchronize()
        ov = self.bracket_overhead_ms()
        out = {}
        for k, v in self.pairs.items():
            d = [max(a.elapsed_time(b) - ov, 0.0) for a, b in v]
            if steps and len(d) % steps == 0 and len(d) >= steps:
                per = len(d) // steps
                tot = sorted(sum(d[i * per:(i + 1) * per]) for i in range(steps))
                out[k] = (len(d), steps * tot[len(tot) // 2])
            else:
                out[k] = (len(d), sum(d))
        return out


TIMER = KernelTimer()


def _timed(name):
    def deco(fn):
        def wrapper(*a, **kw):
            e1 = TIMER.start(name)
            r = fn(*a, **kw)
            if e1 is not None:
                e1.record()
            return r
        wrapper.__name__ = fn.__name__
        return wrapper
    return deco


# ------------------------------------------------------------------ wrappers
def debug_stamp(buf, slot):
    """Store the device wall clock into buf[slot] (int64 tensor), in stream order (diagnostics only)."""
    _check(load().rn_debug_stamp(buf.data_ptr() + 8 * slot, _stream()), "rn_debug_stamp")


def probe_mfma_stream(out, workgroups, waves_per_simd, iters, dtype, zero_operands=False):
    """Launch the bare MFMA stream (diagnostics: the matrix pipe's sustained rate on this device); returns its flop count."""
    assert out.numel() >= workgroups * 256 * waves_per_simd and out.dtype == torch.float32
    _check(load().rn_probe_mfma_stream_ops(out.data_ptr(), workgroups, waves_per_simd, iters, {torch.float16: RN_F16, torch.bfloat16: RN_BF16}[dtype],
                                           int(zero_operands), _stream()), "rn_probe_mfma_stream")
    return workgroups * 4 * waves_per_simd * iters * 16 * 32768.0


@_timed("pair_build")
def pair_build_fwd(x, q, P, code, B, n, k, Q, ld):
    _dev(x, "x")
    sx = x.stride()
    _check(load().rn_pair_build_fwd(x.data_ptr(), sx[0], sx[1], sx[2], _ptr(q), q.stride(0) if q is not None else 0,
                                    P.data_ptr(), code, B, n, k, Q, ld, _stream()), "rn_pair_build_fwd")


def qst_broadcast(q, A, code, B, n, Q, col0, ld):
    _check(load().rn_qst_broadcast(q.data_ptr(), q.stride(0), A.data_ptr(), code, B, n, Q, col0, ld, _stream()),
           "rn_qst_broadcast")


def pack_matrix(src, sr, sc, R, Cc, dst, code, ld, Rpad, src_offset=0):
    _check(load().rn_pack_matrix(src.data_ptr() + 4 * src_offset, sr, sc, R, Cc, dst.data_ptr(), code, ld, Rpad, _stream()),
           "rn_pack_matrix")


@_timed("g_fwd")
def g_linear_fwd(A, lda, Wp, ldw, bias, H, ldh, code, M, N, K, h_offset_elems=0):
    esz = 2 if code == RN_BF16 else 4
    _check(load().rn_g_linear_fwd(A.data_ptr(), lda, Wp.data_ptr(), ldw, bias.data_ptr(), H.data_ptr() + esz * h_offset_elems,
                                  ldh, code, M, N, K, _stream()), "rn_g_linear_fwd")


def g_chain_rr_tile() -> int:
    return load().rn_g_chain_rr_tile()


def pack_matrix_frag_many(jobs):
    """jobs: list of (src, sr, sc, R, C, dst, natural) -- one launch."""
    n = len(jobs)
    _check(load().rn_pack_matrix_frag_many((C.c_void_p * n)(*[j[0].data_ptr() for j in jobs]), (C.c_long * n)(*[j[1] for j in jobs]),
                                           (C.c_long * n)(*[j[2] for j in jobs]), (C.c_int * n)(*[j[3] for j in jobs]),
                                           (C.c_int * n)(*[j[4] for j in jobs]), (C.c_void_p * n)(*[j[5].data_ptr() for j in jobs]),
                                           (C.c_int * n)(*[int(j[6]) for j in jobs]), n, _stream()), "rn_pack_matrix_frag_many")


@_timed("pair_build")
def pair_tables(x, q, W0T, b0, Xp, Vc, B, n, k, Q, N, coord=None):
    """Tables of the factored first layer: Xp (B*n, 64) bf16 object rows, Vc (B*n, N) fp32 bias rows (rn_pair_tables).
    coord (k - kf, n) fp32: x holds only the first kf = x.shape[2] columns, the coordinate tags come from the table."""
    _dev(x, "x")
    sx = x.stride()
    xdt = RN_F16 if Xp.dtype == torch.float16 else RN_BF16
    _check(load().rn_pair_tables(x.data_ptr(), sx[0], sx[1], sx[2], _ptr(coord), x.shape[2], _ptr(q), q.stride(0) if q is not None else 0,
                                 W0T.data_ptr(), b0.data_ptr(), Xp.data_ptr(), xdt, Vc.data_ptr(), B, n, k, Q, N, _stream()), "rn_pair_tables")


F16S_DITHER = 4      # tile-dithered hi images per g layer >= 1 (include/rn_hip.h, rn_g_chain_fwd_rr_f16s_alg0)


@_timed("g_fwd")
def g_chain_fwd_rr_f16s_alg0(Xp16, Vc, n, Whis, Wlo0, biases, Hs, masks, xg_part, M, G, Vq=None, inject=0, njp=None, gate=False):
    """f16s forward chain with the factored first layer (fp16 object rows, no pair matrix).  Whis[0] / Wlo0: the hi / lo images of
    layer 0; Whis[1..3]: (dither, 65536) fp16 tile-dithered hi images each.  Wlo0 a LIST of L lo images: the two-pass inference
    arithmetic (hi + lo on every layer, Whis = the plain hi images; no Hs / masks).  njp > n: padded j axis (M = B * n * njp, Xp16
    with a trailing zero row, two partial rows per tile in xg_part).  gate: the last layer's ReLU gate also goes into the sign bits
    of the e4m3 Hs[2] image (what relu_gate_image merges from masks[3]) -- the operand of the gate job of g_wgrad_blocked."""
    L = len(Whis)
    two = isinstance(Wlo0, (list, tuple))
    dither = 0 if two else Whis[1].numel() // 65536
    hp = (C.c_void_p * L)(*[w.data_ptr() for w in Whis])
    lp = (C.c_void_p * L)(*([w.data_ptr() for w in Wlo0] if two else [Wlo0.data_ptr()] + [None] * (L - 1)))
    bp = (C.c_void_p * L)(*[b.data_ptr() for b in biases])
    op = (C.c_void_p * L)(*[(h.data_ptr() if h is not None else None) for h in Hs]) if Hs is not None else None
    mp = (C.c_void_p * L)(*[m.data_ptr() for m in masks]) if masks is not None else None
    _check(load().rn_g_chain_fwd_rr_f16s_alg0(Xp16.data_ptr(), Vc.data_ptr(), n, njp or n, hp, lp, dither, bp, op, _h_code(Hs), mp, int(bool(gate)),
                                              xg_part.data_ptr(), _ptr(Vq), inject, M, L, G, _stream()), "rn_g_chain_fwd_rr_f16s_alg0")


def g_chain_rr_mask_bytes(M) -> int:
    return workspace_bytes(WS_RR_MASK, M, 0, 0, 0)


@_timed("g_dgrad")
def g_chain_bwd_rr(dxg, masks, Wtfs, dZs, M, rows_per_question, G):
    """Register-resident backward chain: dZs[0..3] from dxg, the forward masks and the transposed fragment images."""
    L = len(dZs)
    mp = (C.c_void_p * L)(*[m.data_ptr() for m in masks])
    wp = (C.c_void_p * (L - 1))(*[w.data_ptr() for w in Wtfs])
    zp = (C.c_void_p * L)(*[(z.data_ptr() if z is not None else None) for z in dZs])     # dZs[0] None: not stored
    _check(load().rn_g_chain_bwd_rr(dxg.data_ptr(), mp, wp, zp, M, rows_per_question, L, G, _stream()), "rn_g_chain_bwd_rr")


def g_chain_bwd_rr_red_tpu(M, n, njp=None):
    """Tiles per unit of the reducing backward chain for this shape (0: not supported -- store dZ_0 and use pair_reduce_bwd).
    njp > n: padded j axis (M = B * n * njp)."""
    return int(load().rn_g_chain_bwd_rr_red_tpu(M, n, njp or n))


def g_chain_bwd_rr_red_units(M, n, njp, tpu):
    """Units of the reducing backward chain: B * njp/32 * ceil(n/8) / tpu."""
    return (M // (n * njp)) * (njp // 32) * ((n + 7) // 8) // tpu


def g_chain_bwd_rr_red_whole(M, n, njp, tpu):
    """The library's split point of the balanced schedule: units below it run as a whole, the tiles of the others one by one."""
    w = int(load().rn_g_chain_bwd_rr_red_whole(M, n, njp or n, tpu))
    if w < 0:
        raise ValueError(f"rn_g_chain_bwd_rr_red_whole: shape not supported (M={M} n={n} njp={njp} tpu={tpu})")
    return w


def g_chain_bwd_rr_red_records(M, n, njp, tpu, whole=None):
    """Records (rows / 32 of rj_part) the chain leaves: one per whole unit, one per TILE of the units behind them."""
    u = g_chain_bwd_rr_red_units(M, n, njp, tpu)
    whole = u if whole is None else whole
    return u + (u - whole) * (tpu - 1)


@_timed("g_dgrad")
def g_chain_bwd_rr_red(dxg, masks, Wtfs, dZs, M, n, G, rj_part, ri_part, tpu, njp=None, whole=None):
    """Register-resident backward chain with the pair-axis reductions of layer 0's gradient formed on chip: dZs[0] None (gate job),
    dZs[1..2] row-blocked images, dZs[3] not written; partial sums to rj_part / ri_part (pair_reduce_parts adds them up)."""
    L = len(dZs)
    mp = (C.c_void_p * L)(*[m.data_ptr() for m in masks])
    wp = (C.c_void_p * (L - 1))(*[w.data_ptr() for w in Wtfs])
    zp = (C.c_void_p * L)(*[(z.data_ptr() if z is not None else None) for z in dZs])
    whole = g_chain_bwd_rr_red_units(M, n, njp or n, tpu) if whole is None else whole      # (None: no balanced tail)
    assert rj_part.numel() >= g_chain_bwd_rr_red_records(M, n, njp or n, tpu, whole) * 32 * G, "rj_part: one record per whole unit + one per tile behind them"
    _check(load().rn_g_chain_bwd_rr_red(dxg.data_ptr(), mp, wp, zp, M, n, njp or n, L, G, rj_part.data_ptr(), ri_part.data_ptr(), tpu, whole, _stream()),
           "rn_g_chain_bwd_rr_red")


@_timed("pair_reduce")
def pair_reduce_parts(rj_part, ri_part, Rj, Ri, Rq, B, n, G, nu, njp=None, tpu=1, whole=None):
    whole = B * ((njp or n) // 32) * nu if whole is None else whole
    _check(load().rn_pair_reduce_parts(rj_part.data_ptr(), ri_part.data_ptr(), Rj.data_ptr(), Ri.data_ptr(), _ptr(Rq), B, n, njp or n, G, nu, tpu, whole,
                                       _stream()), "rn_pair_reduce_parts")


@_timed("pair_sum")
def pair_sum_fwd(HL, ldh, xg, code, B, npairs, G):
    lib = load()
    nb = workspace_bytes(WS_PAIR_SUM, B, npairs, G, 0)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=xg.device)
    _check(lib.rn_pair_sum_fwd(HL.data_ptr(), ldh, xg.data_ptr(), ws.data_ptr(), code, B, npairs, G, _stream()), "rn_pair_sum_fwd")


@_timed("pair_sum_bwd")
def pair_sum_bwd(dxg, HL, ldh, dZ, lddz, code, B, npairs, G):
    _check(load().rn_pair_sum_bwd(dxg.data_ptr(), HL.data_ptr(), ldh, dZ.data_ptr(), lddz, code, B, npairs, G, _stream()),
           "rn_pair_sum_bwd")


@_timed("g_dgrad")
def g_linear_bwd_dgrad(dZ, lddz, Wt, ldwt, Hprev, ldhp, dZprev, lddzp, code, M, N, Kin):
    _check(load().rn_g_linear_bwd_dgrad(dZ.data_ptr(), lddz, Wt.data_ptr(), ldwt, Hprev.data_ptr(), ldhp, dZprev.data_ptr(),
                                        lddzp, code, M, N, Kin, _stream()), "rn_g_linear_bwd_dgrad")


@_timed("g_wgrad")
def g_linear_bwd_wgrad(dZ, lddz, A, lda, dW, db, code, M, N, K, Ktrue):
    """General weight gradient on ROW-MAJOR operands (any N % 256 == 0, K % 32 == 0; bf16 or fp32)."""
    lib = load()
    nb = workspace_bytes(WS_WGRAD, M, N, K, 0)
    if nb == 0:
        raise RuntimeError("rn_wgrad_ws_bytes: unsupported shape M=%d N=%d K=%d" % (M, N, K))
    ws = torch.empty(nb, dtype=torch.uint8, device=dW.device)
    _check(lib.rn_g_linear_bwd_wgrad(dZ.data_ptr(), lddz, A.data_ptr(), lda, dW.data_ptr(), _ptr(db),
                                     ws.data_ptr(), code, M, N, K, Ktrue, _stream()), "rn_g_linear_bwd_wgrad")


def wgrad_blocked_splits(M, rows_per_question=0, njobs=1, aligned=False):
    """Row splits per job of an rn_g_wgrad_blocked launch of `njobs` jobs on M pair rows (0: not covered)."""
    return load().rn_wgrad_blocked_splits(M, rows_per_question, njobs, int(aligned))


@_timed("pair_reduce")
def blocked_question_sums(img, Rq, M, rows_per_question):
    _check(load().rn_blocked_question_sums(img.data_ptr(), Rq.data_ptr(), M, rows_per_question, _stream()), "rn_blocked_question_sums")


def rows_to_blocked(src, back=False):
    """Row-major (M, 256) bf16 / e4m3 matrix <-> its row-blocked image (tests, tools)."""
    M = src.numel() // 256
    dst = torch.empty_like(src)
    _check(load().rn_rows_to_blocked(src.data_ptr(), dst.data_ptr(), RN_FP8 if src.dtype in FP8_DTYPES else RN_BF16, M, int(back), _stream()),
           "rn_rows_to_blocked")
    return dst


@_timed("g_wgrad")
def relu_gate_image(mask, img, M):
    """Layer-3 lane masks of the forward chain -> the sign bits of the e4m3 row-blocked image img (in place): the H_2 image of a
    gate job, as the forward chain writes it with gate=True."""
    _check(load().rn_relu_gate_image(mask.data_ptr(), img.data_ptr(), M, _stream()), "rn_relu_gate_image")
    return img


@_timed("g_wgrad")
def g_wgrad_blocked(jobs, M, dxg=None, rows_per_question=0, aligned=False, abl=0):
    """Weight gradients of the 256-wide g layers on the chains' row-blocked images, one launch for all `jobs`:
    jobs = [(dZ, A, dW, db), ...]; a job whose dZ is None is a gate job (the last layer: the gate in the sign bits of its e4m3 A image
    x dxg per question).
    aligned: question-aligned row splits (the db partials are then per-question sums of dZ).
    -> (ws, [db partials (Z, 4, 256) per job] when aligned, else []): the workspace must stay alive while the partials are in use."""
    lib = load()
    nj = len(jobs)
    nb = workspace_bytes(WS_WGRAD_BLOCKED, M, rows_per_question, nj, int(aligned))
    if nb == 0:
        raise RuntimeError("rn_g_wgrad_blocked: unsupported shape M=%d, %d jobs" % (M, nj))
    a8 = jobs[0][1].dtype in FP8_DTYPES
    ws = torch.empty(nb, dtype=torch.uint8, device=jobs[0][2].device)
    zp = (C.c_void_p * nj)(*[(j[0].data_ptr() if j[0] is not None else None) for j in jobs])
    zt = (C.c_int * nj)(*[(RN_FP8 if j[0] is None else RN_BF16) for j in jobs])
    ap = (C.c_void_p * nj)(*[j[1].data_ptr() for j in jobs])
    wp = (C.c_void_p * nj)(*[j[2].data_ptr() for j in jobs])
    bp = (C.c_void_p * nj)(*[j[3].data_ptr() for j in jobs])
    args = (zp, zt, ap, RN_FP8 if a8 else RN_BF16, _ptr(dxg), rows_per_question, int(aligned), wp, bp, nj, ws.data_ptr(), M, _stream())
    if abl:                                                   # RN_DIAG builds only (tools/): timing ablations, wrong results
        fn = lib.rn_diag_wgrad_blocked
        fn.restype, fn.argtypes = _I, SIGNATURES["rn_g_wgrad_blocked"][1] + [_I]
        _check(fn(*args, abl), "rn_diag_wgrad_blocked")
    else:
        _check(lib.rn_g_wgrad_blocked(*args), "rn_g_wgrad_blocked")
    parts = []
    if not aligned:                                           # (split counts per job kind are the library's own: wide / quad units)
        return ws, parts
    Z = lib.rn_wgrad_blocked_splits(M, rows_per_question, nj, int(aligned))
    for j in range(nj):
        off = lib.rn_wgrad_blocked_db_partials_offset(M, rows_per_question, nj, int(aligned), j) // 4
        parts.append(ws.view(torch.float32)[off: off + Z * 4 * 256].view(Z, 4, 256))
    return ws, parts


@_timed("pair_reduce")
def pair_reduce_bwd(dZ, lddz, Rj, Ri, Rq, code, B, n, G, njp=None):
    lib = load()
    ws = torch.empty(max(workspace_bytes(WS_PAIR_REDUCE, B, n, G, 0), 16), dtype=torch.uint8, device=dZ.device)
    _check(lib.rn_pair_reduce_bwd(dZ.data_ptr(), lddz, _ptr(Rj), _ptr(Ri), _ptr(Rq), ws.data_ptr(), code, B, n, njp or n, G, _stream()),
           "rn_pair_reduce_bwd")


@_timed("pair_sum")
def pair_sum_tiles(part, xg, M, rows_per_question, G):
    _check(load().rn_pair_sum_tiles(part.data_ptr(), xg.data_ptr(), M, rows_per_question, G, _stream()), "rn_pair_sum_tiles")


@_timed("pair_reduce")
def pair_dx_dq(Rj, Ri, Rq, W0, dx, dq, B, n, k, Q, N):
    """dx: any (B, n, kout <= k) tensor / view -- its strides say where the gradient goes (e.g. a permuted view of a
    (B, 24, n) buffer: the conv grid's own layout, coordinate columns dropped)."""
    sd = dx.stride()
    _check(load().rn_pair_dx_dq(Rj.data_ptr(), Ri.data_ptr(), _ptr(Rq), W0.data_ptr(), dx.data_ptr(), sd[0], sd[1], sd[2], dx.shape[2], _ptr(dq),
                                B, n, k, Q, N, _stream()), "rn_pair_dx_dq")


@_timed("g_wgrad0")
def wgrad0_from_reductions(Rj, Ri, Rq, x, q, dW0, db0, coord=None):
    B, n, kf = x.shape
    k = kf + (coord.shape[0] if coord is not None else 0)      # coord (k - kf, n): the coordinate tags are not part of x
    N, Q = Rj.shape[1], (q.shape[1] if q is not None else 0)      # q None: no question columns in layer 0 (Rq still gives db0)
    lib = load()
    ws = torch.empty(max(workspace_bytes(WS_WGRAD0, B, n, N, 0), 16), dtype=torch.uint8, device=Rj.device)
    sx = x.stride()
    _check(lib.rn_wgrad0_from_reductions(Rj.data_ptr(), Ri.data_ptr(), _ptr(Rq), x.data_ptr(), sx[0], sx[1], sx[2], _ptr(coord), kf, _ptr(q),
                                         q.stride(0) if q is not None else 0, dW0.data_ptr(), db0.data_ptr(), ws.data_ptr(), B, n, k, Q, N,
                                         _stream()), "rn_wgrad0_from_reductions")


def gemm_f32(A, sam, sak, Bm, sbk, sbn, Cm, ldc, M, N, K, bias=None, mul=None, ldmul=0, gate=None, ldgate=0, flags=0,
             a_off=0, b_off=0, c_off=0):
    """C[m,n] (+)= epi(sum_k A[a_off + m*sam + k*sak] * B[b_off + k*sbk + n*sbn]); offsets in elements."""
    _check(load().rn_gemm_f32(A.data_ptr() + 4 * a_off, sam, sak, Bm.data_ptr() + 4 * b_off, sbk, sbn,
                              Cm.data_ptr() + 4 * c_off, ldc, M, N, K, _ptr(bias), _ptr(mul), ldmul, _ptr(gate), ldgate,
                              flags, _stream()), "rn_gemm_f32")


# ------------------------------------------------------------------ conv stack: BatchNorm2d + ReLU
@_timed("bn_relu")
def bn_relu_fwd(x, y, gamma, beta, conv_bias, running_mean, running_var, num_batches, mean, invstd, eps, momentum):
    N, Cc, Hh, Ww = x.shape
    lib = load()
    ws = torch.empty(max(workspace_bytes(WS_BN_RELU, N, Cc, Hh * Ww, 0), 16), dtype=torch.uint8, device=x.device)
    _check(lib.rn_bn_relu_fwd(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(conv_bias), _ptr(running_mean),
                              _ptr(running_var), _ptr(num_batches), mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(), eps, momentum,
                              N, Cc, Hh * Ww, _stream()), "rn_bn_relu_fwd")


@_timed("bn_relu")
def bn_relu_apply(x, y, gamma, beta, mean, invstd):
    N, Cc, Hh, Ww = x.shape
    _check(load().rn_bn_relu_apply(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                   N, Cc, Hh * Ww, _stream()), "rn_bn_relu_apply")


@_timed("bn_relu")
def bn_relu_bwd(dy, x, dx, gamma, beta, mean, invstd, dgamma, dbeta, zero_out=None):
    """zero_out: C floats the same launch sets to zero (the conv-bias gradient)."""
    N, Cc, Hh, Ww = x.shape
    lib = load()
    ws = torch.empty(max(workspace_bytes(WS_BN_RELU, N, Cc, Hh * Ww, 0), 16), dtype=torch.uint8, device=x.device)
    _check(lib.rn_bn_relu_bwd(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
                              invstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), _ptr(zero_out), ws.data_ptr(), N, Cc, Hh * Ww, _stream()),
           "rn_bn_relu_bwd")


@_timed("conv")
def bn_relu_bwd_conv_wgrad(dy, xc, inp, gamma, beta, mean, invstd, dgamma, dbeta, dw, zero_out=None):
    """Batch-norm + ReLU backward of a block whose input needs no gradient, fused with its convolution's weight gradient
    (the conv output gradient stays in the kernel).  xc: the conv output; inp: the block's input."""
    N, Cin, Hh, Ww = inp.shape
    lib = load()
    ws_bn = torch.empty(max(workspace_bytes(WS_BN_RELU, N, 24, (Hh // 2) * (Ww // 2), 0), 16), dtype=torch.uint8, device=inp.device)
    ws_cv = torch.empty(max(workspace_bytes(WS_CONV_BWD_WEIGHT, N, Cin, Hh, Ww), 16), dtype=torch.uint8, device=inp.device)
    _check(lib.rn_bn_relu_bwd_conv_wgrad(dy.data_ptr(), xc.data_ptr(), inp.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
                                         invstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), _ptr(zero_out), dw.data_ptr(),
                                         ws_bn.data_ptr(), ws_cv.data_ptr(), N, Cin, Hh, Ww, _stream()), "rn_bn_relu_bwd_conv_wgrad")


# ------------------------------------------------------------------ fused f_phi
@_timed("f_phi")
def f_phi_fwd(xg, fw, fb, mask, f1, f2, out, transposed=False):
    """fw: the three weights as (out, in), or with transposed=True their (in, out) copies."""
    B, G = xg.shape
    F1, F2, A = (fw[0].shape[1], fw[1].shape[1], fw[2].shape[1]) if transposed else (fw[0].shape[0], fw[1].shape[0], fw[2].shape[0])
    _check(load().rn_f_phi_fwd(xg.data_ptr(), fw[0].data_ptr(), fb[0].data_ptr(), fw[1].data_ptr(), fb[1].data_ptr(), fw[2].data_ptr(),
                               fb[2].data_ptr(), _ptr(mask), f1.data_ptr(), f2.data_ptr(), out.data_ptr(), int(transposed), B, G, F1, F2, A,
                               _stream()), "rn_f_phi_fwd")


_NLL_WS = {}


def _nll_sync_ws(B, device):
    """The zeroed-once workspace of rn_f_phi_fwd_nll (block partials + completion counter), one per (device, B)."""
    key = (device, B)
    ws = _NLL_WS.get(key)
    if ws is None:
        ws = _NLL_WS[key] = torch.zeros(max(workspace_bytes(WS_F_PHI_NLL, B, 0, 0, 0), 16), dtype=torch.uint8, device=device)
    return ws


@_timed("f_phi")
def f_phi_fwd_nll(xg, fw, fb, mask, label, f1, f2, out, loss, transposed=False):
    """f_phi + log_softmax + mean NLL in one launch (rn_f_phi_fwd_nll); label: int64 (B,)."""
    B, G = xg.shape
    F1, F2, A = (fw[0].shape[1], fw[1].shape[1], fw[2].shape[1]) if transposed else (fw[0].shape[0], fw[1].shape[0], fw[2].shape[0])
    _check(load().rn_f_phi_fwd_nll(xg.data_ptr(), fw[0].data_ptr(), fb[0].data_ptr(), fw[1].data_ptr(), fb[1].data_ptr(), fw[2].data_ptr(),
                                   fb[2].data_ptr(), _ptr(mask), label.data_ptr(), f1.data_ptr(), f2.data_ptr(), out.data_ptr(),
                                   loss.data_ptr(), _nll_sync_ws(B, xg.device).data_ptr(), int(transposed), B, G, F1, F2, A, _stream()),
           "rn_f_phi_fwd_nll")


@_timed("f_phi")
def f_phi_fwd_from_partials(xg_part, parts_per_row, xg, fw, fb, mask, label, f1, f2, out, loss, transposed=False):
    """pair sum of the chain partials + f_phi + log_softmax (+ mean NLL when label is given) in one launch; xg (B, G) is written."""
    B, G = xg.shape
    F1, F2, A = (fw[0].shape[1], fw[1].shape[1], fw[2].shape[1]) if transposed else (fw[0].shape[0], fw[1].shape[0], fw[2].shape[0])
    ws = _nll_sync_ws(B, xg.device).data_ptr() if label is not None else None
    _check(load().rn_f_phi_fwd_from_partials(xg_part.data_ptr(), parts_per_row, xg.data_ptr(), fw[0].data_ptr(), fb[0].data_ptr(), fw[1].data_ptr(),
                                             fb[1].data_ptr(), fw[2].data_ptr(), fb[2].data_ptr(), _ptr(mask), _ptr(label), f1.data_ptr(),
                                             f2.data_ptr(), out.data_ptr(), _ptr(loss), ws, int(transposed), B, G, F1, F2, A, _stream()),
           "rn_f_phi_fwd_from_partials")


@_timed("f_phi")
def f_phi_fwd_bwd_from_partials(xg_part, parts_per_row, xg, fwT, fb, fw, mask, label, f1, f2, out, loss, dxg):
    """The training step's f_phi up to dxg in one launch (fwT: transposed forward weights, fw: the natural ones; d loss = 1).
    -> the dz workspace f_phi_bwd_grads finishes from."""
    B, G = xg.shape
    F1, F2, A = fw[0].shape[0], fw[1].shape[0], fw[2].shape[0]
    lib = load()
    ws = torch.empty(max(workspace_bytes(WS_F_PHI_BWD, B, F1, F2, A), 16), dtype=torch.uint8, device=xg.device)
    _check(lib.rn_f_phi_fwd_bwd_from_partials(xg_part.data_ptr(), parts_per_row, xg.data_ptr(), fwT[0].data_ptr(), fb[0].data_ptr(),
                                              fwT[1].data_ptr(), fb[1].data_ptr(), fwT[2].data_ptr(), fb[2].data_ptr(), fw[0].data_ptr(),
                                              fw[1].data_ptr(), fw[2].data_ptr(), _ptr(mask), label.data_ptr(), f1.data_ptr(), f2.data_ptr(),
                                              out.data_ptr(), loss.data_ptr(), _nll_sync_ws(B, xg.device).data_ptr(), ws.data_ptr(),
                                              dxg.data_ptr(), B, G, F1, F2, A, _stream()), "rn_f_phi_fwd_bwd_from_partials")
    return ws


_SPLIT_WS = []          # weak references to every sync workspace handed out (f_phi_split_status walks them)


def f_phi_split_sync_ws(device):
    """A zeroed sync workspace for rn_f_phi_split (epoch, completion counter, error word, stage flags).  ONE per caller that may
    launch concurrently with others (functional.PackedWeights keeps one per module): launches that share a workspace must be
    stream-ordered.  Allocate it OUTSIDE a hipGraph capture (a captured torch.zeros would re-zero it on every replay: harmless,
    but a memset node per step)."""
    import weakref
    ws = torch.zeros(max(workspace_bytes(WS_F_PHI_SPLIT, 0, 0, 0, 0), 16), dtype=torch.uint8, device=device)
    _SPLIT_WS.append(weakref.ref(ws))
    return ws


def f_phi_split_ok(B, G, F1, F2, A) -> bool:
    return bool(load().rn_f_phi_split_ok(B, G, F1, F2, A))


def f_phi_split_status(device=None) -> int:
    """0 when no launch of rn_f_phi_split (on `device`) ever gave up on a poll, else 1 + the stage (synchronises)."""
    worst = 0
    for ref in list(_SPLIT_WS):
        ws = ref()
        if ws is None:
            _SPLIT_WS.remove(ref)
        elif device is None or ws.device == torch.device(device):
            worst = max(worst, load().rn_f_phi_split_status(ws.data_ptr(), torch.cuda.current_stream(ws.device).cuda_stream))
    return worst


@_timed("f_phi")
def f_phi_split(xg_part, parts_per_row, xg, fw, fb, fwT, mask, label, f1, f2, out, loss, sync_ws, dxg=None):
    """f_phi as a feature-split fp32 MFMA chain in one launch (rn_f_phi_split): fw = the natural (out, in) weights, fwT the
    (in, out) copies (needed with dxg).  dxg given: the backward dz chain for d loss = 1 rides along -> the dz workspace."""
    B, G = xg.shape
    F1, F2, A = fw[0].shape[0], fw[1].shape[0], fw[2].shape[0]
    ws = None
    if dxg is not None:
        ws = torch.empty(max(workspace_bytes(WS_F_PHI_BWD, B, F1, F2, A), 16), dtype=torch.uint8, device=xg.device)
    _check(load().rn_f_phi_split(_ptr(xg_part), parts_per_row, xg.data_ptr(), fw[0].data_ptr(), fb[0].data_ptr(), fw[1].data_ptr(),
                                 fb[1].data_ptr(), fw[2].data_ptr(), fb[2].data_ptr(), _ptr(fwT[0]) if fwT else None,
                                 _ptr(fwT[1]) if fwT else None, _ptr(mask), _ptr(label), f1.data_ptr(), f2.data_ptr(), out.data_ptr(),
                                 _ptr(loss), _ptr(ws), _ptr(dxg), sync_ws.data_ptr(), B, G, F1, F2, A, _stream()),
           "rn_f_phi_split")
    return ws


@_timed("f_phi")
def f_phi_bwd_grads(ws, xg, f1, f2, dW, db):
    B, G = xg.shape
    F1, F2, A = dW[0].shape[0], dW[1].shape[0], dW[2].shape[0]
    _check(load().rn_f_phi_bwd_grads(ws.data_ptr(), xg.data_ptr(), f1.data_ptr(), f2.data_ptr(), dW[0].data_ptr(), db[0].data_ptr(),
                                     dW[1].data_ptr(), db[1].data_ptr(), dW[2].data_ptr(), db[2].data_ptr(), B, G, F1, F2, A, _stream()),
           "rn_f_phi_bwd_grads")


@_timed("f_phi")
def f_phi_bwd_nll(gloss, label, out, f2, f1, xg, fw, mask, dW, db, dxg):
    B, G = xg.shape
    F1, F2, A = fw[0].shape[0], fw[1].shape[0], fw[2].shape[0]
    lib = load()
    ws = torch.empty(max(workspace_bytes(WS_F_PHI_BWD, B, F1, F2, A), 16), dtype=torch.uint8, device=xg.device)
    _check(lib.rn_f_phi_bwd_nll(gloss.data_ptr(), label.data_ptr(), out.data_ptr(), f2.data_ptr(), f1.data_ptr(), xg.data_ptr(),
                                fw[0].data_ptr(), fw[1].data_ptr(), fw[2].data_ptr(), _ptr(mask), dW[0].data_ptr(), db[0].data_ptr(),
                                dW[1].data_ptr(), db[1].data_ptr(), dW[2].data_ptr(), db[2].data_ptr(), dxg.data_ptr(), ws.data_ptr(),
                                B, G, F1, F2, A, _stream()), "rn_f_phi_bwd_nll")


@_timed("f_phi")
def f_phi_bwd(gout, out, f2, f1, xg, fw, mask, dW, db, dxg):
    B, G = xg.shape
    F1, F2, A = fw[0].shape[0], fw[1].shape[0], fw[2].shape[0]
    lib = load()
    ws = torch.empty(max(workspace_bytes(WS_F_PHI_BWD, B, F1, F2, A), 16), dtype=torch.uint8, device=xg.device)
    _check(lib.rn_f_phi_bwd(gout.data_ptr(), out.data_ptr(), f2.data_ptr(), f1.data_ptr(), xg.data_ptr(), fw[0].data_ptr(), fw[1].data_ptr(),
                            fw[2].data_ptr(), _ptr(mask), dW[0].data_ptr(), db[0].data_ptr(), dW[1].data_ptr(), db[1].data_ptr(),
                            dW[2].data_ptr(), db[2].data_ptr(), dxg.data_ptr(), ws.data_ptr(), B, G, F1, F2, A, _stream()), "rn_f_phi_bwd")


# ------------------------------------------------------------------ clip + Adam on the flat gradient
def clip_adam_step(chunks, nchunks, g, m, v, ws, max_norm, lr, beta1, beta2, eps, wd, step, norm_out=None, grad_scale=1.0):
    _check(load().rn_clip_adam_step(chunks.data_ptr(), nchunks, g.data_ptr(), m.data_ptr(), v.data_ptr(), g.numel(), ws.data_ptr(),
                                    float(grad_scale), float(max_norm or 0.0), float(lr), float(beta1), float(beta2), float(eps), float(wd), int(step),
                                    _ptr(norm_out), _stream()), "rn_clip_adam_step")


def clip_adam_step_dev(chunks, nchunks, g, m, v, ws, hyper, step_dev, norm_out=None):
    """clip + Adam with the per-step scalars in device memory (hyper: 7 floats, step_dev: int32 count) -- capturable."""
    _check(load().rn_clip_adam_step_dev(chunks.data_ptr(), nchunks, g.data_ptr(), m.data_ptr(), v.data_ptr(), g.numel(), ws.data_ptr(),
                                        hyper.data_ptr(), step_dev.data_ptr(), _ptr(norm_out), _stream()), "rn_clip_adam_step_dev")


# ------------------------------------------------------------------ question encoder (embedding + LSTM)
@_timed("lstm")
def lstm_fwd(idx, emb, W_ih, W_hh, b_ih, b_hh, xs, gates, cs, hs):
    B, T = idx.shape
    _check(load().rn_lstm_fwd(idx.data_ptr(), emb.data_ptr(), W_ih.data_ptr(), W_hh.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr(),
                              _ptr(xs), _ptr(gates), _ptr(cs), hs.data_ptr(), B, T, emb.shape[0], emb.shape[1], W_hh.shape[1], _stream()),
           "rn_lstm_fwd")


@_timed("lstm")
def lstm_bwd(dhn, gates, cs, W_hh, dgates):
    T, B = gates.shape[0], gates.shape[1]
    _check(load().rn_lstm_bwd(dhn.data_ptr(), gates.data_ptr(), cs.data_ptr(), W_hh.data_ptr(), dgates.data_ptr(), B, T, W_hh.shape[1],
                              _stream()), "rn_lstm_bwd")


@_timed("lstm")
def embedding_bwd(idx, dx, demb):
    B, T = idx.shape
    _check(load().rn_embedding_bwd(idx.data_ptr(), dx.data_ptr(), demb.data_ptr(), B, T, demb.shape[0], demb.shape[1], _stream()),
           "rn_embedding_bwd")


@_timed("lstm")
def lstm_bwd_tail(idx, dx, demb, dgates, db_ih, db_hh=None):
    """Embedding gradient (demb may be None) and the bias gradients (column sums of dgates (T, B, 4H)) in one launch."""
    B, T = idx.shape
    _check(load().rn_lstm_bwd_tail(idx.data_ptr(), _ptr(dx), _ptr(demb), dgates.data_ptr(), db_ih.data_ptr(), _ptr(db_hh), B, T,
                                   demb.shape[0] if demb is not None else 0, demb.shape[1] if demb is not None else 32,
                                   dgates.shape[-1] // 4, _stream()), "rn_lstm_bwd_tail")


# ------------------------------------------------------------------ R-CBIR pair features (extract.py)
def pair_features(A, lda, F, code, B, npairs):
    """(maxf, avgf), each (B, F) fp32: L2-normalised pair rows of A[:, :F], max / mean over each question's pairs."""
    lib = load()
    maxf = torch.empty(B, F, dtype=torch.float32, device=A.device)
    avgf = torch.empty(B, F, dtype=torch.float32, device=A.device)
    ws = torch.empty(max(workspace_bytes(WS_PAIR_FEATURES, B, npairs, F, 0), 16), dtype=torch.uint8, device=A.device)
    _check(lib.rn_pair_features(A.data_ptr(), lda, F, maxf.data_ptr(), avgf.data_ptr(), ws.data_ptr(), code, B, npairs, _stream()),
           "rn_pair_features")
    return maxf, avgf


def fp8_copy_health(mask, img, M):
    """-> int64 tensor (4,) on the device: positive elements, flushed (positive but byte 0), clamped (byte 0x7e), largest byte of
    the e4m3 copy `img` of a swapped layer (0..2) whose lane masks are `mask` (rn_fp8_copy_health)."""
    out = torch.zeros(4, dtype=torch.int64, device=img.device)
    _check(load().rn_fp8_copy_health(mask.data_ptr(), img.data_ptr(), out.data_ptr(), M, _stream()), "rn_fp8_copy_health")
    return out


def extract_features(x, Wts, biases, per_q, F):
    """(maxf, avgf), each (B, F) fp32, of the input of g layer len(Wts) -- formed on chip from the objects x (B, n, k), never
    materialised (rn_extract_features).  Wts[l]: (K_l, 256) fp32 transposed weights; biases[l]: (256,) or, with per_q[l], (B, 256)."""
    lib = load()
    B, n, k = x.shape
    L = len(Wts)
    maxf = torch.empty(B, F, dtype=torch.float32, device=x.device)
    avgf = torch.empty(B, F, dtype=torch.float32, device=x.device)
    ws = torch.empty(max(workspace_bytes(WS_EXTRACT, B, n, F, 0), 16), dtype=torch.uint8, device=x.device)
    wp = (C.c_void_p * max(L, 1))(*[w.data_ptr() for w in Wts])
    bp = (C.c_void_p * max(L, 1))(*[b_.data_ptr() for b_ in biases])
    pq = (C.c_int * max(L, 1))(*[int(bool(v)) for v in per_q])
    sx = x.stride()
    _check(lib.rn_extract_features(x.data_ptr(), sx[0], sx[1], sx[2], wp, bp, pq, L, F, maxf.data_ptr(), avgf.data_ptr(), ws.data_ptr(),
                                   B, n, k, _stream()), "rn_extract_features")
    return maxf, avgf


# ------------------------------------------------------------------ conv stack: 3x3 / stride-2 convolutions
@_timed("conv")
def conv3x3s2_fwd(x, w, y):
    N, Cin, Hh, Ww = x.shape
    _check(load().rn_conv3x3s2_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), N, Cin, w.shape[0], Hh, Ww, _stream()), "rn_conv3x3s2_fwd")


@_timed("conv")
def conv3x3s2_bwd_data(dy, w, dx):
    N, Cin, Hh, Ww = dx.shape
    _check(load().rn_conv3x3s2_bwd_data(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), N, Cin, w.shape[0], Hh, Ww, _stream()),
           "rn_conv3x3s2_bwd_data")


@_timed("conv")
def conv3x3s2_bwd_weight(x, dy, dw):
    """dw (24, Cin, 3, 3) of the 3x3 / stride-2 / pad-1 convolution from x (N, Cin, H, W) and dy (N, 24, H/2, W/2)."""
    N, Cin, Hh, Ww = x.shape
    ws = torch.empty(max(workspace_bytes(WS_CONV_BWD_WEIGHT, N, Cin, Hh, Ww), 16), dtype=torch.uint8, device=x.device)
    _check(load().rn_conv3x3s2_bwd_weight(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), N, Cin, dy.shape[1], Hh, Ww, _stream()),
           "rn_conv3x3s2_bwd_weight")


# ------------------------------------------------------------------ mean NLL loss
def nll_mean_fwd(logp, label, loss):
    _check(load().rn_nll_mean_fwd(logp.data_ptr(), label.data_ptr(), loss.data_ptr(), logp.shape[0], logp.shape[1], _stream()), "rn_nll_mean_fwd")


def nll_mean_bwd(label, gloss, gout):
    _check(load().rn_nll_mean_bwd(label.data_ptr(), gloss.data_ptr(), gout.data_ptr(), gout.shape[0], gout.shape[1], _stream()), "rn_nll_mean_bwd")


def dropout_mask(mask, p, seed, state):
    """mask (fp32, contiguous) <- keep ? 1 / (1 - p) : 0 from the library's counter-based generator; state: int64 (2,) device tensor
    {draws so far, 0}, advanced by the launch (rn_dropout_mask)."""
    _dev(mask, "mask"); _dev(state, "state")
    if not (mask.is_contiguous() and mask.dtype == torch.float32 and state.dtype == torch.int64 and state.numel() >= 2 and state.is_contiguous()):
        raise ValueError("dropout_mask: mask must be contiguous fp32, state a contiguous int64 (2,) tensor")
    _check(load().rn_dropout_mask(mask.data_ptr(), mask.numel(), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, state.data_ptr(), _stream()), "rn_dropout_mask")
    return mask


def copy_many(pairs):
    """[(dst, src), ...] (<= 4 pairs of contiguous device tensors of equal byte size): ONE launch instead of one library copy each --
    the batch hand-off in front of a captured step."""
    n = len(pairs)
    for d, s_ in pairs:
        if not (d.is_contiguous() and s_.is_contiguous() and d.dtype == s_.dtype and d.numel() == s_.numel() and d.is_cuda and s_.is_cuda):
            raise ValueError("copy_many: tensors must be contiguous device tensors of the same type and size")
    dst = (C.c_void_p * n)(*[d.data_ptr() for d, _ in pairs])
    src = (C.c_void_p * n)(*[s_.data_ptr() for _, s_ in pairs])
    nb = (C.c_size_t * n)(*[d.numel() * d.element_size() for d, _ in pairs])
    _check(load().rn_copy_many(dst, src, nb, n, _stream()), "rn_copy_many")
