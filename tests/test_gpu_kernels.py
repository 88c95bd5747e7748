"""GPU: every HIP kernel against the CPU oracle, called through the C-ABI (ctypes) wrappers.

Tolerances (written here, per ③ of the build contract):
  * pure data movement (pair build, question broadcast, pack, pair-sum backward): bit-exact
    against the oracle rounded to the storage dtype (RNE);
  * fp32 storage / fp32 MFMA: <= 1e-4 relative (summation-order noise only; measured <= 4e-5);
  * bf16 storage / bf16 MFMA: compared with an oracle evaluated on the SAME bf16-rounded
    operands, so only accumulation order and the final RNE differ: <= 1 bf16 ulp (2^-7 rel).
"""
import os

import numpy as np
import pytest
import torch

from oracle import formula, rn_oracle as O

pytestmark = pytest.mark.gpu

BF16_ULP = 2.0 ** -7
F32_TOL = 1e-4


@pytest.fixture(scope="module")
def H():
    import relationnetworks_clevr_amd as pkg
    pkg.rn_hip.load()
    torch.cuda.set_device(0)
    return pkg.rn_hip


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).bfloat16().float().numpy()


def tdt(code):
    return torch.bfloat16 if code == 0 else torch.float32


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def unblock(t):
    """Row-blocked image (include/rn_hip.h: what the register-resident chains write for the weight gradient) -> the row-major
    (M, 256) matrix, in plain torch: 16-bit element (m, f) sits at ((m / 8) * 256 + f) * 8 + m % 8, e4m3 byte (m, f) at
    ((m / 16) * 256 + f) * 16 + m % 16."""
    if t is None:
        return None
    M = t.numel() // 256
    rb = 8 if t.element_size() == 2 else 16
    raw = t.contiguous().view(torch.int16 if rb == 8 else torch.uint8)
    return raw.view(M // rb, 256, rb).permute(0, 2, 1).reshape(M, 256).contiguous().view(t.dtype)


def unblock_h(Hs, upto=3):
    """H_0..2 (or dZ of layers 3..1) of a chain call are row-blocked images; a stored fourth entry is row-major."""
    return None if Hs is None else [unblock(h) if i < upto else h for i, h in enumerate(Hs)]


F16S_V = 4


def dither_images(W, V=F16S_V):
    """The V tile-dithered fp16 hi images of a weight matrix (include/rn_hip.h, rn_g_chain_fwd_rr_f16s): image d =
    fp16(W + ((d + 1/2) / V - 1/2) ulp_fp16(W)), as float32 arrays -- the test's own restatement of the pack kernel."""
    W = np.asarray(W, np.float32)
    a = np.abs(W)
    e = np.where(a >= 2.0 ** -14, np.floor(np.log2(np.maximum(a, 1e-30))), -14.0)
    ulp = (2.0 ** (e - 10)).astype(np.float32)
    return [(W + np.float32((d + 0.5) / V - 0.5) * ulp).astype(np.float32).astype(np.float16).astype(np.float32) for d in range(V)]


def f16s_images(H, wd, kt, k0img, L=4, G=256, V=F16S_V):
    """(his, lo0, jobs): fragment-major images of the f16s chains for device weights wd -- layer 0 hi / lo (natural K order, k0img
    columns), layers 1..3 V dithered hi images each."""
    his = [torch.empty(65536, dtype=torch.float16, device="cuda")] + [torch.empty(V, 65536, dtype=torch.float16, device="cuda") for _ in range(1, L)]
    lo0 = torch.empty(65536, dtype=torch.float16, device="cuda")
    jobs = [(wd[0], kt, 1, G, k0img, his[0], 4 | 1), (wd[0], kt, 1, G, k0img, lo0, 8 | 1)]
    jobs += [(wd[l], G, 1, G, G, his[l], 4 | (V << 8)) for l in range(1, L)]
    return his, lo0, jobs


def test_dithered_hi_images(H):
    """rn_pack_matrix_frag_many mode 4 | V << 8: the V images are the fragment packing of the dithered roundings, their mean is
    within ulp / (2 V) (+ an fp16 rounding of slack) of the fp32 weight, and every image is within one ulp of it."""
    G, V = 256, F16S_V
    W = formula.hash_uniform((G, G), 731, -0.2, 0.2).astype(np.float32)
    W[0, :8] = [0.0, 2.0 ** -15, -2.0 ** -14, 1.0, -0.5, 3e-6, 65000.0, 0.1]
    dst = torch.empty(V, 65536, dtype=torch.float16, device="cuda")
    H.pack_matrix_frag_many([(dev(W), G, 1, G, G, dst, 4 | (V << 8))])
    torch.cuda.synchronize()
    imgs = dither_images(W, V)
    for d in range(V):
        assert np.array_equal(dst[d].float().cpu().numpy(), frag_pack_ref(imgs[d], 0)), d
    a = np.abs(W)
    ulp = 2.0 ** (np.where(a >= 2.0 ** -14, np.floor(np.log2(np.maximum(a, 1e-30))), -14.0) - 10)
    mean = np.mean([im.astype(np.float64) for im in imgs], axis=0)
    assert (np.abs(mean - W) <= ulp * (0.5 / V + 0.26)).all()
    assert np.percentile(np.abs(mean - W) / ulp, 99) <= 0.5 / V + 1e-6          # (the slack above is for binade edges only)
    for im in imgs:
        assert (np.abs(im - W) <= ulp * 1.0).all()


def test_rows_to_blocked(H):
    for dt in (torch.bfloat16, torch.float8_e4m3fn):
        M = 16 * 37
        a = (torch.rand(M, 256, device="cuda") * 4).to(dt)
        img = H.rows_to_blocked(a)
        torch.cuda.synchronize()
        assert torch.equal(unblock(img).view(torch.uint8), a.view(torch.uint8))
        assert torch.equal(H.rows_to_blocked(img, back=True).view(torch.uint8), a.view(torch.uint8))


# ----------------------------------------------------------------------------- probe
def test_probe_tr16_mapping(H):
    """ds_read_b64_tr_b16 on a linear image (lane l supplies &lds[4l]): within each 16-lane block,
    lane i receives column i of the 4x16 block formed by the 16 lanes' 8-byte pieces."""
    src = np.arange(4096, dtype=np.uint16)
    out = torch.zeros(256, dtype=torch.int16, device="cuda")
    rc = H.load().rn_probe_tr16(dev(src.view(np.int16)).data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint16).reshape(64, 4)
    exp = np.zeros((64, 4), dtype=np.uint16)
    for lane in range(64):
        blk, i = lane // 16, lane % 16
        for j in range(4):       # row j of the block, column i; linear image: block blk = elements 64*blk..+64, row stride 16
            exp[lane, j] = 64 * blk + 16 * j + i
    print("tr16 lanes 0..3:", got[:4].tolist(), "lane 17:", got[17].tolist())
    assert np.array_equal(got, exp), got[:20]


def test_probe_tr8_mapping(H):
    """ds_read_b64_tr_b8 on a linear image (lane l supplies &lds[8l]): within each 16-lane block the 16 lanes' 8-byte pieces
    form an 8-row x 16-byte block and lane i receives column i (8 rows, ascending) -- what the e4m3 wgrad operand relies on."""
    src = (np.arange(4096) % 251).astype(np.uint8)
    out = torch.zeros(512, dtype=torch.uint8, device="cuda")
    rc = H.load().rn_probe_tr8(dev(src).data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(64, 8)
    exp = np.zeros((64, 8), dtype=np.uint8)
    for lane in range(64):
        blk, i = lane // 16, lane % 16
        for j in range(8):
            exp[lane, j] = src[128 * blk + 16 * j + i]
    print("tr8 lanes 0..2:", got[:3].tolist(), "lane 17:", got[17].tolist())
    assert np.array_equal(got, exp), got[:20]


@pytest.mark.parametrize("scale", [1.0, 4.0, 0.25])
def test_probe_fp8_conversions(H, scale):
    """The e4m3 conversions of the kernels (v_cvt_scalef32_pk_fp8_bf16 / _f16 / v_cvt_scalef32_pk_bf16_fp8): OCP e4m3fn,
    round to nearest even of value / scale -- the raw instruction turns what rounds beyond 448 into the NaN byte, the kernels'
    helpers (scale 1, non-negative groups of four) clamp to 448 first; the up-conversion returns byte value * scale exactly."""
    vals = np.concatenate([formula.hash_uniform((4096,), 77, 0, 12), formula.hash_uniform((2048,), 78, 0, 0.05), formula.hash_uniform((1024,), 79, -3, 0),
                           np.array([0.0, 448.0, 449.0, 1000.0, 6e4, 2.0 ** -9, 2.0 ** -10, 3 * 2.0 ** -11, 1.0625, 1.1875, 0.9375, 17.0], np.float32)]).astype(np.float32)
    vals = (vals * scale).astype(np.float32)
    n = vals.size
    o_bf = torch.zeros(n, dtype=torch.uint8, device="cuda"); o_h = torch.zeros(n, dtype=torch.uint8, device="cuda")
    back = torch.zeros(n, device="cuda")
    rc = H.load().rn_probe_fp8_cvt(dev(vals).data_ptr(), scale, o_bf.data_ptr(), o_h.data_ptr(), back.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    tv = torch.from_numpy(vals)
    grp_clamped = torch.from_numpy(np.repeat((vals.reshape(-1, 4) >= 0).all(1) & (scale == 1.0), 4))
    def ref8(t16):
        v = t16.float() / scale
        r = v.clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
        over = (v.abs() > 464) & ~grp_clamped                        # 464 = the tie between 448 and the next (non-existent) value
        return torch.where(over, torch.full_like(r, 0x7f) | (r & 0x80), r)
    for name, got, t16 in (("bf16", o_bf, tv.bfloat16()), ("f16", o_h, tv.half())):
        exp = ref8(t16).numpy()
        g = got.cpu().numpy()
        bad = g != exp
        # -0 vs +0 is not a difference that matters
        bad &= ~(((g & 0x7f) == 0) & ((exp & 0x7f) == 0))
        assert not bad.any(), (name, vals[bad][:8], g[bad][:8], exp[bad][:8])
    ok = torch.from_numpy((o_bf.cpu().numpy() & 0x7f) != 0x7f)
    exp_back = o_bf.cpu().view(torch.float8_e4m3fn).float() * scale
    assert torch.equal(back.cpu()[ok], exp_back[ok])


# ----------------------------------------------------------------------------- K1
@pytest.mark.parametrize("code", [0, 1])
@pytest.mark.parametrize("B,n,k,Q,strided", [(3, 64, 26, 128, True), (3, 64, 26, 0, False), (4, 12, 7, 256, False),
                                            (2, 196, 26, 128, True), (1, 5, 3, 8, False)])
def test_pair_build_exact(H, code, B, n, k, Q, strided):
    x = formula.hash_uniform((B, n, k), 7, -2, 2)
    q = formula.hash_uniform((B, max(Q, 1)), 8, -1, 1)
    ld = (2 * k + Q + 63) // 64 * 64
    if strided:      # physical (B,k,n) viewed as (B,n,k), like RN.forward's permute (model.py:200-201)
        xt = dev(x.transpose(0, 2, 1)).permute(0, 2, 1)
        assert not xt.is_contiguous()
    else:
        xt = dev(x)
    P = torch.full((B * n * n, ld), 7.0, dtype=tdt(code), device="cuda")
    H.pair_build_fwd(xt, dev(q) if Q else None, P, code, B, n, k, Q, ld)
    torch.cuda.synchronize()
    ref = np.zeros((B * n * n, ld), np.float32)
    ref[:, : 2 * k + Q] = O.pair_matrix(x, q[:, :Q] if Q else None)
    if code == 0:
        ref = bf16_round(ref)
    assert np.array_equal(P.float().cpu().numpy(), ref)


@pytest.mark.parametrize("code", [0, 1])
def test_qst_broadcast_exact(H, code):
    B, n, Q, col0, ld = 3, 12, 128, 256, 384
    q = formula.hash_uniform((B, Q), 9)
    A = torch.zeros(B * n * n, ld, dtype=tdt(code), device="cuda")
    H.qst_broadcast(dev(q), A, code, B, n, Q, col0, ld)
    ref = np.zeros((B * n * n, ld), np.float32)
    ref[:, col0:col0 + Q] = np.repeat(q, n * n, axis=0)
    if code == 0:
        ref = bf16_round(ref)
    assert np.array_equal(A.float().cpu().numpy(), ref)


@pytest.mark.parametrize("code", [0, 1])
def test_pack_matrix(H, code):
    N, K, ld = 256, 180, 192
    w = formula.hash_uniform((N, K), 10)
    wp = torch.empty(N, ld, dtype=tdt(code), device="cuda")
    H.pack_matrix(dev(w), K, 1, N, K, wp, code, ld, N)
    ref = np.zeros((N, ld), np.float32); ref[:, :K] = w
    wt = torch.empty(128, N, dtype=tdt(code), device="cuda")        # transposed, first 128 input columns
    H.pack_matrix(dev(w), 1, K, 128, N, wt, code, N, 128)
    reft = w[:, :128].T.copy()
    if code == 0:
        ref, reft = bf16_round(ref), bf16_round(reft)
    assert np.array_equal(wp.float().cpu().numpy(), ref)
    assert np.array_equal(wt.float().cpu().numpy(), reft)


# ----------------------------------------------------------------------------- K2 + dgrad
def _gemm_case(M, N, K, Ktrue, seed):
    A = np.zeros((M, K), np.float32); A[:, :Ktrue] = formula.hash_uniform((M, Ktrue), seed, -1, 1)
    W = np.zeros((N, K), np.float32); W[:, :Ktrue] = formula.hash_uniform((N, Ktrue), seed + 1, -0.2, 0.2)
    b = formula.hash_uniform((N,), seed + 2, -0.5, 0.5)
    return A, W, b


@pytest.mark.parametrize("code", [0, 1])
@pytest.mark.parametrize("M,N,K,Ktrue", [(128, 256, 192, 180), (576, 512, 320, 270), (1000, 256, 256, 256), (4096, 256, 64, 52),
                                          (131200, 256, 64, 52), (200, 320, 128, 128)])
def test_g_linear_fwd(H, code, M, N, K, Ktrue):
    A, W, b = _gemm_case(M, N, K, Ktrue, 20)
    if code == 0:
        A, W = bf16_round(A), bf16_round(W)
    ref = np.maximum(A.astype(np.float64) @ W.astype(np.float64).T + b, 0)
    out = torch.full((M, N), -3.0, dtype=tdt(code), device="cuda")
    H.g_linear_fwd(dev(A).to(tdt(code)), K, dev(W).to(tdt(code)), K, dev(b), out, N, code, M, N, K)
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    tol = BF16_ULP if code == 0 else F32_TOL
    err = np.abs(got - ref) / np.maximum(np.abs(ref), np.abs(ref).max() * 1e-2)
    assert err.max() <= tol, (err.max(), np.unravel_index(err.argmax(), err.shape))


@pytest.mark.parametrize("M,N,K", [(576, 512, 512), (9216, 512, 64), (70000, 512, 512), (300, 256, 192)])
def test_g_linear_split_bf16_arithmetic(H, M, N, K):
    """RN_F32X3: fp32 operands split into hi + lo bf16 as they are staged, hi*hi + hi*lo + lo*hi on the bf16 pipe, fp32 accumulate
    (both workgroup tiles) -- forward and gated dgrad against fp64: 2^-16 of a product is dropped, so 3e-5 of the largest output
    bounds it with room (measured ~4e-6); the exact-fp32 mode on the same operands sits at 1e-6."""
    A, W, b = _gemm_case(M, N, K, K, 60)
    ref = np.maximum(A.astype(np.float64) @ W.astype(np.float64).T + b, 0)
    out = torch.full((M, N), -3.0, device="cuda")
    H.g_linear_fwd(dev(A), K, dev(W), K, dev(b), out, N, H.RN_F32X3, M, N, K)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    e = np.abs(got - ref).max() / np.abs(ref).max()
    assert e <= 3e-5, e
    gate = np.maximum(formula.hash_uniform((M, N), 61, -1, 1), 0)
    H.g_linear_bwd_dgrad(dev(A), K, dev(W), K, dev(gate), N, out, N, H.RN_F32X3, M, K, N)
    ref = (A.astype(np.float64) @ W.astype(np.float64).T) * (gate > 0)
    got = out.cpu().numpy()
    assert np.abs(got - ref).max() / np.abs(ref).max() <= 3e-5 and np.all(got[gate <= 0] == 0)


@pytest.mark.parametrize("code", [0, 1])
def test_g_linear_tiles_agree_bitwise(H, code):
    """rn_gemm.hip picks 64 x 64 workgroup tiles for short matrices (the state-description models: M = B * 144) and 128 x 256 ones
    when those fill the chip; every output element's k order is the same in both, so the first rows of a long matrix (big tiles)
    equal the same rows computed as a short matrix (small tiles) bit for bit -- forward and gated dgrad."""
    Ms, Mb, N, K = 576, 128 * 512 + 40, 512, 512                   # 513 x 2 big tiles = 1026 workgroups >= GEMM_SMALL_BELOW
    A, W, b = _gemm_case(Mb, N, K, K, 50)
    Ad, Wd, bd = dev(A).to(tdt(code)), dev(W).to(tdt(code)), dev(b)
    big = torch.empty(Mb, N, dtype=tdt(code), device="cuda")
    small = torch.empty(Ms, N, dtype=tdt(code), device="cuda")
    H.g_linear_fwd(Ad, K, Wd, K, bd, big, N, code, Mb, N, K)
    H.g_linear_fwd(Ad, K, Wd, K, bd, small, N, code, Ms, N, K)
    assert torch.equal(big[:Ms], small) and small.float().abs().sum() > 0
    gate = dev(np.maximum(formula.hash_uniform((Mb, N), 51, -1, 1), 0)).to(tdt(code))
    H.g_linear_bwd_dgrad(Ad, K, Wd, K, gate, N, big, N, code, Mb, K, N)
    H.g_linear_bwd_dgrad(Ad, K, Wd, K, gate, N, small, N, code, Ms, K, N)
    assert torch.equal(big[:Ms], small) and small.float().abs().sum() > 0


@pytest.mark.parametrize("code", [0, 1])
def test_g_linear_fwd_wide_output(H, code):
    """layer output written into a wider buffer (ld 384) ahead of an injected question (ir-fp)."""
    M, N, K = 256, 256, 256
    A, W, b = _gemm_case(M, N, K, K, 30)
    if code == 0:
        A, W = bf16_round(A), bf16_round(W)
    buf = torch.full((M, 384), 5.0, dtype=tdt(code), device="cuda")
    H.g_linear_fwd(dev(A).to(tdt(code)), K, dev(W).to(tdt(code)), K, dev(b), buf, 384, code, M, N, K)
    got = buf.float().cpu().numpy()
    ref = np.maximum(A.astype(np.float64) @ W.astype(np.float64).T + b, 0)
    assert rel(got[:, :N], ref) <= (BF16_ULP if code == 0 else F32_TOL)
    assert np.all(got[:, N:] == 5.0)


@pytest.mark.parametrize("code", [0, 1])
@pytest.mark.parametrize("M,N,Kin", [(384, 256, 256), (700, 512, 512)])
def test_g_linear_bwd_dgrad(H, code, M, N, Kin):
    dZ = formula.hash_uniform((M, N), 40, -1, 1)
    W = formula.hash_uniform((N, Kin), 41, -0.2, 0.2)               # nn.Linear (out=N, in=Kin)
    Hp = np.maximum(formula.hash_uniform((M, Kin), 42, -1, 1), 0)
    if code == 0:
        dZ, W, Hp = bf16_round(dZ), bf16_round(W), bf16_round(Hp)
    ref = (dZ.astype(np.float64) @ W.astype(np.float64)) * (Hp > 0)
    Wt = np.ascontiguousarray(W.T)                                  # (Kin, N)
    out = torch.empty(M, Kin, dtype=tdt(code), device="cuda")
    H.g_linear_bwd_dgrad(dev(dZ).to(tdt(code)), N, dev(Wt).to(tdt(code)), N, dev(Hp).to(tdt(code)), Kin, out, Kin, code, M, N, Kin)
    got = out.float().cpu().numpy()
    tol = BF16_ULP if code == 0 else F32_TOL
    err = np.abs(got - ref) / np.maximum(np.abs(ref), np.abs(ref).max() * 1e-2)
    assert err.max() <= tol, err.max()
    assert np.all(got[Hp <= 0] == 0)


def pack_frag(H, src, sr, sc, R, Cc, dst, natural):
    """one fragment-major image (rn_pack_matrix_frag_many with a single job)"""
    H.pack_matrix_frag_many([(src, sr, sc, R, Cc, dst, natural)])


def frag_pack_ref(W, natural):
    """numpy statement of rn_pack_matrix_frag (include/rn_hip.h)."""
    R, Cc = W.shape
    out = np.zeros((8, 16, 64, 8), np.float32)
    for ks in range(16):
        for lane in range(64):
            h, m = lane >> 5, lane & 31
            for e in range(8):
                kidx = 16 * ks + 8 * h + e if natural else 32 * (ks >> 1) + 4 * h + 8 * (2 * (ks & 1) + (e >> 2)) + (e & 3)
                if kidx < Cc:
                    out[:, ks, lane, e] = W[m:256:32, kidx] if R == 256 else [W[32 * ob + m, kidx] if 32 * ob + m < R else 0 for ob in range(8)]
    return out.reshape(-1)


@pytest.mark.parametrize("natural", [1, 0])
def test_pack_matrix_frag(H, natural):
    W = bf16_round(formula.hash_uniform((256, 180 if natural else 256), 330, -1, 1))
    dst = torch.empty(65536, dtype=torch.bfloat16, device="cuda")
    pack_frag(H, dev(W), W.shape[1], 1, 256, W.shape[1], dst, natural)
    assert np.array_equal(dst.float().cpu().numpy(), frag_pack_ref(W, natural))


def rr_mask_decode(buf, M, l):
    """lane masks of rn_g_chain_fwd_rr (include/rn_hip.h) -> boolean (M, 256) gate matrix.
    layers 0..2: mask i, lane -> row 32 wt + lane % 32, feature 32 ob + 8 (i / 4) + 4 (lane / 32) + i % 4;
    layer 3 (operands un-swapped): mask i, lane -> row 32 wt + 8 (i / 4) + 4 (lane / 32) + i % 4, feature 32 ob + lane % 32."""
    m = buf.cpu().numpy().view(np.uint64).reshape(M // 32, 8, 16)
    bits = ((m[..., None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).astype(bool)    # (wt, ob, i, lane)
    i = np.arange(16)[:, None]; lane = np.arange(64)[None, :]
    a_idx = lane % 32 + 0 * i                       # index that is "lane % 32"
    b_idx = 8 * (i // 4) + 4 * (lane // 32) + i % 4   # index that comes from (i, lane / 32)
    g = np.zeros((M // 32, 32, 8, 32), bool)        # (wt, row, ob, feature)
    if l < 3:
        g[:, a_idx, :, b_idx] = np.moveaxis(bits, 1, -1)[:, i, lane].transpose(1, 2, 0, 3)
    else:
        g[:, b_idx, :, a_idx] = np.moveaxis(bits, 1, -1)[:, i, lane].transpose(1, 2, 0, 3)
    return g.reshape(M, 256)


def f16s_alg0_emulation(x, q, Ws, bs, n, njp, two_pass=False):
    """(two_pass: layers 1..3 multiply the hi + lo split of the weights instead of the tile's dithered image -- dither == 0.)
    float64 emulation of the FACTORED f16s chain's arithmetic (rn_g_chain_fwd_rr_f16s_alg0; model.py:108-152) on the padded pair
    space: layer 0 = fp16(x_j) @ (hi + lo of W0[:, :k])^T + [W0[:, k:2k] x_i + W0[:, 2k:] q + b0] (the bracket exact), layers 1..3 =
    the fp16-rounded previous activation @ the tile's dithered hi image; -> (list of pre-activations z_l (Mp, G), valid-row mask
    (Mp,), the EXACT fp32-model activations of the last layer (Mp, G))."""
    B, _, k = x.shape
    G, L = Ws[0].shape[0], len(Ws)
    f16r = lambda a: np.asarray(a, np.float32).astype(np.float16).astype(np.float32)
    Mp = B * n * njp
    b_of, i_of, j_of = np.divmod(np.arange(Mp) // njp, n)[0], (np.arange(Mp) // njp) % n, np.arange(Mp) % njp
    valid = j_of < n
    jj = np.where(valid, j_of, 0)
    xj = np.where(valid[:, None], x[b_of, jj], 0.0)
    W0 = Ws[0].astype(np.float64)
    wh = f16r(Ws[0][:, :k]); wl = f16r(Ws[0][:, :k] - wh)
    brack = x[b_of, i_of].astype(np.float64) @ W0[:, k:2 * k].T + q[b_of].astype(np.float64) @ W0[:, 2 * k:].T + bs[0]
    zs = [f16r(xj).astype(np.float64) @ (wh.astype(np.float64) + wl.astype(np.float64)).T + brack]
    exact = np.maximum(xj.astype(np.float64) @ W0[:, :k].T + brack, 0)
    timg = (np.arange(Mp) // 256) % F16S_V
    prev = f16r(np.maximum(zs[0], 0)).astype(np.float64)
    for l in range(1, L):
        z = np.empty((Mp, G))
        if two_pass:
            hi = f16r(Ws[l]); lo = f16r(Ws[l] - hi)
            z[:] = prev @ (hi.astype(np.float64) + lo.astype(np.float64)).T + bs[l]
        for d, im in enumerate(dither_images(Ws[l]) if not two_pass else []):
            z[timg == d] = prev[timg == d] @ im.astype(np.float64).T + bs[l]
        zs.append(z)
        exact = np.maximum(exact @ Ws[l].astype(np.float64).T + bs[l], 0)
        prev = f16r(np.maximum(z, 0)).astype(np.float64)
    return zs, valid, exact


def check_against_f16s_emulation(zs, valid, exact, Hs, masks, part, M_tiles_rows, n_parts_per_tile=1, grp_tol=3e-4):
    """The kernel's stored bf16 copies (row-major, already un-blocked), lane masks and per-tile pair-sum partials against the
    emulation: copies to 1 bf16 ulp, masks = the gates of the kernel's own pre-activations up to rounding-noise elements, pair
    sums to 1e-3, and to 3e-4 of the EXACT chain over groups of V tiles (one of each dithered image)."""
    L = len(zs)
    Mp, G = zs[0].shape
    for l in range(L):
        ref = np.maximum(zs[l], 0) * valid[:, None]
        if Hs is not None and l < 3:
            got = Hs[l].float().cpu().numpy() * valid[:, None]
            assert rel(got, bf16_round(ref)) <= BF16_ULP, l
        if masks is not None:
            bad = rr_mask_decode(masks[l], Mp, l) != ((zs[l] > 0) & valid[:, None])
            assert np.abs(zs[l][bad]).max(initial=0.0) <= 2e-3 * np.abs(zs[l]).max() and bad.mean() <= 2e-3, (l, bad.sum())
    last = np.maximum(zs[-1], 0) * valid[:, None]
    got = part.cpu().numpy().reshape(Mp // 256, n_parts_per_tile, G).sum(1)
    assert rel(got, last.reshape(Mp // 256, 256, G).sum(1)) <= 1e-3
    nt = (Mp // 256) // F16S_V * F16S_V
    e_grp = rel(got[:nt].reshape(nt // F16S_V, F16S_V, G).sum(1), (exact * valid[:, None])[:nt * 256].reshape(nt // F16S_V, 256 * F16S_V, G).sum(1))
    assert e_grp <= grp_tol, e_grp


@pytest.mark.parametrize("mode,B,n", [("train", 19, 64), ("infer", 2, 32)])
def test_g_chain_fwd_rr_f16s_alg0(H, mode, B, n):
    """The forward chain (f16s arithmetic on the factored first layer, model.py:108-152) against the float64 emulation of ITS
    arithmetic and against the exact chain: stored bf16 copies to 1 bf16 ulp of the largest value, masks = the gates of the kernel's
    own pre-activations on all but rounding-noise elements, pair sums to 1e-3 (3e-4 of the exact chain over groups of V tiles);
    then the e4m3 copies and the gate image against the 16-bit run."""
    L, G, k, Q = 4, 256, 26, 128
    M, kt, K0 = B * n * n, 2 * 26 + 128, 192
    x = formula.hash_uniform((B, n, k), 400, -1, 1).astype(np.float32)
    q = formula.hash_uniform((B, Q), 401, -1, 1).astype(np.float32)
    Ws = [formula.hash_uniform((G, kt if l == 0 else G), 410 + l, -0.15, 0.15).astype(np.float32) for l in range(L)]
    bs = [formula.hash_uniform((G,), 420 + l, -0.3, 0.3).astype(np.float32) for l in range(L)]
    wd = [dev(w) for w in Ws]
    w0T = torch.empty(kt, G, device="cuda")
    hiA, loA, jobsA = f16s_images(H, wd, kt, k)             # factored first layer: W0[:, 0:k]
    H.pack_matrix_frag_many(jobsA + [(wd[0], kt, 1, G, kt, w0T, 2)])
    Xp = torch.empty(B * n, 64, dtype=torch.float16, device="cuda"); Vc = torch.empty(B * n, G, device="cuda")
    H.pair_tables(dev(x), dev(q), w0T, dev(bs[0]), Xp, Vc, B, n, k, Q, G)
    train = mode == "train"
    def outs(rows):
        Hs = [torch.full((M, G), float("nan"), dtype=torch.bfloat16, device="cuda") for _ in range(3)] + [None] if train else None
        masks = list(torch.zeros(L, H.g_chain_rr_mask_bytes(M), dtype=torch.uint8, device="cuda")) if train else None
        return Hs, masks, torch.full((M // rows, G), float("nan"), dtype=torch.float32, device="cuda")
    HsA, mA, pA = outs(256)                               # (one partial row per tile)
    bd = [dev(b) for b in bs]
    H.g_chain_fwd_rr_f16s_alg0(Xp, Vc, n, hiA, loA, bd, HsA, mA, pA, M, G)
    torch.cuda.synchronize()
    HsA = unblock_h(HsA)
    zs, valid, exact = f16s_alg0_emulation(x, q, Ws, bs, n, n)
    check_against_f16s_emulation(zs, valid, exact, HsA, mA, pA, M)
    if train:
        # e4m3 copies: same arithmetic (masks, pair sums bitwise); the bytes are the e4m3 rounding of the fp16 operand, i.e. of
        # the value the bf16 copy rounds -- equal to the rounded bf16 copy except where the two 16-bit roundings straddle a tie
        Hs8 = [torch.full((M, G), 0x7f, dtype=torch.uint8, device="cuda").view(torch.float8_e4m3fn) for _ in range(3)] + [None]
        m8 = list(torch.zeros(L, H.g_chain_rr_mask_bytes(M), dtype=torch.uint8, device="cuda"))
        p8 = torch.full((M // 256, G), float("nan"), dtype=torch.float32, device="cuda")
        H.g_chain_fwd_rr_f16s_alg0(Xp, Vc, n, hiA, loA, bd, Hs8, m8, p8, M, G)
        torch.cuda.synchronize()
        Hs8 = unblock_h(Hs8)
        assert torch.equal(pA, p8)
        for l in range(L):
            assert torch.equal(mA[l], m8[l]), l
        for l in range(3):
            ref8 = HsA[l].float().to(torch.float8_e4m3fn)
            same = (Hs8[l].view(torch.uint8) == ref8.view(torch.uint8)).float().mean().item()
            assert same >= 0.97, (l, same)
            d = (Hs8[l].float() - HsA[l].float()).abs()
            assert bool((d <= HsA[l].float().abs() * 2.0 ** -4 + 2.0 ** -10).all()), l
        # ... and with the last layer's gate in the sign bits of the H_2 image (written one layer late, from the last layer's
        # epilogue): everything else bitwise as before, the image bitwise what rn_relu_gate_image merges into the plain H_2 image
        Hg = [torch.full((M, G), 0x7f, dtype=torch.uint8, device="cuda").view(torch.float8_e4m3fn) for _ in range(3)] + [None]
        mg = list(torch.zeros(L, H.g_chain_rr_mask_bytes(M), dtype=torch.uint8, device="cuda"))
        pg = torch.full((M // 256, G), float("nan"), dtype=torch.float32, device="cuda")
        H.g_chain_fwd_rr_f16s_alg0(Xp, Vc, n, hiA, loA, bd, Hg, mg, pg, M, G, gate=True)
        torch.cuda.synchronize()
        assert torch.equal(pg, p8)
        for l in range(L):
            assert torch.equal(mg[l], m8[l]), l
        for l in range(2):
            assert torch.equal(unblock(Hg[l]).view(torch.uint8), Hs8[l].view(torch.uint8)), l
        want = H.relu_gate_image(m8[3], H.rows_to_blocked(Hs8[2]), M)
        torch.cuda.synchronize()
        assert torch.equal(Hg[2].view(torch.uint8), want.view(torch.uint8))
        assert torch.equal(unblock(Hg[2]).view(torch.uint8) & 0x7f, Hs8[2].view(torch.uint8))
        assert torch.equal(unblock(Hg[2]).view(torch.uint8) >> 7, gate_image_ref(m8[3], M).to(torch.uint8))


@pytest.mark.parametrize("B,n", [(5, 64), (3, 32), (2, 196), (4, 40)])
def test_g_chain_fwd_rr_f16s_two_pass(H, B, n):
    """dither == 0: hi + lo split weights on EVERY layer (the inference arithmetic of eval()).  Pair sums against
    the float64 emulation of that arithmetic (2e-5: fp32 accumulation order) and against the exact chain (3e-4: the fp16 rounding of
    the activations is all that is left); and the point of the mode -- NOTHING depends on where a question sits in the batch: the
    same questions in reversed order give the same per-question sums, bitwise where a question is whole tiles (n % 32 == 0),
    to fp32 summation order on the padded j axis; a training-style call (H / masks) with dither == 0 is refused."""
    L, G, k, Q = 4, 256, 26, 128
    njp = (n + 31) // 32 * 32
    Mp, kt = B * n * njp, 2 * 26 + 128
    x = formula.hash_uniform((B, n, k), 400, -1, 1).astype(np.float32)
    q = formula.hash_uniform((B, Q), 401, -1, 1).astype(np.float32)
    Ws = [formula.hash_uniform((G, kt if l == 0 else G), 410 + l, -0.15, 0.15).astype(np.float32) for l in range(L)]
    bs = [formula.hash_uniform((G,), 420 + l, -0.3, 0.3).astype(np.float32) for l in range(L)]
    wd = [dev(w) for w in Ws]
    w0T = torch.empty(kt, G, device="cuda")
    his = [torch.empty(1, 65536, dtype=torch.float16, device="cuda") for _ in range(L)]
    los = [torch.empty(65536, dtype=torch.float16, device="cuda") for _ in range(L)]
    jobs = [(wd[0], kt, 1, G, k, his[0], 4 | 1), (wd[0], kt, 1, G, k, los[0], 8 | 1)]
    jobs += [(wd[l], G, 1, G, G, his[l], 4) for l in range(1, L)] + [(wd[l], G, 1, G, G, los[l], 8) for l in range(1, L)]
    H.pack_matrix_frag_many(jobs + [(wd[0], kt, 1, G, kt, w0T, 2)])
    bd = [dev(b) for b in bs]
    ppt = 2 if njp != n else 1

    def run(xx, qq):
        Xp = torch.zeros(B * n + 1, 64, dtype=torch.float16, device="cuda"); Vc = torch.empty(B * n, G, device="cuda")
        H.pair_tables(dev(xx), dev(qq), w0T, bd[0], Xp, Vc, B, n, k, Q, G)
        part = torch.full((Mp // 256 * ppt, G), float("nan"), device="cuda")
        H.g_chain_fwd_rr_f16s_alg0(Xp, Vc, n, his, los, bd, None, None, part, Mp, G, njp=njp)
        xg = torch.full((B, G), float("nan"), device="cuda")
        if ppt == 2:
            H.pair_sum_tiles(part, xg, Mp, n * njp, G)
        else:
            xg.copy_(part.view(B, (n * n) // 256, G).sum(1))
        torch.cuda.synchronize()
        return xg, part
    xg, part = run(x, q)
    zs, valid, exact = f16s_alg0_emulation(x, q, Ws, bs, n, njp, two_pass=True)
    want = (np.maximum(zs[-1], 0) * valid[:, None]).reshape(B, n * njp, G).sum(1)
    assert rel(xg.cpu().numpy(), want) <= 2e-5
    assert rel(xg.cpu().numpy(), (exact * valid[:, None]).reshape(B, n * njp, G).sum(1)) <= 3e-4
    xg_r, _ = run(x[::-1].copy(), q[::-1].copy())
    if njp == n and (n * n) % 256 == 0:
        assert torch.equal(xg_r.flip(0), xg)
    else:
        assert rel(xg_r.flip(0).cpu().numpy(), xg.cpu().numpy()) <= 2e-6
    with pytest.raises(RuntimeError, match="inference arithmetic"):
        Hs = [torch.empty(Mp, G, dtype=torch.bfloat16, device="cuda") for _ in range(3)] + [None]
        masks = list(torch.zeros(L, H.g_chain_rr_mask_bytes(Mp), dtype=torch.uint8, device="cuda"))
        H.g_chain_fwd_rr_f16s_alg0(torch.zeros(B * n + 1, 64, dtype=torch.float16, device="cuda"), torch.zeros(B * n, G, device="cuda"), n, his, los, bd,
                                   Hs, masks, part, Mp, G, njp=njp)


@pytest.mark.parametrize("B,n", [(16, 196), (4, 40)])
def test_g_chain_fwd_rr_f16s_alg0_padded(H, B, n):
    """Object counts that are no multiple of 32 (the 14 x 14 grid): the factored first layer on a PADDED j axis (njp = 32 ceil(n /
    32) pair rows per (question, i) group).  On the n valid rows of every group the stored copies, the masks and the pair sums must
    be those of the float64 emulation of the kernel's arithmetic on the padded pair space; the invalid rows must have all-zero
    masks in every layer; rn_pair_sum_tiles must add the two-rows-per-tile partials up per question (checked against the exact
    chain's per-question sums)."""
    L, G, k, Q = 4, 256, 26, 128
    njp = (n + 31) // 32 * 32
    M, Mp, kt, K0 = B * n * n, B * n * njp, 2 * 26 + 128, 192
    x = formula.hash_uniform((B, n, k), 400, -1, 1).astype(np.float32)
    q = formula.hash_uniform((B, Q), 401, -1, 1).astype(np.float32)
    Ws = [formula.hash_uniform((G, kt if l == 0 else G), 410 + l, -0.15, 0.15).astype(np.float32) for l in range(L)]
    bs = [formula.hash_uniform((G,), 420 + l, -0.3, 0.3).astype(np.float32) for l in range(L)]
    wd = [dev(w) for w in Ws]
    w0T = torch.empty(kt, G, device="cuda")
    hiA, loA, jobsA = f16s_images(H, wd, kt, k)
    H.pack_matrix_frag_many(jobsA + [(wd[0], kt, 1, G, kt, w0T, 2)])
    Xp = torch.zeros(B * n + 1, 64, dtype=torch.float16, device="cuda"); Vc = torch.empty(B * n, G, device="cuda")
    H.pair_tables(dev(x), dev(q), w0T, dev(bs[0]), Xp, Vc, B, n, k, Q, G)
    assert not Xp[B * n].any()
    bd = [dev(b) for b in bs]
    HsA = [torch.full((Mp, G), float("nan"), dtype=torch.bfloat16, device="cuda") for _ in range(3)] + [None]
    mA = list(torch.full((L, H.g_chain_rr_mask_bytes(Mp)), 0xff, dtype=torch.uint8, device="cuda"))
    pA = torch.full((Mp // 256 * 2, G), float("nan"), device="cuda")
    H.g_chain_fwd_rr_f16s_alg0(Xp, Vc, n, hiA, loA, bd, HsA, mA, pA, Mp, G, njp=njp)
    xgA = torch.full((B, G), float("nan"), device="cuda")
    H.pair_sum_tiles(pA, xgA, Mp, n * njp, G)
    torch.cuda.synchronize()
    HsA = unblock_h(HsA)
    # the direct oracle leg on the PADDED pair space (njp = 224 for the 14 x 14 grid): float64 emulation of the kernel's arithmetic
    zs, vrow, exact = f16s_alg0_emulation(x, q, Ws, bs, n, njp)
    # (groups of V tiles hold different numbers of valid rows here: the image offsets cancel less evenly than on whole tiles)
    check_against_f16s_emulation(zs, vrow, exact, HsA, mA, pA, Mp, n_parts_per_tile=2, grp_tol=1e-3)
    valid = lambda t: t.view(B * n, njp, -1)[:, :n].reshape(M, -1)
    # inference variant (no stores): the same pair sums, bitwise
    pI = torch.full_like(pA, float("nan"))
    H.g_chain_fwd_rr_f16s_alg0(Xp, Vc, n, hiA, loA, bd, None, None, pI, Mp, G, njp=njp)
    torch.cuda.synchronize()
    assert torch.equal(pI, pA)
    xg_exact = (exact * vrow[:, None]).reshape(B, n * njp, G).sum(1)            # per question, over the valid rows: the whole pair sum
    assert rel(xgA.cpu().numpy(), xg_exact) <= 1e-3
    for l in range(L):
        gA = torch.from_numpy(rr_mask_decode(mA[l], Mp, l))
        inval = gA.view(B * n, njp, G)[:, n:]
        assert not inval.any(), l                              # padded rows: gates cleared in every layer
    # e4m3 copies with the last layer's gate in the sign bits of H_2: bitwise rn_relu_gate_image of the layer's masks merged into
    # the plain e4m3 H_2 image (no gate on the padded rows), the masks and pair sums those of the run above
    H8 = [torch.full((Mp, G), 0x7f, dtype=torch.uint8, device="cuda").view(torch.float8_e4m3fn) for _ in range(3)] + [None]
    m8 = list(torch.zeros(L, H.g_chain_rr_mask_bytes(Mp), dtype=torch.uint8, device="cuda"))
    p8 = torch.full_like(pA, float("nan"))
    H.g_chain_fwd_rr_f16s_alg0(Xp, Vc, n, hiA, loA, bd, H8, m8, p8, Mp, G, njp=njp)
    Hg = [torch.full((Mp, G), 0x7f, dtype=torch.uint8, device="cuda").view(torch.float8_e4m3fn) for _ in range(3)] + [None]
    mg = list(torch.zeros(L, H.g_chain_rr_mask_bytes(Mp), dtype=torch.uint8, device="cuda"))
    pg = torch.full_like(pA, float("nan"))
    H.g_chain_fwd_rr_f16s_alg0(Xp, Vc, n, hiA, loA, bd, Hg, mg, pg, Mp, G, njp=njp, gate=True)
    torch.cuda.synchronize()
    assert torch.equal(p8, pA) and torch.equal(pg, pA)
    for l in range(L):
        assert torch.equal(m8[l], mA[l]) and torch.equal(mg[l], mA[l]), l
    for l in range(2):
        assert torch.equal(Hg[l].view(torch.uint8), H8[l].view(torch.uint8)), l
    want = H.relu_gate_image(m8[3], H8[2].clone(), Mp)
    torch.cuda.synchronize()
    assert torch.equal(Hg[2].view(torch.uint8), want.view(torch.uint8))
    assert not (unblock(Hg[2]).view(torch.uint8) >> 7).view(B * n, njp, G)[:, n:].any()


def gate_image_ref(mask, M):
    """e4m3 {0, 1} gate image of the layer-3 masks, from the test's own mask decoder: row-major (M, 256) float 0 / 1."""
    return torch.from_numpy(rr_mask_decode(mask, M, 3)).cuda()


def test_relu_gate_image(H):
    M = 32 * 77
    g = torch.Generator(device="cuda").manual_seed(3)
    mask = torch.randint(0, 256, (H.g_chain_rr_mask_bytes(M),), dtype=torch.uint8, device="cuda", generator=g)
    base = torch.randint(0, 256, (M, 256), dtype=torch.uint8, device="cuda", generator=g)       # (stale sign bits are replaced)
    img = H.relu_gate_image(mask, H.rows_to_blocked(base.view(torch.float8_e4m3fn)), M)
    torch.cuda.synchronize()
    got = unblock(img).view(torch.uint8)
    assert torch.equal(got, (base & 0x7f) | (gate_image_ref(mask, M).to(torch.uint8) << 7))


@pytest.mark.parametrize("B,n", [(17, 64), (3, 64), (2, 32)])
def test_wgrad_gated_and_bwd_skip0(H, B, n):
    """The last g layer without its gradient matrix: rn_g_chain_bwd_rr with dZ[0] = NULL must give the same dZ[1..3] as
    the full call; the stored dZ[0] image (row-blocked) = bf16(dxg) where the gate is set; and the GATE job of
    rn_g_wgrad_blocked (the gate in the sign bits of the e4m3 activation image, both on the fp8 pipe, scaled by the un-rounded dxg per question) must give the
    float64 product gate * dxg -- to fp32 accumulation accuracy, i.e. closer to the reference than the job on the stored,
    bf16-rounded dZ[0], which is checked against ITS float64 product.  B = 17: 272 tiles > 256 CUs; the stored job runs as 160 wide
    units (6.8 64-row steps each), the gate job as 40 splits x 4 workgroups straddling questions; B = 3: 40 splits; (2, 32): 32
    steps in all -- fewer than the ring is deep, 16 steps per question."""
    L, G = 4, 256
    M = B * n * n
    g = torch.Generator(device="cuda").manual_seed(5)
    masks = list(torch.randint(0, 256, (L, H.g_chain_rr_mask_bytes(M)), dtype=torch.uint8, device="cuda", generator=g))
    dxg = (torch.rand(B, G, device="cuda", generator=g) - 0.5)
    Wt = list(torch.empty(L - 1, 65536, dtype=torch.bfloat16, device="cuda"))
    for st in range(L - 1):
        W = dev(bf16_round(formula.hash_uniform((G, G), 330 + st, -0.15, 0.15)))
        pack_frag(H, W, 1, G, G, G, Wt[st], st == 0)
    full = list(torch.zeros(L, M, G, dtype=torch.bfloat16, device="cuda"))
    H.g_chain_bwd_rr(dxg, masks, Wt, full, M, n * n, G)
    part = [None] + list(torch.zeros(L - 1, M, G, dtype=torch.bfloat16, device="cuda"))
    H.g_chain_bwd_rr(dxg, masks, Wt, part, M, n * n, G)
    torch.cuda.synchronize()
    for s in range(1, L):
        assert torch.equal(full[s], part[s]), s
    gate = gate_image_ref(masks[L - 1], M)
    assert torch.equal(unblock(full[0]).float(), dxg.bfloat16().float().repeat_interleave(n * n, 0) * gate)
    A8 = (torch.rand(M, G, device="cuda", generator=g) * 4).to(torch.float8_e4m3fn)
    Ab = H.rows_to_blocked(A8)
    dW0 = torch.full((G, G), float("nan"), device="cuda"); db0 = torch.full((G,), float("nan"), device="cuda")
    H.g_wgrad_blocked([(full[0], Ab, dW0, db0)], M, rows_per_question=n * n)
    dW1 = torch.full((G, G), float("nan"), device="cuda"); db1 = torch.full((G,), float("nan"), device="cuda")
    H.g_wgrad_blocked([(None, H.relu_gate_image(masks[L - 1], Ab.clone(), M), dW1, db1)], M, dxg=dxg, rows_per_question=n * n)
    torch.cuda.synchronize()
    dz_stored = unblock(full[0]).double()
    dz_exact = dxg.double().repeat_interleave(n * n, 0) * gate
    for dW, db, dz in ((dW0, db0, dz_stored), (dW1, db1, dz_exact)):
        assert rel(dW.cpu().numpy(), (dz.t() @ A8.double()).cpu().numpy()) <= 2e-5
        assert rel(db.cpu().numpy(), dz.sum(0).cpu().numpy()) <= 2e-5


def _padded_forward_masks(H, B, n, njp):
    """Lane masks of a real forward call on the PADDED j axis (their bits are cleared for the rows j >= n in every layer)."""
    L, G, k, Q = 4, 256, 26, 128
    Mp, kt = B * n * njp, 2 * 26 + 128
    x = formula.hash_uniform((B, n, k), 400, -1, 1).astype(np.float32)
    q = formula.hash_uniform((B, Q), 401, -1, 1).astype(np.float32)
    Ws = [formula.hash_uniform((G, kt if l == 0 else G), 410 + l, -0.15, 0.15).astype(np.float32) for l in range(L)]
    bs = [formula.hash_uniform((G,), 420 + l, -0.3, 0.3).astype(np.float32) for l in range(L)]
    wd = [dev(w) for w in Ws]
    w0T = torch.empty(kt, G, device="cuda")
    hiA, loA, jobsA = f16s_images(H, wd, kt, k)
    H.pack_matrix_frag_many(jobsA + [(wd[0], kt, 1, G, kt, w0T, 2)])
    Xp = torch.zeros(B * n + 1, 64, dtype=torch.float16, device="cuda"); Vc = torch.empty(B * n, G, device="cuda")
    H.pair_tables(dev(x), dev(q), w0T, dev(bs[0]), Xp, Vc, B, n, k, Q, G)
    Hs = [torch.empty(Mp, G, dtype=torch.bfloat16, device="cuda") for _ in range(3)] + [None]
    masks = list(torch.zeros(L, H.g_chain_rr_mask_bytes(Mp), dtype=torch.uint8, device="cuda"))
    part = torch.empty(Mp // 256 * 2, G, device="cuda")
    H.g_chain_fwd_rr_f16s_alg0(Xp, Vc, n, hiA, loA, [dev(b) for b in bs], Hs, masks, part, Mp, G, njp=njp)
    torch.cuda.synchronize()
    return masks


@pytest.mark.parametrize("B,n,tpu,whole", [(17, 64, 0, "lib"), (64, 64, 0, "lib"), (3, 64, 1, "lib"), (3, 64, 8, "all"), (3, 64, 8, 0), (3, 64, 4, 5), (2, 32, 2, "lib"),
                                          (2, 96, 0, "lib"), (2, 196, 0, "lib"), (32, 196, 0, "lib"), (32, 196, 0, "all"), (2, 196, 5, 33), (4, 40, 0, "lib"), (4, 40, 1, "lib")])
def test_g_chain_bwd_rr_red(H, B, n, tpu, whole):
    """rn_g_chain_bwd_rr_red: the pair-axis reductions of layer 0's gradient formed inside the backward chain.  The stored images
    dZ[1], dZ[2] must be bitwise those of rn_g_chain_bwd_rr; Rj / Ri / Rq (rn_pair_reduce_parts) must equal, to fp32 accumulation
    accuracy, the float64 reductions of (dZ[2] @ W_1) * gate_0 computed from the kernel's OWN stored dZ[2] and the forward's
    layer-0 lane masks (model.py:117-127 backward; oracle.rl_backward_np's Rj / Ri).  tpu = 0: the library's choice of tiles per
    unit; (17, 64): more units than CUs; (64, 64): the headline shape; (2, 96): three j blocks per (question, i); (., 196), (4, 40):
    the PADDED j axis (njp = 224 / 64 pair rows per (question, i), masks of a real padded forward call) with n % 8 != 0 -- the last
    tile of a (question, j block) has four valid i, its spare waves repeat the last one and must add nothing; (32, 196) is
    BASELINE.json configs[4] at its real size.  whole: the split point of the BALANCED schedule -- units below it run as a whole,
    the tiles of the others one by one with a record each ("lib": rn_g_chain_bwd_rr_red_whole's choice -- 1024 of the 1120 units of
    (32, 196) on 256 CUs --, "all": no tail, 0: every tile on its own, 5 / 33: a split inside a (question, j block))."""
    L, G = 4, 256
    njp = (n + 31) // 32 * 32
    M = B * n * njp
    g = torch.Generator(device="cuda").manual_seed(7)
    if njp == n:
        masks = list(torch.randint(0, 256, (L, H.g_chain_rr_mask_bytes(M)), dtype=torch.uint8, device="cuda", generator=g))
    else:
        masks = _padded_forward_masks(H, B, n, njp)
    dxg = (torch.rand(B, G, device="cuda", generator=g) - 0.5)
    Wt = list(torch.empty(L - 1, 65536, dtype=torch.bfloat16, device="cuda"))
    Ws = []
    for st in range(L - 1):
        Ws.append(bf16_round(formula.hash_uniform((G, G), 340 + st, -0.15, 0.15)))
        pack_frag(H, dev(Ws[-1]), 1, G, G, G, Wt[st], st == 0)
    old = [None] + list(torch.zeros(L - 1, M, G, dtype=torch.bfloat16, device="cuda"))
    H.g_chain_bwd_rr(dxg, masks, Wt, old, M, n * njp, G)
    tpu = tpu or H.g_chain_bwd_rr_red_tpu(M, n, njp)
    tpbj = (n + 7) // 8
    assert tpu > 0 and tpbj % tpu == 0
    nu = tpbj // tpu
    nunits = H.g_chain_bwd_rr_red_units(M, n, njp, tpu)
    whole = {"lib": H.g_chain_bwd_rr_red_whole(M, n, njp, tpu), "all": nunits}.get(whole, whole)
    assert 0 <= whole <= nunits
    if (B, n) == (32, 196) and whole != nunits:
        assert tpu == 5 and whole == (nunits // 256) * 256 < nunits            # (the stress shape HAS a tail on a 256-CU chip)
    new = [None] + list(torch.zeros(L - 2, M, G, dtype=torch.bfloat16, device="cuda")) + [None]
    rj_part = torch.full((H.g_chain_bwd_rr_red_records(M, n, njp, tpu, whole), 32, G), float("nan"), device="cuda")
    ri_part = torch.full((M // 16, G), float("nan"), device="cuda")
    H.g_chain_bwd_rr_red(dxg, masks, Wt, new, M, n, G, rj_part, ri_part, tpu, njp=njp, whole=whole)
    Rj = torch.full((B * n, G), float("nan"), device="cuda"); Ri = torch.full((B * n, G), float("nan"), device="cuda")
    Rq = torch.full((B, G), float("nan"), device="cuda")
    H.pair_reduce_parts(rj_part, ri_part, Rj, Ri, Rq, B, n, G, nu, njp=njp, tpu=tpu, whole=whole)
    # a second run must give the same bits (fixed summation orders)
    rj2, ri2 = torch.empty_like(rj_part), torch.empty_like(ri_part)
    H.g_chain_bwd_rr_red(dxg, masks, Wt, new, M, n, G, rj2, ri2, tpu, njp=njp, whole=whole)
    torch.cuda.synchronize()
    assert torch.equal(rj2, rj_part) and torch.equal(ri2, ri_part)
    assert not torch.isnan(rj_part).any() and not torch.isnan(ri_part).any()
    for s_ in (1, 2):
        assert torch.equal(old[s_], new[s_]), s_
    gate0 = torch.from_numpy(rr_mask_decode(masks[0], M, 0)).cuda()
    dz2 = unblock(new[2]).double()
    dz0 = ((dz2 @ dev(Ws[2]).double()) * gate0).view(B, n, njp, G)                 # [b, i, j, :]
    assert not dz0[:, :, n:].any()                                                 # (padded rows: exact zeros)
    scale = dz0.abs().max().item()
    assert (Rj.double().view(B, n, G) - dz0.sum(1)[:, :n]).abs().max().item() <= 2e-4 * scale * max(1.0, n / 64)   # (n terms of ~1e-6 relative error each)
    assert (Ri.double().view(B, n, G) - dz0.sum(2)).abs().max().item() <= 2e-4 * scale * max(1.0, n / 64)
    assert (Rq.double() - dz0.sum((1, 2))).abs().max().item() <= 2e-3 * scale * max(1.0, n / 64)
    # ... and the stored-dZ_0 path (bf16 rows + rn_pair_reduce_bwd) agrees within ITS rounding (2^-9 per term)
    Rj_o = torch.empty_like(Rj); Ri_o = torch.empty_like(Ri); Rq_o = torch.empty_like(Rq)
    H.pair_reduce_bwd(old[3], G, Rj_o, Ri_o, Rq_o, 0, B, n, G, njp=njp)
    torch.cuda.synchronize()
    assert (Rj_o - Rj).abs().max().item() <= n * 2.0 ** -9 * scale and (Ri_o - Ri).abs().max().item() <= n * 2.0 ** -9 * scale


@pytest.mark.parametrize("a8", [False, True])
@pytest.mark.parametrize("B,n", [(16, 64), (5, 64), (1, 32)])
def test_wgrad_blocked_three_jobs(H, B, n, a8):
    """The step's three weight gradients as ONE rn_g_wgrad_blocked launch (e4m3 activations: two stored gradients + the gate job;
    16-bit activations: three stored gradients): every dW / db against a float64 product, bitwise what three single-job launches
    give (same splits, same order), bitwise repeatable; the db partials add up to per-question column sums of dZ when the splits
    are question-aligned."""
    G = 256
    M = B * n * n
    g = torch.Generator(device="cuda").manual_seed(17)
    dZ = [((torch.rand(M, G, device="cuda", generator=g) - 0.5) * 1e-2).bfloat16() for _ in range(3)]
    Hc = [(torch.rand(M, G, device="cuda", generator=g) * 3).clamp_min(0.0) for _ in range(3)]
    Hc = [torch.where(torch.rand(M, G, device="cuda", generator=g) < 0.4, torch.zeros_like(h), h) for h in Hc]
    Hc = [h.to(torch.float8_e4m3fn) if a8 else h.bfloat16() for h in Hc]
    mask = torch.randint(0, 256, (H.g_chain_rr_mask_bytes(M),), dtype=torch.uint8, device="cuda", generator=g)
    dxg = (torch.rand(B, G, device="cuda", generator=g) - 0.5)
    dZb = [H.rows_to_blocked(z) for z in dZ]
    Hb = [H.rows_to_blocked(h) for h in Hc]
    if a8:
        dZb[2] = None                                          # the gate job: gate = the sign bits of its activation image
        H.relu_gate_image(mask, Hb[2], M)
    def run(sel, aligned=True):
        outs = [(torch.full((G, G), float("nan"), device="cuda"), torch.full((G,), float("nan"), device="cuda")) for _ in sel]
        ws, parts = H.g_wgrad_blocked([(dZb[j], Hb[j], o[0], o[1]) for j, o in zip(sel, outs)], M, dxg=dxg, rows_per_question=n * n, aligned=aligned)
        torch.cuda.synchronize()
        return outs, [p.clone() for p in parts]
    all3, parts3 = run([0, 1, 2])
    again, _ = run([0, 1, 2])
    dz3 = dxg.double().repeat_interleave(n * n, 0) * gate_image_ref(mask, M)
    Z = H.wgrad_blocked_splits(M, n * n, 3, aligned=True)
    # not question-aligned -- the product launch: on e4m3 activations the stored gradients run as WIDE units (one workgroup per row
    # split holds the whole 256 x 256 dW, db from in-lane dot products) beside the gate job's quad units; same sums, other order
    free3, _ = run([0, 1, 2], aligned=False)
    free3b, _ = run([0, 1, 2], aligned=False)
    for j in range(3):
        assert torch.equal(all3[j][0], again[j][0]) and torch.equal(all3[j][1], again[j][1]), j
        assert torch.equal(free3[j][0], free3b[j][0]) and torch.equal(free3[j][1], free3b[j][1]), j          # bitwise repeatable
        assert rel(free3[j][0].cpu().numpy(), all3[j][0].cpu().numpy()) <= 2e-5 and rel(free3[j][1].cpu().numpy(), all3[j][1].cpu().numpy()) <= 2e-5, j
        dzj = (dxg.double().repeat_interleave(n * n, 0) * gate_image_ref(mask, M)) if (a8 and j == 2) else dZ[j].double()
        assert rel(free3[j][0].cpu().numpy(), (dzj.t() @ Hc[j].double()).cpu().numpy()) <= 2e-5, j
        assert rel(free3[j][1].cpu().numpy(), dzj.sum(0).cpu().numpy()) <= 2e-5, j
        if H.wgrad_blocked_splits(M, n * n, 1, aligned=True) == Z:       # (same splits alone and in the launch of three: bitwise)
            single, _ = run([j])
            assert torch.equal(all3[j][0], single[0][0]) and torch.equal(all3[j][1], single[0][1]), j
        dz = dz3 if (a8 and j == 2) else dZ[j].double()
        eW = rel(all3[j][0].cpu().numpy(), (dz.t() @ Hc[j].double()).cpu().numpy())
        eb = rel(all3[j][1].cpu().numpy(), dz.sum(0).cpu().numpy())
        assert eW <= 2e-5 and eb <= 2e-5, (j, eW, eb)
    if Z % B == 0 and (M // 64) % Z == 0:
        rq = parts3[0].view(B, (Z // B) * 4, G).sum(1)
        assert rel(rq.cpu().numpy(), dZ[0].double().view(B, n * n, G).sum(1).cpu().numpy()) <= 2e-5
    # the stand-alone per-question sums of a blocked image
    Rq = torch.full((B, G), float("nan"), device="cuda")
    H.blocked_question_sums(dZb[1], Rq, M, n * n)
    torch.cuda.synchronize()
    assert rel(Rq.cpu().numpy(), dZ[1].double().view(B, n * n, G).sum(1).cpu().numpy()) <= 2e-5


def test_wgrad_fp8_operand(H):
    """The activation operand as an e4m3 image (a_dtype = RN_FP8): every e4m3 value is a bf16 value, so both images must give the
    float64 product to fp32-accumulation accuracy -- the e4m3 one as WIDE units (round 6: one workgroup per row split, other
    split count and add order than the bf16 image's quad units) -- and, on the SAME mapping (quad units, an `aligned` launch),
    bitwise the same sums: both images put a pair row into the same MFMA k slot."""
    B, n, G = 17, 64, 256
    M = B * n * n
    g = torch.Generator(device="cuda").manual_seed(11)
    A8 = (torch.rand(M, G, device="cuda", generator=g) * 6).clamp_min(0.0)
    A8 = torch.where(torch.rand(M, G, device="cuda", generator=g) < 0.5, torch.zeros_like(A8), A8).to(torch.float8_e4m3fn)
    A16 = A8.float().bfloat16()
    assert torch.equal(A16.float(), A8.float())
    dZ = ((torch.rand(M, G, device="cuda", generator=g) - 0.5) * 1e-2).bfloat16()
    dZb = H.rows_to_blocked(dZ)
    out = []
    for A in (A16, A8):
        dW = torch.full((G, G), float("nan"), device="cuda"); db = torch.full((G,), float("nan"), device="cuda")
        H.g_wgrad_blocked([(dZb, H.rows_to_blocked(A), dW, db)], M, rows_per_question=n * n)
        out.append((dW, db))
    torch.cuda.synchronize()
    ref = dZ.double().t() @ A8.double()
    refb = dZ.double().sum(0)
    for dW, db in out:
        assert rel(dW.cpu().numpy(), ref.cpu().numpy()) <= 1e-5 and rel(db.cpu().numpy(), refb.cpu().numpy()) <= 1e-5
    # the same mapping (quad units; what an `aligned` launch runs on either image) is bitwise the same on both images
    outq = []
    for A in (A16, A8):
        dW = torch.full((G, G), float("nan"), device="cuda"); db = torch.full((G,), float("nan"), device="cuda")
        H.g_wgrad_blocked([(dZb, H.rows_to_blocked(A), dW, db)], M, rows_per_question=n * n, aligned=True)
        outq.append((dW, db))
    torch.cuda.synchronize()
    assert torch.equal(outq[0][0], outq[1][0]) and torch.equal(outq[0][1], outq[1][1])


@pytest.mark.parametrize("B,npairs", [(2, 512), (24, 32 * 100), (16, 144)])
def test_g_chain_bwd_rr(H, B, npairs):
    """Register-resident backward chain on arbitrary (random) lane masks: dZ[0] = bf16(dxg) where gate_3; every further dZ equals
    one un-fused dgrad step on the kernel's OWN previous dZ (<= 1 bf16 ulp), zero where gated.  (16, 144): a wave's 32 rows
    straddle two questions."""
    G, L = 256, 4
    M = B * npairs
    g = torch.Generator(device="cuda").manual_seed(11)
    masks = list(torch.randint(0, 256, (L, H.g_chain_rr_mask_bytes(M)), dtype=torch.uint8, device="cuda", generator=g))       # equally spaced
    Ws = [bf16_round(formula.hash_uniform((G, G), 310 + l, -0.15, 0.15)) for l in range(L)]
    dxg = formula.hash_uniform((B, G), 410, -1, 1)
    Wtf = list(torch.empty(L - 1, 65536, dtype=torch.bfloat16, device="cuda"))
    for s, f in enumerate(Wtf):
        pack_frag(H, dev(Ws[L - 1 - s]), 1, G, G, G, f, s == 0)       # element (in, out) = W[out][in]
    dZs = list(torch.full((L, M, G), float("nan"), dtype=torch.bfloat16, device="cuda"))
    H.g_chain_bwd_rr(dev(dxg), masks, Wtf, dZs, M, npairs, G)
    torch.cuda.synchronize()
    dZs = unblock_h(dZs)                                       # layers 3..1: row-blocked images; layer 0: row-major
    gates = [rr_mask_decode(masks[l], M, l) for l in range(L)]
    got = dZs[0].float().cpu().numpy()
    assert np.array_equal(got, bf16_round(np.repeat(dxg, npairs, axis=0)) * gates[L - 1])
    for s in range(L - 1):
        ref = (got.astype(np.float64) @ Ws[L - 1 - s].astype(np.float64)) * gates[L - 2 - s]
        got = dZs[s + 1].float().cpu().numpy()
        err = np.abs(got - ref) / np.maximum(np.abs(ref), np.abs(ref).max() * 1e-2)
        assert err.max() <= BF16_ULP, (s, err.max())
        assert np.all(got[~gates[L - 2 - s]] == 0)


# ----------------------------------------------------------------------------- K3
@pytest.mark.parametrize("code", [0, 1])
@pytest.mark.parametrize("B,npairs,G", [(4, 4096, 256), (3, 144, 512), (2, 38416, 256)])
def test_pair_sum_fwd_bwd(H, code, B, npairs, G):
    Hl = np.maximum(formula.hash_uniform((B * npairs, G), 50, -1, 1), 0)
    if code == 0:
        Hl = bf16_round(Hl)
    xg = torch.empty(B, G, dtype=torch.float32, device="cuda")
    Hd = dev(Hl).to(tdt(code))
    H.pair_sum_fwd(Hd, G, xg, code, B, npairs, G)
    ref = Hl.reshape(B, npairs, G).sum(1, dtype=np.float64)
    assert rel(xg.cpu().numpy(), ref) <= F32_TOL
    dxg = formula.hash_uniform((B, G), 51)
    dZ = torch.empty(B * npairs, G, dtype=tdt(code), device="cuda")
    H.pair_sum_bwd(dev(dxg), Hd, G, dZ, G, code, B, npairs, G)
    refz = np.repeat(dxg, npairs, axis=0) * (Hl > 0)
    if code == 0:
        refz = bf16_round(refz)
    assert np.array_equal(dZ.float().cpu().numpy(), refz)


# ----------------------------------------------------------------------------- wgrad
@pytest.mark.parametrize("code", [0, 1])
@pytest.mark.parametrize("M,N,K,Ktrue", [(4096, 256, 192, 180), (576, 512, 320, 270), (5000, 256, 384, 384), (2048, 256, 64, 52),
                                         (64 * 700, 256, 256, 256), (64 * 333, 256, 192, 180)])
def test_g_linear_bwd_wgrad(H, code, M, N, K, Ktrue):
    """The general weight-gradient kernel on row-major operands (rn_wgrad.hip): bf16 (LDS transpose reads) and fp32."""
    dZ = formula.hash_uniform((M, N), 60, -1, 1)
    A = np.zeros((M, K), np.float32); A[:, :Ktrue] = formula.hash_uniform((M, Ktrue), 61, -1, 1)
    if code == 0:
        dZ, A = bf16_round(dZ), bf16_round(A)
    dW = torch.full((N, Ktrue), 9.0, dtype=torch.float32, device="cuda")
    db = torch.full((N,), 9.0, dtype=torch.float32, device="cuda")
    H.g_linear_bwd_wgrad(dev(dZ).to(tdt(code)), N, dev(A).to(tdt(code)), K, dW, db, code, M, N, K, Ktrue)
    torch.cuda.synchronize()
    refW = dZ.astype(np.float64).T @ A[:, :Ktrue].astype(np.float64)
    refb = dZ.sum(0, dtype=np.float64)
    eW, eb = rel(dW.cpu().numpy(), refW), rel(db.cpu().numpy(), refb)
    assert eW <= F32_TOL * 5 and eb <= F32_TOL * 5, (eW, eb)
    # determinism: a second call gives bit-identical results
    dW2 = torch.empty_like(dW); db2 = torch.empty_like(db)
    H.g_linear_bwd_wgrad(dev(dZ).to(tdt(code)), N, dev(A).to(tdt(code)), K, dW2, db2, code, M, N, K, Ktrue)
    assert torch.equal(dW, dW2) and torch.equal(db, db2)


@pytest.mark.parametrize("M,N,K,Ktrue", [(576, 512, 512, 512), (9216, 512, 32, 14), (5000, 256, 192, 180)])
def test_g_linear_bwd_wgrad_split_bf16_arithmetic(H, M, N, K, Ktrue):
    """RN_F32X3 weight gradient: the fp32 operands split into hi + lo bf16 tiles in LDS, three bf16 MFMAs per 16 rows -- dW and db
    against fp64 (2^-16 of a product dropped: 3e-5 of the largest entry with room), and bitwise reproducible."""
    dZ = formula.hash_uniform((M, N), 70, -1, 1)
    A = np.zeros((M, K), np.float32); A[:, :Ktrue] = np.maximum(formula.hash_uniform((M, Ktrue), 71, -1, 1), 0)
    dW = torch.full((N, Ktrue), float("nan"), device="cuda"); db = torch.full((N,), float("nan"), device="cuda")
    H.g_linear_bwd_wgrad(dev(dZ), N, dev(A), K, dW, db, H.RN_F32X3, M, N, K, Ktrue)
    torch.cuda.synchronize()
    refW = dZ.astype(np.float64).T @ A[:, :Ktrue].astype(np.float64)
    refb = dZ.sum(0, dtype=np.float64)
    eW = np.abs(dW.cpu().numpy() - refW).max() / np.abs(refW).max()
    eb = np.abs(db.cpu().numpy() - refb).max() / np.abs(refb).max()
    assert eW <= 3e-5 and eb <= 3e-5, (eW, eb)
    dW2 = torch.empty_like(dW); db2 = torch.empty_like(db)
    H.g_linear_bwd_wgrad(dev(dZ), N, dev(A), K, dW2, db2, H.RN_F32X3, M, N, K, Ktrue)
    assert torch.equal(dW, dW2) and torch.equal(db, db2)


@pytest.mark.parametrize("B,n,k,Q,N", [(64, 64, 26, 128, 256), (3, 12, 7, 256, 512), (5, 9, 32, 40, 100)])
def test_pair_dx_dq(H, B, n, k, Q, N):
    """dx = Rj W0[:, :k] + Ri W0[:, k:2k], dq = Rq W0[:, 2k:] in one launch, against float64."""
    Rj = formula.hash_uniform((B * n, N), 90, -1, 1); Ri = formula.hash_uniform((B * n, N), 91, -1, 1)
    Rq = formula.hash_uniform((B, N), 92, -1, 1); W0 = formula.hash_uniform((N, 2 * k + Q), 93, -0.2, 0.2)
    dx = torch.full((B, n, k), 9.0, device="cuda"); dq = torch.full((B, Q), 9.0, device="cuda")
    H.pair_dx_dq(dev(Rj), dev(Ri), dev(Rq), dev(W0), dx, dq, B, n, k, Q, N)
    torch.cuda.synchronize()
    W = W0.astype(np.float64)
    assert rel(dx.cpu().numpy().reshape(B * n, k), Rj @ W[:, :k] + Ri @ W[:, k:2 * k]) <= F32_TOL
    assert rel(dq.cpu().numpy(), Rq @ W[:, 2 * k:]) <= F32_TOL


@pytest.mark.parametrize("B,n,k,Q", [(64, 64, 26, 128), (3, 12, 7, 256), (2, 196, 26, 128)])
def test_wgrad0_from_reductions(H, B, n, k, Q):
    """dW_0 = [Rj^T X | Ri^T X | Rq^T q], db_0 = sum_b Rq: must equal dZ_0^T P / column sums of dZ_0 computed directly
    from a pair matrix (fp64), for a strided x view as RN.forward produces it."""
    N = 256
    xs = formula.hash_uniform((B, k, n), 80, -1, 1)                      # (B, k, n) storage, (B, n, k) view
    q = formula.hash_uniform((B, Q), 81, -1, 1)
    dZ = formula.hash_uniform((B * n * n, N), 82, -1, 1).astype(np.float64)
    x = np.transpose(xs, (0, 2, 1))
    d4 = dZ.reshape(B, n, n, N)
    Rj, Ri, Rq = d4.sum(1).reshape(B * n, N), d4.sum(2).reshape(B * n, N), d4.sum((1, 2))
    P = np.concatenate([np.broadcast_to(x[:, None, :, :], (B, n, n, k)), np.broadcast_to(x[:, :, None, :], (B, n, n, k)),
                        np.broadcast_to(q[:, None, None, :], (B, n, n, Q))], -1).reshape(B * n * n, 2 * k + Q)
    ref = dZ.T @ P
    dW = torch.full((N, 2 * k + Q), 9.0, device="cuda"); db = torch.full((N,), 9.0, device="cuda")
    H.wgrad0_from_reductions(dev(Rj.astype(np.float32)), dev(Ri.astype(np.float32)), dev(Rq.astype(np.float32)),
                             dev(xs).permute(0, 2, 1), dev(q), dW, db)
    torch.cuda.synchronize()
    assert rel(dW.cpu().numpy(), ref) <= F32_TOL and rel(db.cpu().numpy(), dZ.sum(0)) <= F32_TOL


# ----------------------------------------------------------------------------- pair reduce
@pytest.mark.parametrize("code", [0, 1])
@pytest.mark.parametrize("B,n,G", [(3, 64, 256), (2, 12, 512), (1, 196, 256)])
def test_pair_reduce_bwd(H, code, B, n, G):
    dZ = formula.hash_uniform((B * n * n, G), 70, -1, 1)
    if code == 0:
        dZ = bf16_round(dZ)
    Rj = torch.empty(B * n, G, dtype=torch.float32, device="cuda")
    Ri = torch.empty_like(Rj); Rq = torch.empty(B, G, dtype=torch.float32, device="cuda")
    H.pair_reduce_bwd(dev(dZ).to(tdt(code)), G, Rj, Ri, Rq, code, B, n, G)
    d4 = dZ.reshape(B, n, n, G).astype(np.float64)
    assert rel(Rj.cpu().numpy().reshape(B, n, G), d4.sum(1)) <= F32_TOL
    assert rel(Ri.cpu().numpy().reshape(B, n, G), d4.sum(2)) <= F32_TOL
    assert rel(Rq.cpu().numpy(), d4.sum((1, 2))) <= F32_TOL
    Rq2 = torch.empty_like(Rq)
    H.pair_reduce_bwd(dev(dZ).to(tdt(code)), G, None, None, Rq2, code, B, n, G)
    assert torch.equal(Rq, Rq2)
    # the padded pair space of the factored-first-layer chain (njp = 32 ceil(n / 32) rows per (b, i) group): the rows j >= n are
    # not read (NaN here) and the sums are bitwise those of the dense matrix
    njp = (n + 31) // 32 * 32
    if njp != n:
        pad = np.full((B, n, njp, G), np.nan, np.float32)
        pad[:, :, :n] = dZ.reshape(B, n, n, G)
        Rj3 = torch.empty_like(Rj); Ri3 = torch.empty_like(Ri); Rq3 = torch.empty_like(Rq)
        H.pair_reduce_bwd(dev(pad.reshape(-1, G)).to(tdt(code)), G, Rj3, Ri3, Rq3, code, B, n, G, njp=njp)
        assert torch.equal(Rj3, Rj) and torch.equal(Ri3, Ri) and torch.equal(Rq3, Rq)


# ----------------------------------------------------------------------------- K4 small fp32
def test_gemm_f32_variants(H):
    B, K, N = 37, 250, 70
    a = formula.hash_uniform((B, K), 80); w = formula.hash_uniform((N, K), 81); bias = formula.hash_uniform((N,), 82)
    mul = formula.hash_uniform((B, N), 83, 0, 2); gate = formula.hash_uniform((B, N), 84)
    c = torch.empty(B, N, dtype=torch.float32, device="cuda")
    H.gemm_f32(dev(a), K, 1, dev(w), 1, K, c, N, B, N, K, bias=dev(bias), mul=dev(mul), ldmul=N, gate=dev(gate), ldgate=N, flags=H.RN_RELU)
    ref = np.maximum((a.astype(np.float64) @ w.T + bias) * mul, 0) * (gate > 0)
    assert rel(c.cpu().numpy(), ref) <= F32_TOL
    # transposed-A form (weight gradients) + accumulate
    g = formula.hash_uniform((B, N), 85)
    dw = torch.ones(N, K, dtype=torch.float32, device="cuda")
    H.gemm_f32(dev(g), 1, N, dev(a), K, 1, dw, K, N, K, B, flags=H.RN_ACCUMULATE)
    assert rel(dw.cpu().numpy(), g.astype(np.float64).T @ a + 1.0) <= F32_TOL
    # column-offset B operand (W[:, off:off+n])
    out = torch.empty(B, 20, dtype=torch.float32, device="cuda")
    H.gemm_f32(dev(g), N, 1, dev(w), K, 1, out, 20, B, 20, N, b_off=30)
    assert rel(out.cpu().numpy(), g.astype(np.float64) @ w[:, 30:50]) <= F32_TOL


def test_argument_errors_are_loud(H):
    P = torch.empty(16, 100, dtype=torch.bfloat16, device="cuda")
    x = torch.zeros(1, 4, 3, device="cuda")
    with pytest.raises(RuntimeError, match="multiple of 64"):
        H.pair_build_fwd(x, None, P, 0, 1, 4, 3, 0, 100)
    with pytest.raises(RuntimeError, match="GPU"):
        H.pair_build_fwd(x.cpu(), None, P, 0, 1, 4, 3, 0, 128)


# ----------------------------------------------------------------------------- conv stack: fused BatchNorm2d + ReLU
@pytest.mark.parametrize("N,hw", [(64, 128), (3, 32), (64, 64), (64, 32), (7, 96)])
def test_conv_bn_relu_block(N, hw):
    """ConvInputModel with the fused batch-norm / ReLU kernels (rn_convnorm.hip) against the same module run through
    the stock torch ops (reference model.py:22-35): outputs, running statistics, every gradient.  fp32 tolerance:
    only summation order differs (1e-4 relative in max-norm; the conv-bias gradient is rounding noise around zero
    in the stock path and exactly zero here)."""
    import relationnetworks_clevr_amd as pkg
    torch.manual_seed(5)
    a = pkg.ConvInputModel().cuda().train()
    b = pkg.ConvInputModel().cuda().train()
    b.load_state_dict(a.state_dict())
    img = torch.rand(N, 3, hw, hw, device="cuda")
    tgt = torch.randn(N, 24, hw // 16, hw // 16, device="cuda")
    def stock(m, x):                                          # the reference's op sequence (model.py:22-35) on m's own parameters
        for i in range(1, 5):
            x = torch.nn.functional.relu(m._modules["batchNorm%d" % i](m._modules["conv%d" % i](x)))
        return x
    ya = stock(a, img)
    yb = b(img)
    assert rel(yb.detach().cpu().numpy(), ya.detach().cpu().numpy()) <= F32_TOL
    (ya * tgt).sum().backward()
    (yb * tgt).sum().backward()
    for (na, pa), (nb, pb) in zip(a.named_parameters(), b.named_parameters()):
        ga, gb = pa.grad.cpu().numpy(), pb.grad.cpu().numpy()
        if na.startswith("conv") and na.endswith("bias"):
            assert np.all(gb == 0) and np.abs(ga).max() <= 1e-3 * max(np.abs(a._modules[na.split(".")[0]].weight.grad.cpu().numpy()).max(), 1e-30)
        else:
            # a 1e-6 difference in a conv output flips the ReLU of elements sitting at zero: the weight gradients of the
            # early layers see it amplified (max-norm 5e-3; measured 2.7e-3 with the direct convolutions)
            assert rel(gb, ga) <= 5e-3, (na, rel(gb, ga))
    for (na, ba), (nb, bb) in zip(a.named_buffers(), b.named_buffers()):
        assert rel(bb.float().cpu().numpy(), ba.float().cpu().numpy()) <= F32_TOL, na
    # evaluation mode: running statistics
    a.eval(); b.eval()
    with torch.no_grad():
        ea = stock(a, img)
        eb = b(img)
    assert rel(eb.cpu().numpy(), ea.cpu().numpy()) <= F32_TOL


# ----------------------------------------------------------------------------- question encoder: embedding + LSTM
@pytest.mark.parametrize("B,T", [(64, 20), (5, 7)])
def test_question_lstm(B, T):
    """QuestionEmbedModel through rn_lstm.hip against the stock torch.nn.Embedding + nn.LSTM path (reference
    model.py:39-58): final hidden state and every parameter gradient, fp32 (summation order only: 1e-4 in max-norm)."""
    import relationnetworks_clevr_amd as pkg
    torch.manual_seed(11)
    a = pkg.QuestionEmbedModel(formula.QDICT, 32, 128).cuda()
    b = pkg.QuestionEmbedModel(formula.QDICT, 32, 128).cuda()
    b.load_state_dict(a.state_dict())
    q = torch.from_numpy(formula.hash_ints((B, T), 900, 0, formula.QDICT + 1)).cuda()
    tgt = torch.randn(B, 128, device="cuda")
    ha = a.lstm(a.wembedding(q))[1][0][0]                     # the reference's op sequence (model.py:52-58) on a's own parameters
    hb = b(q)
    assert rel(hb.detach().cpu().numpy(), ha.detach().cpu().numpy()) <= F32_TOL
    (ha * tgt).sum().backward()
    (hb * tgt).sum().backward()
    for (na, pa), (nb, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert rel(pb.grad.cpu().numpy(), pa.grad.cpu().numpy()) <= 5 * F32_TOL, na
    with torch.no_grad():
        assert rel(b(q).cpu().numpy(), ha.detach().cpu().numpy()) <= F32_TOL     # inference entry (nothing saved)


# ----------------------------------------------------------------------------- conv stack: direct 3x3 / stride-2 convolutions
@pytest.mark.parametrize("N,Cin,Hh,Ww", [(64, 3, 128, 128), (64, 24, 64, 64), (5, 24, 32, 32), (3, 24, 16, 16), (2, 3, 12, 20), (2, 24, 6, 4), (3, 3, 64, 128), (2, 3, 132, 128)])
def test_conv3x3s2_bwd_weight(H, N, Cin, Hh, Ww):
    """The MFMA weight gradient of the conv stack against the library's (fp32, summation order differs)."""
    x = dev(formula.hash_uniform((N, Cin, Hh, Ww), 70, -1, 1))
    dy = dev(formula.hash_uniform((N, 24, Hh // 2, Ww // 2), 71, -1, 1))
    dw = torch.full((24, Cin, 3, 3), float("nan"), device="cuda")
    H.conv3x3s2_bwd_weight(x, dy, dw)
    torch.cuda.synchronize()
    ref = torch.ops.aten.convolution_backward(dy.double(), x.double(), torch.zeros(24, Cin, 3, 3, dtype=torch.float64, device="cuda"), None,
                                              (2, 2), (1, 1), (1, 1), False, (0, 0), 1, [False, True, False])[1]
    assert rel(dw.cpu().numpy(), ref.cpu().numpy()) <= 2e-6


@pytest.mark.parametrize("N,Cin,hw", [(64, 3, 128), (64, 24, 64), (3, 24, 16), (2, 3, 34), (32, 24, 72), (33, 24, 66), (32, 24, 112), (64, 24, 32), (64, 24, 16), (17, 24, 32), (16, 24, 16), (17, 3, 128)])
def test_conv3x3s2_direct(H, N, Cin, hw):
    """rn_conv.hip against torch's conv2d (MIOpen) and its input gradient: fp32, summation order only."""
    x = dev(formula.hash_uniform((N, Cin, hw, hw), 950, -1, 1))
    w = dev(formula.hash_uniform((24, Cin, 3, 3), 951, -0.3, 0.3))
    y = torch.empty(N, 24, hw // 2, hw // 2, device="cuda")
    H.conv3x3s2_fwd(x, w, y)
    ref = torch.nn.functional.conv2d(x, w, None, stride=2, padding=1)
    assert rel(y.cpu().numpy(), ref.cpu().numpy()) <= F32_TOL
    if Cin == 24:
        dy = dev(formula.hash_uniform((N, 24, hw // 2, hw // 2), 952, -1, 1))
        dx = torch.full_like(x, float("nan"))
        H.conv3x3s2_bwd_data(dy, w, dx)
        xr = x.clone().requires_grad_(True)
        torch.nn.functional.conv2d(xr, w, None, stride=2, padding=1).backward(dy)
        assert rel(dx.cpu().numpy(), xr.grad.cpu().numpy()) <= F32_TOL


@pytest.mark.parametrize("N,Cin,hw", [(64, 3, 128), (3, 3, 36), (5, 24, 32), (2, 3, 136), (3, 3, 128)])
def test_bn_relu_bwd_conv_wgrad(H, N, Cin, hw):
    """The first block's fused backward (batch-norm pass 2 inside the weight-gradient kernel) against the two separate entries:
    the same expressions on the same slice sums (small layers: the one-launch batch-norm kernel sums in another order), dw to fp32
    rounding (summation order of the products).  128-column images take the LDS-DMA kernel."""
    inp = dev(formula.hash_uniform((N, Cin, hw, hw), 960, -1, 1))
    xc = dev(formula.hash_uniform((N, 24, hw // 2, hw // 2), 961, -2, 2))
    dy = dev(formula.hash_uniform((N, 24, hw // 2, hw // 2), 962, -1, 1))
    gamma = dev(formula.hash_uniform((24,), 963, 0.5, 1.5)); beta = dev(formula.hash_uniform((24,), 964, -0.5, 0.5))
    mean = xc.mean((0, 2, 3)).contiguous(); invstd = torch.rsqrt(xc.var((0, 2, 3), unbiased=False) + 1e-5).contiguous()
    nan = lambda *sh: torch.full(sh, float("nan"), device="cuda")
    dx = nan(*xc.shape); dg0, db0, z0, dw0 = nan(24), nan(24), nan(24), nan(24, Cin, 3, 3)
    H.bn_relu_bwd(dy, xc, dx, gamma, beta, mean, invstd, dg0, db0, zero_out=z0)
    H.conv3x3s2_bwd_weight(inp, dx, dw0)
    dg1, db1, z1, dw1 = nan(24), nan(24), nan(24), nan(24, Cin, 3, 3)
    H.bn_relu_bwd_conv_wgrad(dy, xc, inp, gamma, beta, mean, invstd, dg1, db1, dw1, zero_out=z1)
    torch.cuda.synchronize()
    assert torch.equal(z1, torch.zeros_like(z1))
    if N * (hw // 2) ** 2 > 16384 * 4:
        assert torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert rel(dg1.cpu().numpy(), dg0.cpu().numpy()) <= 2e-6 and rel(db1.cpu().numpy(), db0.cpu().numpy()) <= 2e-6
    assert rel(dw1.cpu().numpy(), dw0.cpu().numpy()) <= 2e-6


def test_f_phi_nll_fused(H):
    """f_phi + log_softmax + mean NLL in one launch each way: loss and every gradient must equal the two-step path
    (rn_f_phi_fwd + rn_nll_mean_*); the completion counter must re-arm (second call)."""
    B, G, F1, F2, A = 37, 256, 256, 256, 28
    xg = dev(formula.hash_uniform((B, G), 500, -1, 1))
    fw = [dev(formula.hash_uniform(sh, 501 + i, -0.1, 0.1)) for i, sh in enumerate([(F1, G), (F2, F1), (A, F2)])]
    fb = [dev(formula.hash_uniform((n_,), 505 + i, -0.1, 0.1)) for i, n_ in enumerate([F1, F2, A])]
    mask = dev((formula.hash_uniform((B, F2), 509, 0, 1) > 0.5).astype(np.float32) * 2)
    label = torch.tensor(formula.hash_uniform((B,), 510, 0, A).astype(np.int64).clip(0, A - 1), device="cuda")
    wT = [w.t().contiguous() for w in fw]
    f32 = dict(dtype=torch.float32, device="cuda")
    for rep in range(2):
        f1 = torch.empty(B, F1, **f32); f2 = torch.empty(B, F2, **f32); out = torch.empty(B, A, **f32); loss = torch.full((), 9.0, **f32)
        H.f_phi_fwd_nll(xg, wT, fb, mask, label, f1, f2, out, loss, transposed=True)
        r1 = torch.empty(B, F1, **f32); r2 = torch.empty(B, F2, **f32); ro = torch.empty(B, A, **f32)
        H.f_phi_fwd(xg, wT, fb, mask, r1, r2, ro, transposed=True)
        torch.cuda.synchronize()
        assert torch.equal(out, ro) and torch.equal(f2, r2)
        ref_loss = -ro[torch.arange(B), label].double().mean()
        assert abs(float(loss) - float(ref_loss)) <= 1e-6 * abs(float(ref_loss))
    gl = torch.tensor(0.7, **f32)
    gout = torch.zeros(B, A, **f32); gout[torch.arange(B), label] = -0.7 / B
    mk = lambda: ([torch.empty_like(w) for w in fw], [torch.empty_like(b) for b in fb], torch.empty(B, G, **f32))
    dWa, dba, dxa = mk(); dWb, dbb, dxb = mk()
    H.f_phi_bwd_nll(gl, label, out, f2, f1, xg, fw, mask, dWa, dba, dxa)
    H.f_phi_bwd(gout, out, f2, f1, xg, fw, mask, dWb, dbb, dxb)
    torch.cuda.synchronize()
    for a, b in zip(dWa + dba + [dxa], dWb + dbb + [dxb]):
        assert rel(a.cpu().numpy(), b.cpu().numpy()) <= 1e-6


@pytest.mark.parametrize("B,G,F1,F2,A,use_mask", [(4, 512, 512, 1024, 28, True), (64, 512, 512, 1024, 28, True), (7, 256, 256, 256, 28, False),
                                                  (37, 512, 512, 1024, 30, True)])
def test_f_phi_wide_path_bitwise(H, B, G, F1, F2, A, use_mask):
    """The state-description models' f_phi (512 -> 512 -> 1024 -> 28: 3 MB of weights) runs as one launch PER LAYER, split over output
    features (rn_small.hip, fp_wide_*): forward with the loss and both backward entries equal the one-launch row-split kernels bit for
    bit (same k-slices, same fmaf chains, same slice-order sums), whichever path the size rule picks (rn_debug_f_phi_wide forces each);
    and the log-probs agree with an fp64 evaluation."""
    lib = H.load()
    xg = dev(formula.hash_uniform((B, G), 520, -1, 1))
    fw = [dev(formula.hash_uniform(sh, 521 + i, -0.06, 0.06)) for i, sh in enumerate([(F1, G), (F2, F1), (A, F2)])]
    fb = [dev(formula.hash_uniform((n_,), 525 + i, -0.1, 0.1)) for i, n_ in enumerate([F1, F2, A])]
    mask = dev((formula.hash_uniform((B, F2), 529, 0, 1) > 0.05).astype(np.float32) / 0.95) if use_mask else None
    label = torch.tensor(formula.hash_uniform((B,), 530, 0, A).astype(np.int64).clip(0, A - 1), device="cuda")
    wT = [w.t().contiguous() for w in fw]
    f32 = dict(dtype=torch.float32, device="cuda")
    gl = torch.tensor(0.7, **f32)
    gout = dev(formula.hash_uniform((B, A), 531, -1, 1))
    res = {}
    was = lib.rn_debug_f_phi_wide(0)
    try:
        for mode in (-1, 1):
            lib.rn_debug_f_phi_wide(mode)
            nan = lambda *sh: torch.full(sh, float("nan"), **f32)
            for rep in range(2):                                             # the loss counter re-arms
                f1, f2, out, loss = nan(B, F1), nan(B, F2), nan(B, A), nan()
                H.f_phi_fwd_nll(xg, wT, fb, mask, label, f1, f2, out, loss, transposed=True)
            e1, e2, eo = nan(B, F1), nan(B, F2), nan(B, A)
            H.f_phi_fwd(xg, wT, fb, mask, e1, e2, eo, transposed=True)
            mk = lambda: ([nan(*w.shape) for w in fw], [nan(*b.shape) for b in fb], nan(B, G))
            dWa, dba, dxa = mk(); dWb, dbb, dxb = mk()
            H.f_phi_bwd_nll(gl, label, out, f2, f1, xg, fw, mask, dWa, dba, dxa)
            H.f_phi_bwd(gout, out, f2, f1, xg, fw, mask, dWb, dbb, dxb)
            torch.cuda.synchronize()
            res[mode] = [f1, f2, out, loss, e1, e2, eo] + dWa + dba + [dxa] + dWb + dbb + [dxb]
    finally:
        lib.rn_debug_f_phi_wide(was)
    for a, b in zip(res[-1], res[1]):
        assert not torch.isnan(a).any() and torch.equal(a, b)
    f1, f2, out, loss = res[1][:4]
    x64 = xg.double().cpu().numpy()
    h1 = np.maximum(x64 @ fw[0].double().cpu().numpy().T + fb[0].double().cpu().numpy(), 0)
    z2 = h1 @ fw[1].double().cpu().numpy().T + fb[1].double().cpu().numpy()
    h2 = np.maximum(z2 * (mask.double().cpu().numpy() if use_mask else 1.0), 0)
    z3 = h2 @ fw[2].double().cpu().numpy().T + fb[2].double().cpu().numpy()
    ref = z3 - np.log(np.exp(z3 - z3.max(1, keepdims=True)).sum(1, keepdims=True)) - z3.max(1, keepdims=True)
    assert rel(out.cpu().numpy(), ref) <= 2e-6
    assert abs(float(loss) + ref[np.arange(B), label.cpu().numpy()].mean()) <= 2e-6 * abs(float(loss))


# ----------------------------------------------------------------------------- mean NLL loss
def test_nll_mean():
    """NllMeanFunction against F.nll_loss (mean): value and gradient (exact up to the final rounding)."""
    from relationnetworks_clevr_amd import functional as RF
    torch.manual_seed(3)
    logp = torch.log_softmax(torch.randn(64, 28, device="cuda"), 1)
    y = torch.randint(0, 28, (64,), device="cuda")
    a = logp.clone().requires_grad_(True); b = logp.clone().requires_grad_(True)
    la = torch.nn.functional.nll_loss(a, y); lb = RF.nll_loss_mean(b, y)
    assert abs(float(la) - float(lb)) <= 1e-6 * abs(float(la))
    (la * 3).backward(); (lb * 3).backward()
    assert torch.allclose(a.grad, b.grad, rtol=1e-6, atol=0)


@pytest.mark.parametrize("B", [1100, 2500])
def test_f_phi_large_batch_and_label_clamp(H, B):
    """f_phi + NLL for an evaluation-sized batch (the gradient kernel tiles the batch: B > 1024 used to be
    refused) against plain torch; labels outside [0, A) are CLAMPED to the nearest class, consistently in the fused forward
    and backward kernels (documented in INTEGRATION.md: a kernel cannot raise like F.nll_loss without a host sync)."""
    G, F1, F2, A = 256, 256, 256, 28
    xg = dev(formula.hash_uniform((B, G), 600, -1, 1))
    fw = [dev(formula.hash_uniform(sh, 601 + i, -0.1, 0.1)) for i, sh in enumerate([(F1, G), (F2, F1), (A, F2)])]
    fb = [dev(formula.hash_uniform((n_,), 605 + i, -0.1, 0.1)) for i, n_ in enumerate([F1, F2, A])]
    label = torch.tensor(formula.hash_uniform((B,), 610, 0, A).astype(np.int64).clip(0, A - 1), device="cuda")
    label[3], label[7], label[B - 1] = -100, A + 5, -1              # out of range: clamped to 0, A-1, 0
    clamped = label.clamp(0, A - 1)
    wT = [w.t().contiguous() for w in fw]
    f32 = dict(dtype=torch.float32, device="cuda")
    f1 = torch.empty(B, F1, **f32); f2 = torch.empty(B, F2, **f32); out = torch.empty(B, A, **f32); loss = torch.empty((), **f32)
    H.f_phi_fwd_nll(xg, wT, fb, None, label, f1, f2, out, loss, transposed=True)
    ps = [t.clone().requires_grad_(True) for t in [xg] + fw + fb]
    h = torch.relu(ps[0] @ ps[1].t() + ps[4]); h = torch.relu(h @ ps[2].t() + ps[5])
    ref = torch.log_softmax(h @ ps[3].t() + ps[6], 1)
    ref_loss = torch.nn.functional.nll_loss(ref, clamped)
    (ref_loss * 0.7).backward()
    assert rel(out.cpu().numpy(), ref.detach().cpu().numpy()) <= 1e-5
    assert abs(float(loss) - float(ref_loss)) <= 1e-5 * abs(float(ref_loss))
    dW = [torch.empty_like(w) for w in fw]; db = [torch.empty_like(b) for b in fb]; dxg = torch.empty(B, G, **f32)
    H.f_phi_bwd_nll(torch.tensor(0.7, **f32), label, out, f2, f1, xg, fw, None, dW, db, dxg)
    torch.cuda.synchronize()
    for got, want in zip([dxg] + dW + db, [p.grad for p in ps]):
        assert rel(got.cpu().numpy(), want.cpu().numpy()) <= 2e-4


@pytest.mark.parametrize("f16s", [True])
def test_e4m3_copies_saturate_instead_of_turning_into_nan(H, f16s):
    """Activations beyond the e4m3 range (448) must be stored as 448 (byte 0x7e), never as the NaN byte the raw conversion
    produces: a layer-0 bias of +600 drives every H_0 value there; the wgrad that reads the copies then stays finite."""
    B, n, L, G, k, Q = 1, 64, 4, 256, 26, 128
    M, kt = B * n * n, 2 * 26 + 128
    x = formula.hash_uniform((B, n, k), 500, -1, 1).astype(np.float32)
    q = formula.hash_uniform((B, Q), 501, -1, 1).astype(np.float32)
    Ws = [formula.hash_uniform((G, kt if l == 0 else G), 510 + l, -0.02, 0.02).astype(np.float32) for l in range(L)]
    bs = [np.full((G,), 600.0 if l == 0 else 0.1, np.float32) for l in range(L)]
    wd = [dev(w) for w in Ws]
    w0T = torch.empty(kt, G, device="cuda")
    dt16 = torch.float16 if f16s else torch.bfloat16
    Xp = torch.empty(B * n, 64, dtype=dt16, device="cuda"); Vc = torch.empty(B * n, G, device="cuda")
    Hs8 = [torch.zeros(M, G, dtype=torch.uint8, device="cuda").view(torch.float8_e4m3fn) for _ in range(3)] + [None]
    masks = list(torch.zeros(L, H.g_chain_rr_mask_bytes(M), dtype=torch.uint8, device="cuda"))
    part = torch.empty(M // 32, G, device="cuda")
    bd = [dev(b) for b in bs]
    if f16s:
        hi, lo, jobs = f16s_images(H, wd, kt, k)
        H.pack_matrix_frag_many(jobs + [(wd[0], kt, 1, G, kt, w0T, 2)])
        H.pair_tables(dev(x), dev(q), w0T, bd[0], Xp, Vc, B, n, k, Q, G)
        H.g_chain_fwd_rr_f16s_alg0(Xp, Vc, n, hi, lo, bd, Hs8, masks, part, M, G)
    torch.cuda.synchronize()
    b0 = Hs8[0].view(torch.uint8)
    assert bool((b0 == 0x7e).all()), torch.unique(b0).tolist()[:8]
    for l in (1, 2):
        assert not bool(((Hs8[l].view(torch.uint8) & 0x7f) == 0x7f).any()), l
    dZ = (torch.rand(M, G, device="cuda") - 0.5).bfloat16()
    dW = torch.empty(G, G, device="cuda"); db = torch.empty(G, device="cuda")
    H.g_wgrad_blocked([(H.rows_to_blocked(dZ), Hs8[0], dW, db)], M)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(dW).all())
    ref = (dZ.double().sum(0) * 448.0)[:, None].expand(G, G)
    assert rel(dW.cpu().numpy(), ref.cpu().numpy()) <= 1e-5


@pytest.mark.parametrize("B,parts,nll,tr", [(64, 16, True, True), (5, 7, True, True), (9, 16, False, True), (6, 3, True, False)])
def test_f_phi_from_partials(H, B, parts, nll, tr):
    """rn_f_phi_fwd_from_partials = rn_pair_sum_fwd + rn_f_phi_fwd(_nll) in one launch: the pair sums it writes agree with a
    float64 sum of the partial rows to fp32 rounding, and everything downstream -- bitwise -- with the two-launch path on those sums."""
    G, F1, F2, A = 256, 256, 256, 28
    g = torch.Generator(device="cuda").manual_seed(3)
    part = torch.randn(B * parts, G, device="cuda", generator=g)
    fw = [torch.randn(F1, G, device="cuda", generator=g) * 0.05, torch.randn(F2, F1, device="cuda", generator=g) * 0.05, torch.randn(A, F2, device="cuda", generator=g) * 0.05]
    w = [x.t().contiguous() for x in fw] if tr else fw
    fb = [torch.randn(F1, device="cuda", generator=g) * 0.1, torch.randn(F2, device="cuda", generator=g) * 0.1, torch.randn(A, device="cuda", generator=g) * 0.1]
    mask = (torch.rand(B, F2, device="cuda", generator=g) > 0.5).float() * 2
    label = torch.randint(0, A, (B,), device="cuda", generator=g) if nll else None
    mk = lambda *s: torch.full(s, float("nan"), device="cuda")
    xg, f1, f2, out = mk(B, G), mk(B, F1), mk(B, F2), mk(B, A)
    loss = mk() if nll else None
    H.f_phi_fwd_from_partials(part, parts, xg, w, fb, mask, label, f1, f2, out, loss, transposed=tr)
    torch.cuda.synchronize()
    ref = part.double().view(B, parts, G).sum(1)
    assert rel(xg.cpu().numpy(), ref.cpu().numpy()) <= 1e-6
    f1b, f2b, outb = mk(B, F1), mk(B, F2), mk(B, A)
    if nll:
        lossb = mk()
        H.f_phi_fwd_nll(xg, w, fb, mask, label, f1b, f2b, outb, lossb, transposed=tr)
    else:
        H.f_phi_fwd(xg, w, fb, mask, f1b, f2b, outb, transposed=tr)
    torch.cuda.synchronize()
    assert torch.equal(f1, f1b) and torch.equal(f2, f2b) and torch.equal(out, outb)
    if nll:
        assert torch.equal(loss, lossb)
    xg2 = mk(B, G)
    H.f_phi_fwd_from_partials(part, parts, xg2, w, fb, mask, label, mk(B, F1), mk(B, F2), mk(B, A), mk() if nll else None, transposed=tr)
    torch.cuda.synchronize()
    assert torch.equal(xg, xg2)
    if nll and tr:
        # the training step's variant: forward + the backward dz chain for d loss = 1 in ONE launch, the parameter gradients from
        # its workspace -- bitwise the forward above and rn_f_phi_bwd_nll with a loss gradient of 1
        xg3, f1c, f2c, outc, lossc, dxg = mk(B, G), mk(B, F1), mk(B, F2), mk(B, A), mk(), mk(B, G)
        ws = H.f_phi_fwd_bwd_from_partials(part, parts, xg3, w, fb, fw, mask, label, f1c, f2c, outc, lossc, dxg)
        dWa = [mk(*x.shape) for x in fw]; dba = [mk(x.shape[0]) for x in fw]
        H.f_phi_bwd_grads(ws, xg3, f1c, f2c, dWa, dba)
        dWb = [mk(*x.shape) for x in fw]; dbb = [mk(x.shape[0]) for x in fw]; dxgb = mk(B, G)
        H.f_phi_bwd_nll(torch.ones((), device="cuda"), label, out, f2, f1, xg, fw, mask, dWb, dbb, dxgb)
        torch.cuda.synchronize()
        assert torch.equal(xg3, xg) and torch.equal(f1c, f1) and torch.equal(f2c, f2) and torch.equal(outc, out) and torch.equal(lossc, loss)
        assert torch.equal(dxg, dxgb)
        for a, b in zip(dWa + dba, dWb + dbb):
            assert torch.equal(a, b)


@pytest.mark.parametrize("B,parts,use_mask,bwd", [(64, 16, True, True), (64, 16, False, False), (5, 7, True, True), (33, 0, True, True), (64, 0, False, False)])
def test_f_phi_split(H, B, parts, use_mask, bwd):
    """rn_f_phi_split -- f_phi as a feature-split fp32 MFMA chain in one launch with in-launch hand-offs (model.py:155-162, VERDICT r4
    item 5) -- against a float64 restatement of the same formulas (<= 2e-6: fp32 products in another summation order) and against
    the row-split kernels (rn_f_phi_fwd_bwd_from_partials: same bound); the pair sums = the partial rows added in order (bitwise the
    row-split kernel's); B < 64 leaves the rows beyond B untouched; a second launch on the same workspace (epoch 2) and 50 more
    under a competing stream give bitwise the first launch's results; no poll ever gave up (rn_f_phi_split_status == 0)."""
    G, F1, F2, A = 256, 256, 256, 28
    g = torch.Generator(device="cuda").manual_seed(5)
    part = torch.randn(B * max(parts, 1), G, device="cuda", generator=g)
    fw = [torch.randn(F1, G, device="cuda", generator=g) * 0.05, torch.randn(F2, F1, device="cuda", generator=g) * 0.05, torch.randn(A, F2, device="cuda", generator=g) * 0.05]
    fwT = [x.t().contiguous() for x in fw]
    fb = [torch.randn(F1, device="cuda", generator=g) * 0.1, torch.randn(F2, device="cuda", generator=g) * 0.1, torch.randn(A, device="cuda", generator=g) * 0.1]
    mask = (torch.rand(B, F2, device="cuda", generator=g) > 0.5).float() * 2 if use_mask else None
    label = torch.randint(0, A, (B,), device="cuda", generator=g)
    NAN = float("nan")
    mk = lambda *s: torch.full(s, NAN, device="cuda")
    sync = H.f_phi_split_sync_ws("cuda")

    def run():
        xg = mk(B, G) if parts else part.clone()
        f1, f2, out, loss = mk(B, F1), mk(B, F2), mk(B, A), mk()
        dxg = mk(B, G) if bwd else None
        ws = H.f_phi_split(part if parts else None, parts, xg, fw, fb, fwT, mask, label, f1, f2, out, loss, sync, dxg=dxg)
        torch.cuda.synchronize()
        dz = ws.view(torch.float32).clone() if bwd else None
        return xg, f1, f2, out, loss, dxg, dz
    xg, f1, f2, out, loss, dxg, dz = run()
    assert H.f_phi_split_ok(B, G, F1, F2, A) and not H.f_phi_split_ok(65, G, F1, F2, A) and not H.f_phi_split_ok(B, G, 512, F2, A)
    # float64 restatement
    d = lambda t: t.double()
    xr = d(part).view(B, parts, G).sum(1) if parts else d(part)
    m = d(mask) if use_mask else 1.0
    f1r = torch.relu(xr @ d(fw[0]).t() + d(fb[0]))
    f2r = torch.relu((f1r @ d(fw[1]).t() + d(fb[1])) * m)
    lpr = torch.log_softmax(f2r @ d(fw[2]).t() + d(fb[2]), 1)
    lossr = -lpr[torch.arange(B), label].mean()
    tol = 2e-6
    assert rel(xg.cpu().numpy(), xr.cpu().numpy()) <= 1e-6
    assert rel(f1.cpu().numpy(), f1r.cpu().numpy()) <= tol and rel(f2.cpu().numpy(), f2r.cpu().numpy()) <= tol
    assert rel(out.cpu().numpy(), lpr.cpu().numpy()) <= tol and abs(float(loss) - float(lossr)) <= tol * abs(float(lossr))
    if bwd:
        gl = torch.zeros(B, A, dtype=torch.float64, device="cuda"); gl[torch.arange(B), label] = -1.0 / B
        dz3r = gl - lpr.exp() * gl.sum(1, keepdim=True)
        dz2r = (dz3r @ d(fw[2])) * m * (f2r > 0)
        dz1r = (dz2r @ d(fw[1])) * (f1r > 0)
        dxgr = dz1r @ d(fw[0])
        dz1, dz2, dz3 = dz[:B * F1].view(B, F1), dz[B * F1:B * (F1 + F2)].view(B, F2), dz[B * (F1 + F2):B * (F1 + F2 + A)].view(B, A)
        for got, want in ((dz3, dz3r), (dz2, dz2r), (dz1, dz1r), (dxg, dxgr)):
            assert rel(got.cpu().numpy(), want.cpu().numpy()) <= 5e-6
    # the row-split kernels on the same inputs
    if parts and bwd and use_mask:
        xg3, f1c, f2c, outc, lossc, dxgc = mk(B, G), mk(B, F1), mk(B, F2), mk(B, A), mk(), mk(B, G)
        wsc = H.f_phi_fwd_bwd_from_partials(part, parts, xg3, fwT, fb, fw, mask, label, f1c, f2c, outc, lossc, dxgc)
        torch.cuda.synchronize()
        assert torch.equal(xg3, xg)                                   # (the same partial order)
        assert rel(f2.cpu().numpy(), f2c.cpu().numpy()) <= tol and rel(out.cpu().numpy(), outc.cpu().numpy()) <= tol
        assert rel(dxg.cpu().numpy(), dxgc.cpu().numpy()) <= 5e-6 and rel(dz.cpu().numpy(), wsc.view(torch.float32).cpu().numpy()) <= 5e-6
        # ... and the parameter gradients from the split launch's workspace
        ws2 = H.f_phi_split(part, parts, xg3, fw, fb, fwT, mask, label, f1c, f2c, outc, lossc, sync, dxg=dxgc)
        dWa = [mk(*x.shape) for x in fw]; dba = [mk(x.shape[0]) for x in fw]
        H.f_phi_bwd_grads(ws2, xg3, f1c, f2c, dWa, dba)
        torch.cuda.synchronize()
        for got, want in zip(dWa + dba, [dz1r.t() @ xr, dz2r.t() @ f1r, dz3r.t() @ f2r, dz1r.sum(0), dz2r.sum(0), dz3r.sum(0)]):
            assert rel(got.cpu().numpy(), want.cpu().numpy()) <= 5e-6
    # epoch 2, 3, ...: bitwise the same, also while another stream keeps the chip busy (uneven load: the hand-offs' worst case)
    first = [t_.clone() for t_ in (xg, f1, f2, out, loss) + ((dxg, dz) if bwd else ())]
    side = torch.cuda.Stream()
    big = torch.randn(8192, 8192, device="cuda")
    for it in range(50):
        if it % 5 == 0:
            with torch.cuda.stream(side):
                for _ in range(3):
                    big.mul_(1.0001)
        got = run()
        for a_, b_ in zip(first, [t_ for t_ in got if t_ is not None]):
            assert torch.equal(a_, b_), it
    torch.cuda.synchronize()
    assert H.f_phi_split_status() == 0


def test_copy_many(H):
    """The batch hand-off: up to four copies in one launch, sizes with and without a 16-byte tail; bytes outside stay untouched."""
    srcs = [torch.randn(64, 3, 128, 128, device="cuda"), torch.randint(0, 80, (64, 43), device="cuda"), torch.randint(0, 28, (64,), device="cuda"),
            torch.randint(0, 255, (16 * 1024 + 7,), dtype=torch.uint8, device="cuda")]
    pads = [torch.full((s.numel() * s.element_size() + 64,), 0xA5, dtype=torch.uint8, device="cuda") for s in srcs]
    dsts = [p[16:16 + s.numel() * s.element_size()].view(s.dtype).view(s.shape) for p, s in zip(pads, srcs)]
    H.copy_many(list(zip(dsts, srcs)))
    torch.cuda.synchronize()
    for d, s, p in zip(dsts, srcs, pads):
        assert torch.equal(d, s)
        assert bool((p[:16] == 0xA5).all()) and bool((p[16 + s.numel() * s.element_size():] == 0xA5).all())
    H.copy_many([(dsts[2], srcs[2])])
    with pytest.raises(ValueError):
        H.copy_many([(dsts[0], srcs[1])])


@pytest.mark.parametrize("B,T,V", [(64, 43, 82), (5, 7, 11), (200, 43, 82)])
def test_lstm_bwd_tail(H, B, T, V):
    """Embedding gradient + bias gradients in one launch: the embedding part bitwise the stand-alone entry (same order), the
    column sums against torch (fp64)."""
    idx = torch.tensor(formula.hash_uniform((B, T), 970, 0, V).astype(np.int64).clip(0, V - 1), device="cuda")
    dx = dev(formula.hash_uniform((T * B, 32), 971, -1, 1))
    dg = dev(formula.hash_uniform((T, B, 512), 972, -1, 1))
    d0 = torch.full((V, 32), float("nan"), device="cuda"); d1 = torch.full_like(d0, float("nan"))
    b1 = torch.full((512,), float("nan"), device="cuda"); b2 = torch.full_like(b1, float("nan")); b3 = torch.full_like(b1, float("nan"))
    H.embedding_bwd(idx, dx, d0)
    H.lstm_bwd_tail(idx, dx, d1, dg, b1, b2)
    H.lstm_bwd_tail(idx, None, None, dg, b3)
    torch.cuda.synchronize()
    assert torch.equal(d0, d1) and torch.equal(b1, b2) and torch.equal(b1, b3)
    ref = torch.zeros(V, 32, dtype=torch.float64, device="cuda").index_add_(0, idx.t().reshape(-1), dx.double())
    assert rel(d1.cpu().numpy(), ref.cpu().numpy()) <= F32_TOL
    assert rel(b1.cpu().numpy(), dg.double().sum((0, 1)).cpu().numpy()) <= F32_TOL


def test_dropout_mask_generator(H):
    """rn_dropout_mask: values in {0, 1 / (1 - p)}; keep rate = 1 - p to 4 sigma; the launch advances the device draw counter by one
    and leaves the completion word at zero; same (seed, draw) -> the same mask, the next draw / another seed -> another one; masks of
    successive draws are uncorrelated."""
    n, p = 64 * 256, 0.5
    st = torch.zeros(2, dtype=torch.int64, device="cuda")
    m0 = H.dropout_mask(torch.empty(64, 256, device="cuda"), p, 1234, st)
    torch.cuda.synchronize()
    assert st.tolist() == [1, 0]
    vals = set(m0.unique().tolist())
    assert vals == {0.0, 2.0}
    keep = float((m0 > 0).float().mean())
    assert abs(keep - (1 - p)) <= 4 * (p * (1 - p) / n) ** 0.5, keep
    m1 = H.dropout_mask(torch.empty(64, 256, device="cuda"), p, 1234, st)
    st2 = torch.zeros(2, dtype=torch.int64, device="cuda")
    m0b = H.dropout_mask(torch.empty(64, 256, device="cuda"), p, 1234, st2)
    m0c = H.dropout_mask(torch.empty(64, 256, device="cuda"), p, 99, torch.zeros(2, dtype=torch.int64, device="cuda"))
    torch.cuda.synchronize()
    assert st.tolist() == [2, 0] and torch.equal(m0, m0b) and not torch.equal(m0, m1) and not torch.equal(m0, m0c)
    agree = float(((m0 > 0) == (m1 > 0)).float().mean())
    assert abs(agree - 0.5) <= 4 * (0.25 / n) ** 0.5, agree
    for pp in (0.1, 0.75):
        m = H.dropout_mask(torch.empty(100003, device="cuda"), pp, 7, torch.zeros(2, dtype=torch.int64, device="cuda"))
        k_ = float((m > 0).float().mean())
        assert abs(k_ - (1 - pp)) <= 4 * (pp * (1 - pp) / 100003) ** 0.5 and abs(float(m.max()) - 1 / (1 - pp)) <= 1e-6
