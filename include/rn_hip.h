/* rn_hip.h -- C ABI of the MI355X (gfx950) Relation-Network hot-path library.
 *
 * The reference (mesnico/RelationNetworks-CLEVR) has no native code and no
 * FFI: its seam is the Python module model.py.  This
 * header is the boundary a maintainer binds with ctypes from that module
 * (see INTEGRATION.md); each entry point names the reference lines whose
 * stock-PyTorch op sequence it replaces.
 *
 * Conventions (all entry points):
 *   - plain C, POD arguments only: raw DEVICE pointers, sizes, element
 *     strides / leading dimensions (in ELEMENTS), a dtype enum and a
 *     hipStream_t passed as void*.
 *   - the library never allocates, frees, retains or synchronises device
 *     memory; every buffer (outputs, saved activations, workspace) is owned by
 *     the caller.  All work is enqueued on the caller's stream.
 *   - return value: 0 = OK; negative = argument error; positive = hipError_t.
 *     rn_last_error() returns a thread-local message for the last failure.
 *   - stateless and re-entrant; kernels are compiled for gfx950 (wave64) only.
 *
 * Storage dtypes of the pair / activation / packed-weight buffers:
 *   RN_BF16 : bf16 storage, bf16 MFMA (v_mfma_f32_32x32x16_bf16), fp32 accumulate.
 *   RN_F32  : fp32 storage, fp32 MFMA (v_mfma_f32_32x32x2_f32) -- exact-fp32 parity mode.
 *   RN_F32X3: fp32 storage, products on the bf16 pipe from operands split into hi + lo bf16 while they are staged (hi*hi + hi*lo +
 *             lo*hi, fp32 accumulate: 2^-16 of a product dropped) -- rn_g_linear_fwd / rn_g_linear_bwd_dgrad / rn_g_linear_bwd_wgrad; every other
 *             entry point takes RN_F32 for the same tensors.  The "bf16x3" precision of the 512-wide state-description models.
 *   RN_F16  : accepted by rn_pair_build_fwd / rn_pack_matrix only (an fp16 pair matrix: tools, K1 measurements).
 *   RN_FP8  : OCP e4m3 bytes -- the activation copies the forward chain keeps for the weight gradients (h_dtype / a_dtype).
 * Row index of every "pair" matrix: r = (b*n + i)*n + j   (model.py:127).
 */
#ifndef RN_HIP_H
#define RN_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RN_ABI_VERSION 9   /* bumped whenever a signature or a buffer layout of this header changes */

enum { RN_BF16 = 0, RN_F32 = 1, RN_F16 = 2, RN_FP8 = 3, RN_F32X3 = 4 };   /* RN_F16: pair matrix / split weights of the f16s forward only;
                                                             * RN_FP8: OCP e4m3 copies of the stored activations (h_dtype / a_dtype) */

/* rn_gemm_f32 flags */
enum { RN_RELU = 1, RN_ACCUMULATE = 2 };

int rn_abi_version(void);
const char* rn_last_error(void);

/* Host-side helper of the data-parallel trainer's fallback ladder (no reference counterpart; train.py:256-258 has no graphs): if
 * `stream` is in capture mode -- in particular if its capture was INVALIDATED by something the capture did not allow and the
 * framework's own end-of-capture threw before it reached hipStreamEndCapture -- end that capture and destroy whatever graph comes
 * back, so that the thread leaves (global) capture mode and eager launches work again.  Returns 0 when the stream is not
 * capturing afterwards (also when it never was), else the hipError_t. */
int rn_stream_abandon_capture(void* stream);

/* Bytes of caller-provided scratch (`ws`, `sync_ws`, mask buffers) an entry point needs for a shape -- ONE getter for all of them:
 * op names the entry point, a..d are its shape arguments in the order given here (unused ones 0). */
enum {
  RN_WS_RR_MASK = 0,          /* (M)                                  one lane-mask buffer of the register-resident chains       */
  RN_WS_PAIR_SUM = 1,         /* (B, npairs, G)                       rn_pair_sum_fwd                                             */
  RN_WS_WGRAD = 2,            /* (M, N, K)                            rn_g_linear_bwd_wgrad                                       */
  RN_WS_WGRAD_BLOCKED = 3,    /* (M, rows_per_question, njobs, aligned) rn_g_wgrad_blocked                                        */
  RN_WS_PAIR_REDUCE = 4,      /* (B, n, G)                            rn_pair_reduce_bwd                                          */
  RN_WS_WGRAD0 = 5,           /* (B, n, N)                            rn_wgrad0_from_reductions                                   */
  RN_WS_PAIR_FEATURES = 6,    /* (B, npairs, F)                       rn_pair_features                                            */
  RN_WS_EXTRACT = 7,          /* (B, n, F)                            rn_extract_features                                         */
  RN_WS_F_PHI_BWD = 8,        /* (B, F1, F2, A)                       rn_f_phi_bwd*, rn_f_phi_fwd_bwd_from_partials               */
  RN_WS_F_PHI_NLL = 9,        /* (B)                                  sync_ws of the f_phi launches that carry the loss           */
  RN_WS_CLIP_ADAM = 10,       /* ()                                   rn_clip_adam_step*                                          */
  RN_WS_CONV_BWD_WEIGHT = 11, /* (N, Cin, H, W)                       rn_conv3x3s2_bwd_weight, rn_bn_relu_bwd_conv_wgrad          */
  RN_WS_BN_RELU = 12,         /* (N, C, HW)                           rn_bn_relu_fwd / _bwd                                       */
  RN_WS_F_PHI_SPLIT = 13      /* ()                                   sync_ws of rn_f_phi_split (zeroed ONCE by the caller)       */
};
size_t rn_workspace_bytes(int op, int a, int b, int c, int d);

/* K1 -- fused pair-expand + coord-tag carry + question broadcast.
 * Replaces model.py:112-127 (+ :135-140 when the question is injected at g layer 0).
 *   P[r, 0:k]      = x[b, j, :]      P[r, k:2k] = x[b, i, :]
 *   P[r, 2k:2k+Q]  = q[b, :]  (Q == 0: no question columns)     P[r, 2k+Q:ld] = 0
 * x: fp32 with element strides (sxb, sxn, sxk) -- accepts both a contiguous (B,n,k)
 * tensor and the permuted (B,k,n) view RN.forward produces (model.py:200-201).
 * q: fp32 (B,Q), row stride sqb.  P: (B*n*n, ld) of `dtype`, ld % 64 == 0. */
int rn_pair_build_fwd(const float* x, long sxb, long sxn, long sxk, const float* q, long sqb,
                      void* P, int dtype, int B, int n, int k, int Q, int ld, void* stream);

/* Question broadcast into columns [col0, col0+Q) of a wide activation buffer
 * A (B*n*n, ld) -- model.py:135-140 for question_injection_position > 0. */
int rn_qst_broadcast(const float* q, long sqb, void* A, int dtype, int B, int n, int Q, int col0, int ld,
                     void* stream);

/* Pack a fp32 matrix into a zero-padded `dtype` matrix: dst[r][c] = src[r*sr + c*sc]
 * for r < R, c < C; zero for C <= c < ld and R <= r < Rpad.  Used for nn.Linear
 * weights (out,in) (model.py:96-99) -> MFMA operand layout, plain or transposed. */
int rn_pack_matrix(const float* src, long sr, long sc, int R, int C, void* dst, int dtype, int ld, int Rpad,
                   void* stream);

/* K2 -- one g_theta layer:  H = relu(A @ W^T + bias)      (model.py:141-145)
 * A: (M, lda) dtype, reduction length K (K % 64 == 0, columns >= true K are zero),
 * Wp: (N, ldw) packed dtype (rn_pack_matrix), bias: fp32 (N), H: (M, ldh) dtype.
 * N % 64 == 0.  Workgroup tiles: 128 x 256 when there are at least four of those per CU (N % 256 == 0), 64 x 64 otherwise (the
 * state-description models' short pair matrices) -- the same k-ordered sums either way, bit for bit. */
int rn_g_linear_fwd(const void* A, int lda, const void* Wp, int ldw, const float* bias, void* H, int ldh,
                    int dtype, int M, int N, int K, void* stream);

/* Register-resident chains (rn_chain_rr.hip): the headline shape family -- L == 4, G == 256, at most 32 features per object,
 * question injected at layer 0 or 2.  8 waves x 32 pair rows per workgroup; the activation stays in MFMA operand registers from
 * layer to layer, LDS carries only the weight stream.  The forward chain is rn_g_chain_fwd_rr_f16s_alg0 (declared with the tables
 * of the factored first layer, below); what it leaves for the backward pass:
 *   H[0..2]: copies of the activations as ROW-BLOCKED images (see rn_g_wgrad_blocked, their only reader), e4m3 or bf16;
 *   mask:    four buffers of rn_workspace_bytes(RN_WS_RR_MASK, M) = 32 M bytes: the ReLU gate (pre-activation > 0) of every
 *            element of layer l as 64-bit LANE masks in the kernel's own accumulator layout (opaque; the consumers are
 *            rn_g_chain_bwd_rr* and rn_fp8_copy_health);
 *   xg_part: fp32 sums of the UN-rounded last activation, one row per 256-row tile.
 * rn_g_chain_rr_tile() = 256: the tile height (M and, with the question at layer 2, n*n must be multiples of it). */
int rn_g_chain_rr_tile(void);

/* Backward (pair-sum broadcast + ReLU gates + the three dgrad steps):
 *   dZ[0]   = dxg[b] * gate_3                         b = question of the pair row
 *   dZ[s+1] = (dZ[s] @ W_{3-s}) * gate_{2-s}          s = 0, 1, 2
 * gate_l = mask[l] of the forward call.  Wtf[s]: fragment-major image of W_{3-s}^T, i.e.
 * an entry (W, sr = 1, sc = in_features, 256, 256, dst, natural = (s == 0)) of rn_pack_matrix_frag_many.  All four dZ (M, 256) bf16 are
 * written: dZ[0..2] (layers 3..1, read only by rn_g_wgrad_blocked) as ROW-BLOCKED 16-bit images, dZ[3] (layer 0, read by the
 * pair reduction) row-major.  dZ[0] may be NULL (rn_g_wgrad_blocked's gate job works from the masks instead).  rows_per_question = n*n (any
 * value dividing M). */
int rn_g_chain_bwd_rr(const float* dxg, const void* const* mask, const void* const* Wtf, void* const* dZ, int M,
                      int rows_per_question, int L, int G, void* stream);
/* The same chain with the pair-axis reductions of the expansion backward (model.py:117-127) formed ON CHIP: the gradient of layer
 * 0's pre-activation (dZ[3] above: 134 MB at B = 64, n = 64) is neither written nor read back -- its only readers are
 *   Rj[b,j,:] = sum_i dZ_0[(b,i,j),:]   and   Ri[b,i,:] = sum_j dZ_0[(b,i,j),:]
 * and the kernel leaves fp32 PARTIALS of both, summed from the un-rounded accumulators in a fixed order (bitwise reproducible):
 *   rj_part (records, 32, 256): unit u = ((b * n/32 + jg) * nu + v), nu = (n / 8) / tiles_per_unit, holds the sum
 *            over the i of its tiles_per_unit tiles (8 consecutive i each) for the 32 objects j of block jg;
 *            BALANCED TAIL (ABI 9): the kernel walks the units in the order p = (v * n/32 + jg) * B + b (question fastest).  Units at
 *            walk positions p < units_whole run as a whole and leave record u; the tiles of the units behind them are handed to the
 *            workgroups one by one (U units on C CUs: the last U mod C units would otherwise occupy U mod C CUs for a whole unit
 *            while the rest idle) and leave a record EACH: tile 0 in record u, tile t > 0 in record
 *            nunits + (p - units_whole) (tiles_per_unit - 1) + t - 1 -- spread evenly over the questions, which is what
 *            rn_pair_reduce_parts (one workgroup per question) needs.  records = nunits + (nunits - units_whole) (tiles_per_unit - 1);
 *            rn_g_chain_bwd_rr_red_whole(M, n, njp, tiles_per_unit) returns the library's choice (C floor(U / C); -1: shape not
 *            supported); any value in [0, nunits] is valid (nunits = no tail) as long as rn_pair_reduce_parts gets the same one;
 *   ri_part (M / 16, 256): rows ((b*n + i) * n/32 + jg) * 2 + {0, 1} = the sums over the two 16-object halves of block jg.
 * rn_pair_reduce_parts adds them up to Rj, Ri (B*n, 256) and Rq (B, 256) -- what rn_pair_reduce_bwd produces from a stored dZ_0.
 * Needs the masks of a forward call with the same M and njp, dZ[0] == NULL (the gate job of rn_g_wgrad_blocked stands in for it)
 * and dZ[1], dZ[2] as above; dZ[3] is ignored.  tiles_per_unit: any divisor of ceil(n / 8) -- rn_g_chain_bwd_rr_red_tpu(M, n, njp)
 * returns the largest one that still gives every CU a unit (0: shape not supported).
 * PADDED j axis (njp = 32 ceil(n / 32) > n, e.g. the 14 x 14 grid: n = 196, njp = 224; M = B * n * njp): the same kernel -- a
 * (question, i) group is njp / 32 wave-tiles, the rows j >= n have cleared mask bits and come out as exact zeros, and n need not
 * be a multiple of 8: the last tile of a (question, j block) holds fewer than 8 values of i (its spare waves repeat the last one
 * and contribute zeros).  Index formulas with n/32 -> njp/32, n/8 -> ceil(n/8): rj_part (B * njp/32 * nu, 32, 256),
 * ri_part (M / 16, 256); rn_pair_reduce_parts reads only the rows j < n. */
int rn_g_chain_bwd_rr_red_tpu(int M, int n, int njp);
int rn_g_chain_bwd_rr_red_whole(int M, int n, int njp, int tiles_per_unit);
int rn_g_chain_bwd_rr_red(const float* dxg, const void* const* mask, const void* const* Wtf, void* const* dZ, int M, int n, int njp, int L,
                          int G, float* rj_part, float* ri_part, int tiles_per_unit, int units_whole, void* stream);
int rn_pair_reduce_parts(const float* rj_part, const float* ri_part, float* Rj, float* Ri, float* Rq, int B, int n, int njp, int G, int nu,
                         int tiles_per_unit, int units_whole, void* stream);
/* MFMA-fragment-major weight images for the register-resident chains, `count` (<= 16) of them in ONE launch; all arguments are
 * HOST arrays of `count` entries.  Image i: dst (65536 elements) gets, for output block ob, K16 step ks, lane, element e:
 * src[32 ob + lane % 32][kidx] (0 beyond R rows / C columns) with
 *   natural & 1:    kidx = 16 ks + 8 (lane / 32) + e                            (the operand is read from memory rows)
 *   else:           kidx = 32 (ks / 2) + 4 (lane / 32) + 8 (2 (ks % 2) + e / 4) + e % 4   (the operand is the previous MFMA's output)
 * src: fp32, element (r, c) at src[r * sr + c * sc] (model.py:96-99 nn.Linear weight: sr = in, sc = 1).  Element type by natural[i]:
 *   0 / 1:            bf16 (the backward chain's W^T images);
 *   2:                a plain fp32 TRANSPOSE instead: dst (C, R) fp32 row-major = src^T (the f_phi weights, W_0^T of the tables);
 *   4 | n, 8 | n:     the fp16 hi = fp16(w) / lo = fp16(w - hi) image in K order n (layer 0 of the f16s forward chain);
 *   4 | n | V << 8:   V (2, 4, 8) TILE-DITHERED hi images, 65536 fp16 apart: image d = fp16(w + ((d + 1/2) / V - 1/2) ulp_fp16(w))
 *                     (layers 1..3 of the f16s forward chain, see rn_g_chain_fwd_rr_f16s_alg0). */
int rn_pack_matrix_frag_many(const float* const* src, const long* sr, const long* sc, const int* R, const int* C,
                             void* const* dst, const int* natural, int count, void* stream);

/* K3 -- sum over the n*n pairs of every question: xg[b,:] = sum_p HL[b*npairs+p, :]
 * (model.py:151-152).  ws: >= rn_workspace_bytes(RN_WS_PAIR_SUM, B, npairs, G) bytes of scratch. */
int rn_pair_sum_fwd(const void* HL, int ldh, float* xg, void* ws, int dtype, int B, int npairs, int G,
                    void* stream);

/* ... of the two-rows-per-tile partials of rn_g_chain_fwd_rr_f16s_alg0 with a padded j axis: part (M / 256 * 2, 256) fp32, row
 * 2 t + s = the sum over the rows of tile t that belong to question floor(256 t / rows_per_question) + s;
 * xg[b, :] = the sum of its tiles' rows, in tile order.  M % rows_per_question == 0, M % 256 == 0. */
int rn_pair_sum_tiles(const float* part, float* xg, int M, int rows_per_question, int G, void* stream);

/* Backward of K3 fused with the last layer's ReLU gate:
 * dZ[b*npairs+p, c] = dxg[b, c] * (HL[b*npairs+p, c] > 0). */
int rn_pair_sum_bwd(const float* dxg, const void* HL, int ldh, void* dZ, int lddz, int dtype, int B,
                    int npairs, int G, void* stream);

/* dgrad of one g layer fused with the previous layer's ReLU gate:
 *   dZprev = (dZ @ W[:, :Kin]) * (Hprev > 0)
 * dZ: (M, lddz), reduction length N (layer width, N % 64 == 0); Wt: (Kin, ldwt) packed
 * transposed weight (Wt[k][n] = W[n][k]); Hprev, dZprev: (M, Kin), Kin % 64 == 0 (tiles as above). */
int rn_g_linear_bwd_dgrad(const void* dZ, int lddz, const void* Wt, int ldwt, const void* Hprev, int ldhp,
                          void* dZprev, int lddzp, int dtype, int M, int N, int Kin, void* stream);

/* wgrad of one g layer: dW[n, k] = sum_m dZ[m, n] * A[m, k] (k < Ktrue), db[n] = sum_m dZ[m, n]  (autograd of model.py:141-145).
 * dZ: (M, lddz) width N (N % 256 == 0); A: (M, lda) with K padded columns (K % 32 == 0), both ROW-MAJOR of `dtype`;
 * dW: fp32 (N, Ktrue) contiguous (nn.Linear layout); db: fp32 (N).
 * Deterministic: split over M into per-block partials in ws, then an ordered reduction. */
int rn_g_linear_bwd_wgrad(const void* dZ, int lddz, const void* A, int lda, float* dW, float* db, void* ws,
                          int dtype, int M, int N, int K, int Ktrue, void* stream);

/* ROW-BLOCKED operand images.  The register-resident chains (rn_g_chain_fwd_rr_f16s_alg0, rn_g_chain_bwd_rr*) store what ONLY the weight
 * gradient reads -- the activation copies H[0..2] and the gradients dZ[0..2] (layers 3..1) -- in the layout that product wants,
 * 16 bytes = consecutive pair rows of ONE feature (= one lane's MFMA operand when the contraction runs over the rows):
 *   16-bit image (bf16) of an (M, 256) matrix:  element (m, f) at ((m / 8) * 256 + f) * 8 + m % 8
 *   e4m3 image (H copies with h_dtype = RN_FP8): byte    (m, f) at ((m / 16) * 256 + f) * 16 + m % 16
 * Same byte counts as the row-major matrices.  (rn_rows_to_blocked of include/rn_hip_debug.h converts either way: tests, tools.) */

/* Weight gradients of up to 4 256-wide g layers in ONE launch (+ one reduction launch) on row-blocked images:
 *   job j:  dW[j] (256, 256) = dZ[j]^T A[j],  db[j] (256) = column sums of dZ[j]          (model.py:141-145 autograd)
 * dZ[j]: 16-bit image (dz_dtype[j] = RN_BF16); A[j]: image of the layer's input, a_dtype = RN_BF16 or RN_FP8 (all jobs alike).
 * dz_dtype[j] = RN_FP8 (needs a_dtype = RN_FP8): a GATE job -- the LAST layer, whose gradient is never stored:
 * dZ_3[(b, pair), f] = gate[(b, pair), f] * dxg[b][f].  The gate is read from the SIGN BITS of A[j] (the forward chain called with
 * gate_in_h2 != 0 writes H_2 that way: byte (m, f) = e4m3(H_2[m, f]) | gate_3[m, f] << 7); dZ[j] must be NULL or A[j].  Gate and
 * activation bytes meet on the fp8 matrix pipe (exact products), and each question's sums are scaled by its row of dxg
 * (M / rows_per_question, 256) fp32 -- un-rounded, i.e. closer to the fp32 reference than a stored bf16 dZ_3 -- when the
 * question ends; rows_per_question % 64 == 0 then (otherwise it only steers the row splits; 0 = unknown).  M % 64 == 0.
 * Row splits.  aligned != 0 (or a_dtype = RN_BF16): Z = rn_wgrad_blocked_splits(M, rows_per_question, njobs, aligned) splits per job
 * (0: shape not covered), four workgroups (a 128 x 128 block of dW each) per split, about 48 / njobs so that the njobs x Z x 4
 * workgroups fill three quarters of the chip and all jobs stream at the same time; the count never lets a split straddle two questions
 * (Z = B * d for a divisor d of the 64-row steps per question, or B itself) when one exists within 256.  aligned == 0 on e4m3 images
 * (the product launch, round 6): a stored-gradient job runs as WIDE units -- ONE workgroup per split holds the whole 256 x 256 dW
 * (128 x 128 per wave in accumulator registers, every operand byte read once, db from in-lane dot products) --, a gate job keeps
 * the four-workgroup form, and the library picks the split counts of both kinds so that their workgroups finish together on the
 * same budget of workgroups (not exported: nothing outside reads them).
 * ws: rn_workspace_bytes(RN_WS_WGRAD_BLOCKED, M, rows_per_question, njobs, aligned) bytes (sized for the most splits any mix of job
 * kinds can get).  With aligned != 0, afterwards ws holds, at rn_wgrad_blocked_db_partials_offset(M, rows_per_question, njobs,
 * aligned, j), (Z, 4, 256) fp32: four partial column sums of dZ[j] over the 64-row steps [z*S/Z, (z+1)*S/Z) (S = M / 64) of split z
 * -- per-question sums of dZ for free, the splits being question-aligned.  Deterministic either way (fp32 partial tiles, fixed-order
 * reduction launch: bitwise repeatable).  dZ / dz_dtype / A / dW / db: HOST arrays. */
int rn_wgrad_blocked_splits(int M, int rows_per_question, int njobs, int aligned);
size_t rn_wgrad_blocked_db_partials_offset(int M, int rows_per_question, int njobs, int aligned, int job);
int rn_g_wgrad_blocked(const void* const* dZ, const int* dz_dtype, const void* const* A, int a_dtype, const float* dxg,
                       int rows_per_question, int aligned, float* const* dW, float* const* db, int njobs, void* ws, int M, void* stream);
/* Health of an e4m3 activation copy H_l, l = 0..2 (h_dtype = RN_FP8; fixed scale 1: values below 2^-10 flush to zero, values
 * above 448 are clamped): mask = the layer's lane masks of the SAME forward call (which elements were positive before rounding),
 * img = its row-blocked e4m3 image.  out4 (4 x uint64, ZEROED by the caller, accumulated with integer atomics): positive elements,
 * positive elements whose byte is 0 (flushed), bytes at the clamp 0x7e, largest byte.  The caller decides what to do about a
 * flushed fraction that is too large (dp.DataParallelTrainer switches to 16-bit copies). */
int rn_fp8_copy_health(const void* mask, const void* img, unsigned long long* out4, int M, void* stream);
/* Rq[b, f] (B, 256) fp32 = sum over the rows of question b of a 16-bit row-blocked image -- the per-question sums of dZ when
 * the splits above straddle questions.  rows_per_question % 8 == 0. */
int rn_blocked_question_sums(const void* img, float* Rq, int M, int rows_per_question, void* stream);

/* Backward of the pair expansion, algebraic form: reduce the gradient
 * of a g layer's pre-activation over the pair axes instead of materialising dP:
 *   Rj[b,j,:] = sum_i dZ[(b,i,j),:]   Ri[b,i,:] = sum_j dZ[(b,i,j),:]   Rq[b,:] = sum_ij dZ[(b,i,j),:]
 * Any of Rj / Ri / Rq may be NULL.  Outputs fp32; Rj, Ri: (B,n,G); Rq: (B,G). */
/* njp >= n: pair rows per (question, i) group -- n for the plain n*n pair matrix, 32 ceil(n / 32) for the padded pair space of
 * rn_g_chain_fwd_rr_f16s_alg0 (row (b, i, j) at (b*n + i) * njp + j; the rows j >= n are not read). */
int rn_pair_reduce_bwd(const void* dZ, int lddz, float* Rj, float* Ri, float* Rq, void* ws, int dtype, int B,
                       int n, int njp, int G, void* stream);

/* Input gradients of the pair expansion from the reductions, one launch (question injected at layer 0; W0 (N, 2k+Q) fp32):
 *   dx[b, j, c] = (Rj W0[:, 0:k] + Ri W0[:, k:2k])[b*n + j, c] for c < kout, written at element strides (sdb, sdn, sdk) -- e.g.
 *   straight into the (B, 24, d, d) layout the conv stack's backward reads, without the two coordinate columns (kout = 24,
 *   model.py:216: they carry no gradient);   dq (B, Q) = Rq W0[:, 2k:2k+Q]   (Q == 0: Rq / dq may be NULL). */
int rn_pair_dx_dq(const float* Rj, const float* Ri, const float* Rq, const float* W0, float* dx, long sdb, long sdn, long sdk, int kout,
                  float* dq, int B, int n, int k, int Q, int N, void* stream);

/* Layer-0 weight gradient from the pair reductions (the question injected at layer 0: P = [x_j | x_i | q]):
 *   dW0[:, 0:k] = Rj^T X,  dW0[:, k:2k] = Ri^T X,  dW0[:, 2k:2k+Q] = Rq^T q,  db0 = sum_b Rq[b]
 * with X = x viewed as (B*n, k) -- identical to dZ_0^T P (rounding aside: x enters in fp32 instead of P's storage
 * dtype) without reading dZ_0 or P.  Rj, Ri (B*n, N), Rq (B, N) fp32 from rn_pair_reduce_bwd; x (B, n, k) element
 * strides; q (B, Q) row stride sqb; dW0 (N, 2k+Q), db0 (N); k <= 32; ws: rn_workspace_bytes(RN_WS_WGRAD0, B, n, N) bytes.
 * Q == 0 (no question at layer 0): q may be NULL; Rq (all-pairs sums) still gives db0, NULL leaves db0 = 0.
 * coord != NULL: x supplies only the first kf columns of an object, the rest are the coordinate tags coord (k - kf, n), as in
 * rn_pair_tables. */
int rn_wgrad0_from_reductions(const float* Rj, const float* Ri, const float* Rq, const float* x, long sxb, long sxn, long sxk,
                              const float* coord, int kf, const float* q, long sqb, float* dW0, float* db0, void* ws, int B, int n,
                              int k, int Q, int N, void* stream);

/* K4 -- small fp32 GEMM on the fp32 MFMA, used for f_phi (model.py:155-160), its backward
 * and the (B*n x G) tail of the pair backward:
 *   C[m,n] (+)= epi( sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn] )
 *   epi: + bias[n]; * mul[m*ldmul + n] (dropout mask incl. 1/(1-p), model.py:158);
 *        relu (flags & RN_RELU); * (gate[m*ldgate + n] > 0) (ReLU backward).
 * Order: bias -> mul -> relu -> gate.  flags & RN_ACCUMULATE adds into C. */
int rn_gemm_f32(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long ldc,
                int M, int N, int K, const float* bias, const float* mul, long ldmul, const float* gate,
                long ldgate, int flags, void* stream);

/* R-CBIR pair features (reference extract.py:60-71): for the INPUT of a g layer, A (B*npairs, lda)
 * in `dtype`, first F columns (the question columns of an injection layer excluded by the caller's F):
 * L2-normalise every pair row (F.normalize semantics, eps 1e-12), then maxf / avgf (B, F) fp32 = maximum / mean over the
 * npairs rows of each question.  One pass over A.  F % 64 == 0, F <= 512; ws: rn_workspace_bytes(RN_WS_PAIR_FEATURES, B, npairs, F). */
/* The same features WITHOUT the matrix (SURVEY.md 8f row N3 as specified): the input of g layer `nlayers` is formed tile by tile
 * on chip -- a workgroup builds 64 pair rows [x_j | x_i] (model.py:117-127) in LDS, runs g layers 0 .. nlayers-1 on them in fp32
 * (v_mfma_f32_32x32x2_f32) with the activation tile resident in LDS, L2-normalises the rows of the result over its first F columns
 * and leaves one (max, sum) pair per tile in ws; a finish launch reduces per question.  Neither (B n^2, in) nor any activation
 * is written to memory.  nlayers = 0: the features are the pair rows themselves (F <= 2k).
 *   x: objects (B, n, k) fp32, element strides; Wt[l]: (K_l, 256) fp32 TRANSPOSED weights of layer l -- Wt[l][c][f] = W_l[f][c]
 *   for the first K_0 = 2k (l = 0) / K_l = 256 (l > 0) input columns, i.e. without the question columns; bias[l]: (256) fp32, or
 *   with bias_per_question[l] != 0 a (B, 256) table W_l[:, K_l:] q_b + b_l -- how the question enters at its injection layer
 *   (model.py:131-142).  g width 256, nlayers <= 4, 2k <= 256.  maxf, avgf: (B, F) fp32; ws: rn_workspace_bytes(RN_WS_EXTRACT, B, n, F). */
int rn_extract_features(const float* x, long sxb, long sxn, long sxk, const float* const* Wt, const float* const* bias,
                        const int* bias_per_question, int nlayers, int F, float* maxf, float* avgf, void* ws, int B, int n, int k,
                        void* stream);
int rn_pair_features(const void* A, int lda, int F, float* maxf, float* avgf, void* ws, int dtype, int B, int npairs, void* stream);

/* f_phi + log_softmax (model.py:155-162) in one launch, its backward in two (rn_small.hip; fp32 FMA):
 *   f1 = relu(xg W1^T + b1) (B, F1);  f2 = relu((f1 W2^T + b2) * mask) (B, F2);  out = log_softmax(f2 W3^T + b3) (B, A)
 * W_l: nn.Linear layout (out, in) row-major -- or, with transposed != 0 in the forward call, (in, out) copies (fp32 transpose
 * mode of rn_pack_matrix_frag_many: coalesced weight reads, 41 -> ~12 us); mask: (B, F2) dropout mask already scaled by 1/(1-p), or NULL.
 * Backward: gout = d loss / d out; writes dW_l, db_l and dxg (B, G); ws: rn_workspace_bytes(RN_WS_F_PHI_BWD, B, F1, F2, A) bytes.
 * Widths are multiples of 4 (A excepted) and <= 1024, B <= 1024. */
int rn_f_phi_fwd(const float* xg, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                 const float* b3, const float* mask, float* f1, float* f2, float* out, int transposed, int B, int G, int F1,
                 int F2, int A, void* stream);
/* rn_f_phi_fwd(_nll) with rn_pair_sum_fwd folded in: xg_part (B * parts_per_row, G) fp32 = the forward chains' partial pair sums
 * (one row per 256-row tile on the factored-first-layer paths), xg (B, G) is an OUTPUT (the sums, partial rows added in order:
 * deterministic).  label / loss / sync_ws: all three (loss folded in, as rn_f_phi_fwd_nll) or all NULL. */
int rn_f_phi_fwd_from_partials(const float* xg_part, int parts_per_row, float* xg, const float* W1, const float* b1, const float* W2,
                               const float* b2, const float* W3, const float* b3, const float* mask, const long long* label, float* f1,
                               float* f2, float* out, float* loss, void* sync_ws, int transposed, int B, int G, int F1, int F2, int A,
                               void* stream);
/* The training step's f_phi in ONE launch up to dxg: rn_f_phi_fwd_from_partials (transposed forward weights W1T..W3T, loss folded
 * in) followed, in the same kernel and for the same rows, by the backward dz chain for d loss = 1 -- the log-prob gradient of a
 * mean NLL is -1/B at the label whatever happens in between.  W1..W3: the natural (out, in) weights; bwd_ws:
 * rn_workspace_bytes(RN_WS_F_PHI_BWD, B, F1, F2, A) bytes, receives the dz rows; dxg (B, G) out.  rn_f_phi_bwd_grads then turns bwd_ws into
 * the six parameter gradients (exactly the second launch of rn_f_phi_bwd_nll).  A loss gradient other than 1: rn_f_phi_bwd_nll. */
int rn_f_phi_fwd_bwd_from_partials(const float* xg_part, int parts_per_row, float* xg, const float* W1T, const float* b1, const float* W2T,
                                   const float* b2, const float* W3T, const float* b3, const float* W1, const float* W2, const float* W3,
                                   const float* mask, const long long* label, float* f1, float* f2, float* out, float* loss, void* sync_ws,
                                   void* bwd_ws, float* dxg, int B, int G, int F1, int F2, int A, void* stream);
/* f_phi as a FEATURE-SPLIT fp32 MFMA chain in ONE launch (model.py:155-162; rn_fphi.hip): 16 workgroups, workgroup w owns output
 * features 16 w .. 16 w + 15 of every 256-wide layer (a 16-KB slab of each weight matrix, fetched once into MFMA operand
 * registers; v_mfma_f32_16x16x4_f32 = exact fp32, k-ordered), and the (B x 256) activations are handed from layer to layer INSIDE
 * the launch (sc1 stores + drained flag -> relaxed poll + sc1 loads; no kernel boundary).  Same contract as
 * rn_f_phi_fwd_from_partials / rn_f_phi_fwd_bwd_from_partials:
 *   xg_part != NULL: xg (B, 256) is an OUTPUT = the sums of the parts_per_row partial rows of every question, in order;
 *   xg_part == NULL: xg is the input;   label / loss: both (mean NLL folded in) or both NULL;
 *   bwd_ws != NULL (needs label, dxg and the (in, out) copies W1T, W2T): the backward dz chain for d loss = 1 in the same launch --
 *   bwd_ws (rn_workspace_bytes(RN_WS_F_PHI_BWD, ..)) receives the dz rows for rn_f_phi_bwd_grads, dxg (B, 256) the input gradient.
 * W1..W3: the nn.Linear (out, in) weights.  Results equal the row-split kernels' to fp32 summation order (~1e-7).
 * sync_ws: rn_workspace_bytes(RN_WS_F_PHI_SPLIT) bytes, ZEROED ONCE by the caller and then owned by these launches (an epoch word
 * advanced by every launch, so a replayed hipGraph needs no memset node).  Shapes: rn_f_phi_split_ok (B <= 64, G = F1 = F2 = 256,
 * A <= 32).  A poll that is not answered within ~1 s gives up instead of hanging; rn_f_phi_split_status(sync_ws) (synchronises the
 * stream) returns 0 when that has never happened, else 1 + the stage. */
int rn_f_phi_split_ok(int B, int G, int F1, int F2, int A);
int rn_f_phi_split(const float* xg_part, int parts_per_row, float* xg, const float* W1, const float* b1, const float* W2, const float* b2,
                   const float* W3, const float* b3, const float* W1T, const float* W2T, const float* mask, const long long* label,
                   float* f1, float* f2, float* out, float* loss, void* bwd_ws, float* dxg, void* sync_ws, int B, int G, int F1, int F2,
                   int A, void* stream);
int rn_f_phi_split_status(const void* sync_ws, void* stream);
int rn_f_phi_bwd_grads(const void* bwd_ws, const float* xg, const float* f1, const float* f2, float* dW1, float* db1, float* dW2,
                       float* db2, float* dW3, float* db3, int B, int G, int F1, int F2, int A, void* stream);
int rn_f_phi_bwd(const float* gout, const float* out, const float* f2, const float* f1, const float* xg, const float* W1,
                 const float* W2, const float* W3, const float* mask, float* dW1, float* db1, float* dW2, float* db2,
                 float* dW3, float* db3, float* dxg, void* ws, int B, int G, int F1, int F2, int A, void* stream);
/* The same with the mean negative log-likelihood of the batch (F.nll_loss(output, label), train.py:41) folded in:
 * the forward also writes loss[0] = -mean_b out[b][label[b]] (label: int64, 0 <= label < A), the backward takes
 * gloss = d L / d loss (one device float) instead of a log-prob gradient.  sync_ws: rn_workspace_bytes(RN_WS_F_PHI_NLL, B) bytes that the
 * caller zeroes ONCE; afterwards it belongs to these calls (block partials + a self re-arming completion counter). */
int rn_f_phi_fwd_nll(const float* xg, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                     const float* b3, const float* mask, const long long* label, float* f1, float* f2, float* out, float* loss,
                     void* sync_ws, int transposed, int B, int G, int F1, int F2, int A, void* stream);
int rn_f_phi_bwd_nll(const float* gloss, const long long* label, const float* out, const float* f2, const float* f1, const float* xg,
                     const float* W1, const float* W2, const float* W3, const float* mask, float* dW1, float* db1, float* dW2,
                     float* db2, float* dW3, float* db3, float* dxg, void* ws, int B, int G, int F1, int F2, int A, void* stream);

/* F.nll_loss(log_probs, label), mean reduction (train.py:41): loss[0] = -mean_b logp[b, label[b]];
 * backward: gout (B, A) = -gloss[0] / B at (b, label[b]), 0 elsewhere (the whole tensor is written).
 * label: int64 (B), values clamped to [0, A). */
int rn_nll_mean_fwd(const float* logp, const long long* label, float* loss, int B, int A, void* stream);
int rn_nll_mean_bwd(const long long* label, const float* gloss, float* gout, int B, int A, void* stream);

/* Tail of the training step (reference train.py:45-48) on the flat gradient buffer of the data-parallel bucket:
 * torch.nn.utils.clip_grad_norm_(params, max_norm) (max_norm <= 0: no clip) followed by torch.optim.Adam
 * (amsgrad=False, coupled weight_decay) in two launches.  g / m / v: flat fp32 buffers of n elements; `chunks`:
 * DEVICE array of nchunks records {float* param; long flat_off; int count; int pad} (count <= rn_clip_adam_chunk(),
 * a chunk never crosses a parameter) mapping flat ranges to the parameter tensors; step: 1-based update count;
 * grad_scale multiplies the gradient first (1 / world after a SUM all-reduce; 1 otherwise);
 * ws: rn_workspace_bytes(RN_WS_CLIP_ADAM) bytes; norm_out: optional device float receiving the (scaled) gradient norm. */
int rn_clip_adam_chunk(void);
int rn_clip_adam_step(const void* chunks, int nchunks, float* g, float* m, float* v, long n, void* ws, float grad_scale, float max_norm, float lr,
                      float beta1, float beta2, float eps, float weight_decay, int step, float* norm_out, void* stream);
/* ... with the per-step scalars in device memory, so that the two launches can be part of a captured hipGraph: hyper = 7 floats
 * {grad_scale, max_norm, lr, beta1, beta2, eps, weight_decay} (rewritten by the host only when they change), step_dev[0] = the
 * update count so far (incremented by the call; bias corrections from it, in double, as on the host). */
int rn_clip_adam_step_dev(const void* chunks, int nchunks, float* g, float* m, float* v, long n, void* ws, const float* hyper,
                          int* step_dev, float* norm_out, void* stream);

/* The batch hand-off in front of a captured step (train.py:36-40: img / qst / label of the next batch): up to 4 device-to-device
 * copies as ONE launch (three library copies are three launches, ~12 us in front of every replay at the headline shape).
 * dst[i], src[i]: 16-byte aligned, bytes[i] any size (a multiple of 16 bytes goes through 16-byte accesses, the rest bytewise). */
int rn_copy_many(void* const* dst, const void* const* src, const size_t* bytes, int n, void* stream);
/* The f_phi dropout mask (model.py:76, :158: F.dropout(p) on the (B, f_fc2) activations): mask[i] = keep ? 1 / (1 - p) : 0, keep with
 * probability 1 - p, from a counter-based hash of (seed, draw number, i).  state: two 64-bit device words {draws so far, 0}, advanced
 * by the launch itself (capturable: every replay draws the next mask; no host-side generator state, so a replayed step graph needs
 * no fill launches in front of it).  Not the reference's random stream -- no GPU port reproduces torch's CPU Philox draws. */
int rn_dropout_mask(float* mask, long n, float p, unsigned long long seed, unsigned long long* state, void* stream);

/* Question encoder (reference model.py:39-58): embedding lookup + 1-layer LSTM (E = 32 -> H = 128, gate order
 * i, f, g, o, zero initial state) as ONE launch per direction (rn_lstm.hip), fp32.
 *   fwd: idx (B, T) int64 tokens (clamped to [0, V)); emb (V, E); W_ih (4H, E), W_hh (4H, H), b_ih, b_hh (4H).
 *        Training (xs, gates, cs non-NULL): xs (T, B, E) embedded tokens, gates (T, B, 4H) activated gates,
 *        cs (T, B, H) cell states, hs (T+1, B, H) hidden states with hs[0] = 0 -- the final state is hs[T].
 *        Inference (xs == gates == cs == NULL): hs is (B, H) and receives the final hidden state only.
 *   bwd: dhn (B, H) = gradient of the final hidden state -> dgates (T, B, 4H), the gradient of the pre-activation
 *        gates; the parameter gradients are plain products of the saved matrices: dW_hh = dgates^T hs[:T],
 *        dW_ih = dgates^T xs, db_ih = db_hh = column sums, d xs = dgates W_ih.
 *   rn_embedding_bwd: demb[v] = sum of dx[t*B + b] over the positions holding token v (deterministic order). */
int rn_lstm_fwd(const long long* idx, const float* emb, const float* W_ih, const float* W_hh, const float* b_ih, const float* b_hh,
                float* xs, float* gates, float* cs, float* hs, int B, int T, int V, int E, int H, void* stream);
int rn_lstm_bwd(const float* dhn, const float* gates, const float* cs, const float* W_hh, float* dgates, int B, int T, int H,
                void* stream);
int rn_embedding_bwd(const long long* idx, const float* dx, float* demb, int B, int T, int V, int E, void* stream);
/* rn_embedding_bwd and the two bias gradients in one launch: db_ih = db_hh (may be NULL) = column sums of dgates (T*B, 4H), fixed
 * order.  demb == NULL (the embedding needs no gradient): the bias gradients alone. */
int rn_lstm_bwd_tail(const long long* idx, const float* dx, float* demb, const float* dgates, float* db_ih, float* db_hh, int B, int T,
                     int V, int E, int H, void* stream);

/* The 3x3 / stride-2 / pad-1 convolutions of ConvInputModel (reference model.py:13-20) as direct fp32 kernels (rn_conv.hip):
 * x (N, Cin, H, W), w (Cout, Cin, 3, 3) -- nn.Conv2d layout --, y (N, Cout, H/2, W/2), all contiguous; no bias (the fused
 * batch-norm block drops it).  Cout == 24, Cin in {3, 24} (input gradient: 24), H and W even.
 *   fwd:      y  = conv2d(x, w, stride 2, padding 1)
 *   bwd_data: dx = gradient of that w.r.t. x for the output gradient dy (every element of dx is written). */
int rn_conv3x3s2_fwd(const float* x, const float* w, float* y, int N, int Cin, int Cout, int H, int W, void* stream);
int rn_conv3x3s2_bwd_data(const float* dy, const float* w, float* dx, int N, int Cin, int Cout, int H, int W, void* stream);
/* Weight gradient of the same convolution (autograd of model.py:13-20): dw (24, Cin, 3, 3) fp32 =
 * sum_{n,oy,ox} dy[n][co][oy][ox] * x[n][ci][2 oy + ky - 1][2 ox + kx - 1], on the fp32 matrix pipe; ws =
 * rn_workspace_bytes(RN_WS_CONV_BWD_WEIGHT, N, Cin, H, W) bytes (per-block partials, summed in a fixed order). */
int rn_conv3x3s2_bwd_weight(const float* x, const float* dy, float* dw, void* ws, int N, int Cin, int Cout, int H, int W, void* stream);

/* BatchNorm2d + ReLU of the conv stack in front of the relation layer (reference model.py:22-35), fused into two
 * HBM passes per direction (rn_convnorm.hip).  x: (N, C, H, W) fp32 contiguous conv output computed WITHOUT the
 * conv bias (a bias in front of a batch norm shifts the batch mean by itself and drops out; conv_bias is only added
 * to the running mean, its gradient is identically zero); HW = H * W, a multiple of 4.
 *   fwd   (training): batch statistics -> mean / invstd (C) out; y = relu((x - mean) * invstd * gamma + beta);
 *         running_mean / running_var (may be NULL) updated with `momentum` (unbiased variance), *num_batches += 1.
 *   apply (evaluation): the same map with caller-prepared mean (= running_mean - conv_bias) and invstd.
 *   bwd:  dz = dy * (y > 0) (mask recomputed from x), dgamma / dbeta (C) out,
 *         dx = gamma * invstd * (dz - mean_n(dz) - xhat * mean_n(dz * xhat)).
 * ws: rn_workspace_bytes(RN_WS_BN_RELU, N, C, HW) bytes. */
int rn_bn_relu_fwd(const float* x, float* y, const float* gamma, const float* beta, const float* conv_bias,
                   float* running_mean, float* running_var, long long* num_batches, float* mean, float* invstd, void* ws,
                   float eps, float momentum, int N, int C, int HW, void* stream);
int rn_bn_relu_apply(const float* x, float* y, const float* gamma, const float* beta, const float* mean, const float* invstd,
                     int N, int C, int HW, void* stream);
/* zero_out (may be NULL): C floats set to 0 by the same launch -- the conv-bias gradient (identically zero, see above) as a tensor of
 * its own without a fill launch (a shared zero vector handed to autograd for several leaves is cloned: a memcpy node per layer). */
int rn_bn_relu_bwd(const float* dy, const float* x, float* dx, const float* gamma, const float* beta, const float* mean,
                   const float* invstd, float* dgamma, float* dbeta, float* zero_out, void* ws, int N, int C, int HW, void* stream);

/* rn_bn_relu_bwd of a block whose input needs no gradient (the first block: the image) together with the weight gradient of its
 * convolution (autograd of model.py:22-35 for that block): pass 1 as above; the weight-gradient kernel then forms the conv output
 * gradient from dy and the conv output xc (N, 24, H/2, W/2) while staging, so it is never written.  inp: the block's input
 * (N, Cin, H, W); dw (24, Cin, 3, 3); ws_bn / ws_conv: rn_workspace_bytes(RN_WS_BN_RELU, N, 24, H/2 * W/2) / rn_workspace_bytes(RN_WS_CONV_BWD_WEIGHT, ...) bytes.
 * Results equal rn_bn_relu_bwd + rn_conv3x3s2_bwd_weight. */
int rn_bn_relu_bwd_conv_wgrad(const float* dy, const float* xc, const float* inp, const float* gamma, const float* beta,
                              const float* mean, const float* invstd, float* dgamma, float* dbeta, float* zero_out, float* dw,
                              void* ws_bn, void* ws_conv, int N, int Cin, int H, int W, void* stream);

/* THE FORWARD CHAIN (headline path; model.py:108-152): tables of the factored first layer + one launch for the four g layers and
 * the pair sum.  The pair matrix [x_j | x_i | q] of model.py:117-127 never exists:
 *   W0 [x_j | x_i | q] + b0 = W0a x_j + (W0b x_i + W0c q + b0)
 * rn_pair_tables: Xp (B*n [+ 1], 64) = x[b, j, 0:k] zero padded, in xp_dtype (RN_F16 for the chain);  Vc (B*n, N) fp32 = b0 +
 *   W0b x[b, i] + W0c q[b]   (W0T = W0 transposed, (2k+Q, N) fp32; x (B, n, k) with element strides; k <= 32; Q = 0: no question
 *   term).  coord != NULL: x supplies only the first kf columns of an object (the conv grid), the rest are the coordinate tags
 *   coord (k - kf, n) of model.py:195-201 -- the concatenated object tensor never exists either.
 * rn_g_chain_fwd_rr_f16s_alg0: the register-resident chain on those tables, "f16s" arithmetic -- fp16 operand registers (activations
 *   saturate at 65504), fp32 accumulate:
 *   layer 0:     a K = 64 product on the object rows with the Vc row of (question, i) as its bias; two MFMA passes, against Whi[0] =
 *                fp16(W0[:, 0:k]) and Wlo[0] = fp16(W0[:, 0:k] - hi) (rn_pack_matrix_frag_many modes 4 | 1 and 8 | 1);
 *   layers 1..3: ONE pass on TILE-DITHERED hi images: Whi[l] holds `dither` (1, 2, 4, 8; the module uses 4) images, 65536 fp16 apart,
 *                image d = fp16(W + ((d + 1/2) / dither - 1/2) ulp_fp16(W)) (mode 4 | dither << 8), and 256-row tile t multiplies
 *                image t mod dither: the weight rounding error no longer has the same sign for all pairs of a question and averages
 *                out in the pair sum (model.py:151-152) like the activation rounding does.  Wlo[1..3] are not read (may be NULL).
 *                Error against the fp32 reference on the released checkpoints: 1.7e-4 / 1.9e-4 (original-fp / ir-fp; one plain
 *                pass: 2.6e-3; the bar is 1e-3).  WHICH image a pair row multiplies depends on its tile, i.e. on where the
 *                question sits in the batch (n*n / 256 not a multiple of `dither`) and on the order of its objects: results are
 *                bitwise reproducible for one batch, and equal across batch compositions only to the mode's accuracy (~1e-4).
 *   dither == 0: the TWO-PASS arithmetic (inference only: H and mask must be NULL) -- hi + lo split images on EVERY layer, Whi[l] =
 *                fp16(W_l), Wlo[l] = fp16(W_l - hi) (modes 4 / 8).  Nothing depends on the tile index: a question's outputs are
 *                the same wherever it sits in the batch and for any order of its objects, up to fp32 summation order (~1e-6).
 *                What the module's eval() runs (options.eval_two_pass).  Twice the MFMAs of layers 1..3.
 *   inject_layer = 0: the question is part of the tables (Q > 0 in rn_pair_tables).  inject_layer = 2 (the "IR" variants,
 *     model.py:131-142; tables built with Q = 0): layer 2's input is [H_1 | q[b]], i.e. W_2 [H_1 | q] + b_2 = W_2[:, 0:256] H_1 + Vq[b]
 *     with Vq (B, 256) fp32 = W_2[:, 256:] q[b] + b_2 prepared by the caller (one small rn_gemm_f32); Whi[2] holds W_2[:, 0:256]
 *     only, bias[2] is ignored.  Needs n divisible by 32 and n*n by 256.
 *   H / mask: both NULL (inference) or H[0..2] (H[3] NULL) + the four masks (training).  h_dtype: the type of the stored H_0..2 rows
 *     -- RN_BF16 (M x 256 bf16) or RN_FP8 (M x 256 e4m3 bytes, value = byte value), both as ROW-BLOCKED images (the only reader is
 *     rn_g_wgrad_blocked).  Ignored when H is NULL.
 *   xg_part (M/256, 256) fp32: ONE partial pair-sum row per 256-row tile (the tile's eight waves add their rows on chip) -- reduce
 *     with rn_pair_sum_fwd(xg_part, 256, xg, ws, RN_F32, B, n*n/256, 256) or let rn_f_phi_fwd_from_partials add them up.
 *   njp: pair rows per (question, i) group.  njp == n: the pair rows are the n*n pairs of model.py:117-127 (n a multiple of 32).
 *     njp = 32 ceil(n / 32) > n (n a multiple of 4; e.g. the 14 x 14 grid, n = 196 -> 224): the j axis is PADDED -- pair row m =
 *     (b, i, j) with j = m mod njp, M = B * n * njp; rows with j >= n are invalid (they multiply the all-zero object row
 *     Xp16[B * n], which the caller provides; their mask bits are cleared in every layer and they are left out of the pair sums),
 *     H / mask cover the M padded rows, and xg_part holds TWO partial rows per 256-row tile, (M / 256 * 2, 256) -- reduce with
 *     rn_pair_sum_tiles.  The backward side (rn_g_chain_bwd_rr with rows_per_question = n * njp, rn_g_wgrad_blocked,
 *     rn_pair_reduce_bwd with njp) then sees zero gradients for the invalid rows without knowing about the padding.  Question at
 *     layer 0 only.
 *   gate_in_h2 (with the e4m3 training output set only): != 0 -- the last layer's ReLU gate (= mask[3]) is also written into the
 *     sign bits of the H[2] image: byte (m, f) = e4m3(H_2[m, f]) | gate_3[m, f] << 7 (exactly what rn_relu_gate_image of
 *     include/rn_hip_debug.h merges into a plain image).  That one image is both operands of the last layer's gate job in
 *     rn_g_wgrad_blocked; every other reader masks bit 7 off (rn_fp8_copy_health does). */
int rn_pair_tables(const float* x, long sxb, long sxn, long sxk, const float* coord, int kf, const float* q, long ldq,
                   const float* W0T, const float* b0, void* Xp, int xp_dtype /* RN_BF16 | RN_F16 */, float* Vc, int B, int n, int k,
                   int Q, int N, void* stream);
int rn_g_chain_fwd_rr_f16s_alg0(const void* Xp16, const float* Vc, int n, int njp, const void* const* Whi, const void* const* Wlo, int dither,
                                const float* const* bias, void* const* H, int h_dtype, void* const* mask, int gate_in_h2, float* xg_part,
                                const float* Vq, int inject_layer, int M, int L, int G, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RN_HIP_H */
