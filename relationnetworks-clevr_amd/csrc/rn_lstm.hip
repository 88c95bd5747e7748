// Question encoder of the relation network (reference model.py:39-58): Embedding(V, 32) -> 1-layer LSTM(32 -> 128,
// batch_first), final hidden state.  The stock path runs the recurrence as 2 launches per time step (a GEMM and a
// point-wise cell), 80 dependent launches for 20 tokens forward + backward -- 0.35 ms of launch latency on the
// critical path of a 1.6 ms training step for 0.1 GFLOP of arithmetic.  Here the whole recurrence is ONE launch per
// direction: a workgroup owns two batch rows for all time steps, every thread keeps its slice of the recurrent
// weights in registers, the hidden / cell state lives in LDS.
//   forward : thread = gate column (512): 32 + 128 weights in registers; gates = W_ih x_t + W_hh h_{t-1} + b
//   backward: thread = (hidden unit k, quarter q of the gate columns): its 128 entries of W_hh^T in registers;
//             emits dgates (T, B, 512); the weight gradients are three small GEMMs over the saved (T*B)-row matrices
//             outside (they are off the recurrence), the embedding gradient a deterministic per-row gather-add.
// fp32 throughout (expf / tanhf), gate order i, f, g, o as in torch.nn.LSTM.
#include "rn_common.h"

namespace {
constexpr int LS_H = 128, LS_E = 32, LS_G = 4 * LS_H, LS_RB = 2, LS_NT = 512;
__device__ __forceinline__ float ls_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
}  // namespace

// xs (T, B, E): embedded tokens; gates (T, B, 4H): activated gates; cs (T, B, H); hs (T+1, B, H) with hs[0] = 0
__global__ __launch_bounds__(LS_NT) void lstm_fwd_kernel(const long long* __restrict__ idx, const float* __restrict__ emb,
                                                         const float* __restrict__ W_ih, const float* __restrict__ W_hh,
                                                         const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                         float* __restrict__ xs, float* __restrict__ gates, float* __restrict__ cs,
                                                         float* __restrict__ hs, int B, int T, int V, int save) {
  __shared__ __attribute__((aligned(16))) float x_s[LS_RB][LS_E], h_s[LS_RB][LS_H], g_s[LS_RB][LS_G];
  const int col = threadIdx.x, b0 = blockIdx.x * LS_RB;
  float wih[LS_E], whh[LS_H];
#pragma unroll
  for (int k = 0; k < LS_E; k += 4) {
    const f32x4 w = *reinterpret_cast<const f32x4*>(W_ih + (long)col * LS_E + k);
    wih[k] = w[0]; wih[k + 1] = w[1]; wih[k + 2] = w[2]; wih[k + 3] = w[3];
  }
#pragma unroll
  for (int k = 0; k < LS_H; k += 4) {
    const f32x4 w = *reinterpret_cast<const f32x4*>(W_hh + (long)col * LS_H + k);
    whh[k] = w[0]; whh[k + 1] = w[1]; whh[k + 2] = w[2]; whh[k + 3] = w[3];
  }
  const float bias = b_ih[col] + b_hh[col];
  const int sr = col >> 7, sj = col & 127;                 // state thread: (row, hidden unit) for col < 256
  float c_reg = 0.f;
  if (col < LS_RB * LS_H) {
    h_s[sr][sj] = 0.f;
    if (b0 + sr < B) hs[(long)(b0 + sr) * LS_H + sj] = 0.f;   // hs[0]
  }
  for (int t = 0; t < T; ++t) {
    if (col < LS_RB * LS_E) {
      const int r = col >> 5, k = col & 31;
      float v = 0.f;
      if (b0 + r < B) {
        long long tok = idx[(long)(b0 + r) * T + t];
        tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
        v = emb[tok * LS_E + k];
        if (save) xs[((long)t * B + b0 + r) * LS_E + k] = v;
      }
      x_s[r][k] = v;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < LS_RB; ++r) {
      float a0 = bias, a1 = 0.f;
#pragma unroll
      for (int k = 0; k < LS_E; k += 4) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(&x_s[r][k]);
        a0 = fmaf(wih[k], x[0], a0); a1 = fmaf(wih[k + 1], x[1], a1); a0 = fmaf(wih[k + 2], x[2], a0); a1 = fmaf(wih[k + 3], x[3], a1);
      }
#pragma unroll
      for (int k = 0; k < LS_H; k += 4) {
        const f32x4 h = *reinterpret_cast<const f32x4*>(&h_s[r][k]);
        a0 = fmaf(whh[k], h[0], a0); a1 = fmaf(whh[k + 1], h[1], a1); a0 = fmaf(whh[k + 2], h[2], a0); a1 = fmaf(whh[k + 3], h[3], a1);
      }
      g_s[r][col] = a0 + a1;
    }
    __syncthreads();
    if (col < LS_RB * LS_H) {
      const float gi = ls_sigmoid(g_s[sr][sj]), gf = ls_sigmoid(g_s[sr][LS_H + sj]);
      const float gg = tanhf(g_s[sr][2 * LS_H + sj]), go = ls_sigmoid(g_s[sr][3 * LS_H + sj]);
      c_reg = fmaf(gf, c_reg, gi * gg);
      const float h = go * tanhf(c_reg);
      h_s[sr][sj] = h;
      if (b0 + sr < B) {
        const long row = (long)t * B + b0 + sr;
        if (save) {
          float* gp = gates + row * LS_G + sj;
          gp[0] = gi; gp[LS_H] = gf; gp[2 * LS_H] = gg; gp[3 * LS_H] = go;
          cs[row * LS_H + sj] = c_reg;
        }
        if (save || t == T - 1) hs[(save ? (row + B) : (long)(b0 + sr)) * LS_H + sj] = h;   // hs[t + 1] (or just h_n)
      }
    }
    __syncthreads();
  }
}

// dgates (T, B, 4H): gradient of the pre-activation gates; dhn (B, H): gradient of the final hidden state
__global__ __launch_bounds__(LS_NT) void lstm_bwd_kernel(const float* __restrict__ dhn, const float* __restrict__ gates,
                                                         const float* __restrict__ cs, const float* __restrict__ W_hh,
                                                         float* __restrict__ dgates, int B, int T) {
  __shared__ __attribute__((aligned(16))) float dg_s[LS_RB][LS_G], part[4][LS_RB][LS_H];
  const int tid = threadIdx.x, b0 = blockIdx.x * LS_RB;
  const int k = tid & 127, q = tid >> 7;
  float wt[LS_H];                                          // W_hh[128 q + c][k], c = 0..127
#pragma unroll
  for (int c = 0; c < LS_H; ++c) wt[c] = W_hh[(long)(LS_H * q + c) * LS_H + k];
  const int sr = tid >> 7, sj = tid & 127;                 // state thread (tid < 256): (row, hidden unit)
  const bool st = tid < LS_RB * LS_H, live = st && (b0 + sr < B);
  float dh = live ? dhn[(long)(b0 + sr) * LS_H + sj] : 0.f, dc = 0.f;
  for (int t = T - 1; t >= 0; --t) {
    if (st) {
      float di = 0.f, df = 0.f, dgg = 0.f, dob = 0.f;
      if (live) {
        const long row = (long)t * B + b0 + sr;
        const float* gp = gates + row * LS_G + sj;
        const float gi = gp[0], gf = gp[LS_H], gg = gp[2 * LS_H], go = gp[3 * LS_H];
        const float c = cs[row * LS_H + sj], cprev = t > 0 ? cs[(row - B) * LS_H + sj] : 0.f;
        const float tc = tanhf(c);
        dob = dh * tc * go * (1.f - go);
        const float dct = fmaf(dh * go, 1.f - tc * tc, dc);
        di = dct * gg * gi * (1.f - gi);
        df = dct * cprev * gf * (1.f - gf);
        dgg = dct * gi * (1.f - gg * gg);
        dc = dct * gf;
        float* op = dgates + row * LS_G + sj;
        op[0] = di; op[LS_H] = df; op[2 * LS_H] = dgg; op[3 * LS_H] = dob;
      }
      dg_s[sr][sj] = di; dg_s[sr][LS_H + sj] = df; dg_s[sr][2 * LS_H + sj] = dgg; dg_s[sr][3 * LS_H + sj] = dob;
    }
    __syncthreads();
    // dh_{t-1}[r][k] = sum_col dgates[r][col] W_hh[col][k]: this thread's quarter of the columns
#pragma unroll
    for (int r = 0; r < LS_RB; ++r) {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int c = 0; c < LS_H; c += 4) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(&dg_s[r][LS_H * q + c]);
        a0 = fmaf(wt[c], d[0], a0); a1 = fmaf(wt[c + 1], d[1], a1); a0 = fmaf(wt[c + 2], d[2], a0); a1 = fmaf(wt[c + 3], d[3], a1);
      }
      part[q][r][k] = a0 + a1;
    }
    __syncthreads();
    if (st) dh = (part[0][sr][sj] + part[1][sr][sj]) + (part[2][sr][sj] + part[3][sr][sj]);
    __syncthreads();
  }
}

// demb[v][:] = sum over the positions p = t*B + b with token v of dx[p][:] -- block = vocabulary row.
// Scanning the tokens straight from memory is B T / 8 DEPENDENT loads per thread (344 at the headline shape: 38 us for 350 KB
// of data).  Here every thread tests B T / 256 tokens (loaded back to back), each wave compacts its matches into a list of its
// own by ballot + prefix count (order inside the list: round, lane), and the rows of the lists are added in a fixed order:
// segment s takes entries s, s + 8, .. of list 0, then of list 1, ..; the 8 segments are combined in order.  Deterministic.
// Blocks [V, V + LS_G / 32) (when dgates is given) add up the columns of dgates (T B, 4H) -> db_ih = db_hh: the bias gradients
// ride in the same launch instead of a library reduction and a copy (32 columns x 32 row phases per block, 16-byte loads).
namespace {
constexpr int EB_TOK = 8192;                               // tokens a block can list (beyond: the plain scan)
}
__global__ __launch_bounds__(256) void emb_bwd_kernel(const long long* __restrict__ idx, const float* __restrict__ dx,
                                                      float* __restrict__ demb, int B, int T, int V, const float* __restrict__ dgates,
                                                      float* __restrict__ db_ih, float* __restrict__ db_hh) {
  __shared__ int list_s[4][EB_TOK / 4];
  __shared__ int cnt_s[4];
  __shared__ f32x4 red[32][8];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= V) {                              // ---- column sums
    const int c4 = tid & 7, ph = tid >> 3;
    const long R = (long)T * B;
    const f32x4* src = reinterpret_cast<const f32x4*>(dgates) + ((int)blockIdx.x - V) * 8 + c4;
    f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    long r = ph;
#pragma unroll 4
    for (; r + 96 < R; r += 128) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] += src[(r + 32 * q) * (LS_G / 4)];
    }
    for (; r < R; r += 32) acc[0] += src[r * (LS_G / 4)];
    red[ph][c4] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (tid < 8) {
      f32x4 sum = red[0][tid];
      for (int i = 1; i < 32; ++i) sum += red[i][tid];
      const int c = ((int)blockIdx.x - V) * 32 + 4 * tid;
#pragma unroll
      for (int e = 0; e < 4; ++e) {                        // (the outputs may be slices of a flat gradient buffer: 4-byte aligned)
        db_ih[c + e] = sum[e];
        if (db_hh) db_hh[c + e] = sum[e];
      }
    }
    return;
  }
  const int v = blockIdx.x, k = tid & 31, seg = tid >> 5, lane = tid & 63, w = tid >> 6;
  const int n = B * T;
  float a = 0.f;
  if (n > EB_TOK) {                                        // (straight from memory)
    for (int b = seg; b < B; b += 8)
      for (int t = 0; t < T; ++t) {
        long long tok = idx[(long)b * T + t];
        tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
        if (tok == v) a += dx[((long)t * B + b) * LS_E + k];
      }
  } else {
    const int R = (n + 255) >> 8;                          // <= 32 rounds
    unsigned hit = 0;
    for (int r = 0; r < R; ++r) {
      const int i = (r << 8) + tid;
      long long tok = i < n ? idx[i] : -1;
      if (i < n) tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
      hit |= (tok == v ? 1u : 0u) << r;
    }
    int off = 0;                                           // (wave-uniform)
    for (int r = 0; r < R; ++r) {
      const bool m = (hit >> r) & 1u;
      const unsigned long long bal = __ballot(m);
      if (m) {
        const int i = (r << 8) + tid, b = i / T, t = i - b * T;
        list_s[w][off + __popcll(bal & ((1ull << lane) - 1ull))] = t * B + b;
      }
      off += __popcll(bal);
    }
    if (lane == 0) cnt_s[w] = off;
    __syncthreads();
    for (int q = 0; q < 4; ++q) {
      const int c = cnt_s[q];
      for (int e = seg; e < c; e += 8) a += dx[(long)list_s[q][e] * LS_E + k];
    }
  }
  float* redf = reinterpret_cast<float*>(red);            // [8][32]
  redf[seg * LS_E + k] = a;
  __syncthreads();
  if (seg == 0) {
    float sum = 0.f;
    for (int i = 0; i < 8; ++i) sum += redf[i * LS_E + k];
    demb[(long)v * LS_E + k] = sum;
  }
}

extern "C" int rn_lstm_fwd(const long long* idx, const float* emb, const float* W_ih, const float* W_hh, const float* b_ih,
                           const float* b_hh, float* xs, float* gates, float* cs, float* hs, int B, int T, int V, int E, int Hh,
                           void* stream) {
  RN_CHECK_ARG(idx && emb && W_ih && W_hh && b_ih && b_hh && hs && B > 0 && T > 0 && V > 0, "rn_lstm_fwd: bad pointer/size");
  RN_CHECK_ARG(E == LS_E && Hh == LS_H, "rn_lstm_fwd: built for embedding %d / hidden %d (got %d / %d)", LS_E, LS_H, E, Hh);
  RN_CHECK_ARG(((uintptr_t)W_ih | (uintptr_t)W_hh) % 16 == 0, "rn_lstm_fwd: weights must be 16-byte aligned");
  const int save = (xs && gates && cs) ? 1 : 0;
  lstm_fwd_kernel<<<cdiv(B, LS_RB), LS_NT, 0, (hipStream_t)stream>>>(idx, emb, W_ih, W_hh, b_ih, b_hh, xs, gates, cs, hs, B, T, V, save);
  RN_LAUNCH_CHECK("rn_lstm_fwd");
  return 0;
}

extern "C" int rn_lstm_bwd(const float* dhn, const float* gates, const float* cs, const float* W_hh, float* dgates, int B, int T,
                           int Hh, void* stream) {
  RN_CHECK_ARG(dhn && gates && cs && W_hh && dgates && B > 0 && T > 0, "rn_lstm_bwd: bad pointer/size");
  RN_CHECK_ARG(Hh == LS_H, "rn_lstm_bwd: built for hidden %d (got %d)", LS_H, Hh);
  lstm_bwd_kernel<<<cdiv(B, LS_RB), LS_NT, 0, (hipStream_t)stream>>>(dhn, gates, cs, W_hh, dgates, B, T);
  RN_LAUNCH_CHECK("rn_lstm_bwd");
  return 0;
}

extern "C" int rn_embedding_bwd(const long long* idx, const float* dx, float* demb, int B, int T, int V, int E, void* stream) {
  RN_CHECK_ARG(idx && dx && demb && B > 0 && T > 0 && V > 0 && E == LS_E, "rn_embedding_bwd: bad argument (embedding width must be %d)", LS_E);
  emb_bwd_kernel<<<V, 256, 0, (hipStream_t)stream>>>(idx, dx, demb, B, T, V, nullptr, nullptr, nullptr);
  RN_LAUNCH_CHECK("rn_embedding_bwd");
  return 0;
}

extern "C" int rn_lstm_bwd_tail(const long long* idx, const float* dx, float* demb, const float* dgates, float* db_ih, float* db_hh,
                                int B, int T, int V, int E, int Hh, void* stream) {
  RN_CHECK_ARG(dgates && (uintptr_t)dgates % 16 == 0 && db_ih && B > 0 && T > 0 && Hh == LS_H, "rn_lstm_bwd_tail: bad argument (hidden width must be %d)", LS_H);
  RN_CHECK_ARG(!demb || (idx && dx && V > 0 && E == LS_E), "rn_lstm_bwd_tail: the embedding gradient needs idx, dx, V > 0 and width %d", LS_E);
  const int Vb = demb ? V : 0;
  emb_bwd_kernel<<<Vb + LS_G / 32, 256, 0, (hipStream_t)stream>>>(idx, dx, demb, B, T, Vb, dgates, db_ih, db_hh);
  RN_LAUNCH_CHECK("rn_lstm_bwd_tail");
  return 0;
}
