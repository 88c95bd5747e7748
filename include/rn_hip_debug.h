/* Diagnostic entry points of librn_hip.so -- NOT part of the product ABI (include/rn_hip.h).
 * Used by tests/, tools/ and bench.py only; nothing under relationnetworks-clevr_amd/ on the training / inference path calls them. */
#ifndef RN_HIP_DEBUG_H
#define RN_HIP_DEBUG_H
#ifdef __cplusplus
extern "C" {
#endif

/* Raw lane mapping of ds_read_b64_tr_b16 (tests/test_gpu_kernels.py pins the layout the wgrad kernel relies on). */
int rn_probe_tr16(const unsigned short* in4096, unsigned short* out256, void* stream);

/* ... and of ds_read_b64_tr_b8 (the e4m3 operand of the streaming wgrad): lane l supplies &lds[8 l] of a linear 4096-byte image. */
int rn_probe_tr8(const unsigned char* in4096, unsigned char* out512, void* stream);

/* The e4m3 conversions of the kernels at a power-of-two `scale`: out8_bf16 / out8_f16 (n bytes each) = the down-conversion of
 * bf16(in) / fp16(in) (v_cvt_scalef32_pk_fp8_bf16 / _f16), back (n floats) = the up-conversion of out8_bf16
 * (v_cvt_scalef32_pk_bf16_fp8).  n % 4 == 0. */
int rn_probe_fp8_cvt(const float* in, float scale, void* out8_bf16, void* out8_f16, float* back, int n, void* stream);

/* A one-thread kernel that stores the constant-rate wall clock (wall_clock64) into *slot, in stream order.
 * Captured between the kernels of the step's hipGraph it yields a concurrent multi-stream timeline (tools/step_timeline.py). */
int rn_debug_stamp(unsigned long long* slot, void* stream);

/* Row-major (M, 256) matrix <-> row-blocked image (include/rn_hip.h, "ROW-BLOCKED operand images"; back != 0: the other way).
 * The chains write the images themselves; tests and tools convert with this.  M % 16 == 0, dtype RN_BF16 or RN_FP8. */
int rn_rows_to_blocked(const void* src, void* dst, int dtype, int M, int back, void* stream);

/* Merges the last layer's lane masks into the sign bits of the e4m3 row-blocked image img (M x 256 bytes, in place): byte (m, f) =
 * (byte & 0x7f) | gate[m, f] << 7 -- what the f16s forward chain writes for H_2 with gate_in_h2 != 0; the tests build the reference
 * image with this.  M % 32 == 0. */
int rn_relu_gate_image(const void* mask, void* img, int M, void* stream);

/* The work items of rn_g_chain_bwd_rr_red for a shape, as the kernel decodes them (host code, no launch): out (items, 3) ints =
 * {first tile, tiles, Rj record}; returns the number of items (< 0: argument error).  tests/test_host_cpu.py checks on the CPU that
 * every tile is run exactly once and that every record rn_pair_reduce_parts reads is written exactly once. */
int rn_probe_red_schedule(int M, int n, int njp, int tiles_per_unit, int units_whole, int* out, int max_items);

/* The matrix pipe's SUSTAINED rate on this device: `workgroups` workgroups of 256 * waves_per_simd threads, every wave issuing
 * iters x 16 v_mfma_f32_32x32x16 (dtype RN_F16 | RN_BF16) on two alternating accumulators and nothing else.  The caller times the
 * launch: flops = workgroups * 4 * waves_per_simd * iters * 16 * 32768.  bench.py quotes the chains against this measured rate
 * beside the nominal 2.5 PFLOP/s (the chip clocks to its power budget).  out: workgroups * 256 * waves_per_simd floats (a sink). */
int rn_probe_mfma_stream(float* out, int workgroups, int waves_per_simd, int iters, int dtype, void* stream);
/* Row splits of an rn_g_wgrad_blocked launch (aligned == 0, e4m3 images) with `nw` stored-gradient jobs (WIDE units: one workgroup per
 * split) and `nq` gate jobs (four workgroups per split): nw * Zw + 4 * nq * Zq <= 4 x the library's row-split budget (160 workgroups),
 * Zw ~ 3 Zq so that both kinds of workgroup finish together; never more splits than 64-row steps.  Host arithmetic only. */
int rn_debug_wgrad_blocked_mix(int M, int nw, int nq, int* Zw, int* Zq);

/* ... the same stream on all-zero operands (zero_operands != 0): the data pattern "peak" figures are usually measured with */
int rn_probe_mfma_stream_ops(float* out, int workgroups, int waves_per_simd, int iters, int dtype, int zero_operands, void* stream);

/* Which f_phi kernels rn_f_phi_fwd / _fwd_nll / _bwd / _bwd_nll launch: 0 = by size (the per-layer, feature-split launches for
 * anything wider than the 256-wide image models: config.json's *-sd), 1 = always the per-layer launches, -1 = never.  Returns the
 * previous mode.  tests/ compare the two paths bit for bit with it; the product never calls it. */
int rn_debug_f_phi_wide(int mode);

/* Moves the tile switch of the per-layer GEMMs (rn_g_linear_fwd / _bwd_dgrad): 64 x 64 workgroup tiles where the 128 x 256 ones
 * would be fewer than n workgroups (n < 0: the built-in value).  Returns the previous value.  tools/dbg/time_gemm_tiles.py sweeps
 * with it; the results do not depend on the tile (bit for bit). */
int rn_debug_gemm_small_below(int n);

#ifdef __cplusplus
}
#endif
#endif /* RN_HIP_DEBUG_H */
