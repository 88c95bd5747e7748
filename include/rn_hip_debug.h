/* Diagnostic entry points of librn_hip.so -- NOT part of the product ABI (include/rn_hip.h).
 * Used by tests/ and tools/ only; nothing under relationnetworks-clevr_amd/ on the training / inference path calls them. */
#ifndef RN_HIP_DEBUG_H
#define RN_HIP_DEBUG_H
#ifdef __cplusplus
extern "C" {
#endif

/* Raw lane mapping of ds_read_b64_tr_b16 (tests/test_gpu_kernels.py pins the layout the wgrad kernel relies on). */
int rn_probe_tr16(const unsigned short* in4096, unsigned short* out256, void* stream);

/* A one-thread kernel that stores the constant-rate wall clock (wall_clock64) into *slot, in stream order.
 * Captured between the kernels of the step's hipGraph it yields a concurrent multi-stream timeline (tools/step_timeline.py). */
int rn_debug_stamp(unsigned long long* slot, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RN_HIP_DEBUG_H */
