/* xpoint_hip.h — C ABI of libxpoint_hip.so: the MI355X (gfx950) XPoint inference hot path.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types cross this boundary.
 *   - every function returns int: 0 = ok, <0 = error (text: xp_last_error(), thread-local).
 *   - the CALLER owns every buffer; pointers are device pointers (hipMalloc'd or torch
 *     `tensor.data_ptr()`), contiguous float32 unless stated, 16-byte aligned.
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream) and
 *     are stream-ordered; no function synchronises the device unless it says so.
 *   - no hidden allocation except inside xp_ctx_create / xp_ctx_destroy.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the reference
 * repo canyagmur/XPoint).
 */
#ifndef XPOINT_HIP_H
#define XPOINT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int xp_version(void);
const char* xp_last_error(void);
int xp_device_info(int device, int* cu_count, int* wave_size, char* arch, int arch_len);

/* Selective-scan forward.  Replaces the pybind op `selective_scan_cuda_oflex.fwd(u, delta, A, B, C,
 * D, delta_bias, delta_softplus, nrows, out_float)` —
 * xpoint/models/vmamba_src/kernels/selective_scan/csrc/selective_scan/cusoflex/selective_scan_oflex.cpp:143-231,
 * kernel selective_scan_fwd_kernel_oflex.cuh:67-181; Python caller vmamba_src/csms6s.py:71-87,112-126.
 *   u (batch, dim, seqlen); delta (batch, delta_dim, seqlen), dim % delta_dim == 0;
 *   A (dim, dstate); B, C (batch, ngroups, dstate, seqlen), dim % ngroups == 0;
 *   D (dim) or NULL; delta_bias (delta_dim) or NULL; out (batch, dim, seqlen) float32 ("oflex");
 *   last_state (batch, dim, dstate) or NULL  (= the reference's x[:, :, -1, 1::2]).
 *   dstate <= 256 (reference MAX_DSTATE, selective_scan_oflex.cpp:11). */
int xp_selective_scan_fwd(const float* u, const float* delta, const float* A, const float* B, const float* C,
                          const float* D, const float* delta_bias, float* out, float* last_state,
                          int batch, int dim, int delta_dim, int seqlen, int dstate, int ngroups,
                          int delta_softplus, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* XPOINT_HIP_H */
