// Shared helpers for the XPoint gfx950 kernels (internal; the public C ABI is include/xpoint_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#define XP_OK 0
#define XP_ERR_ARG (-1)
#define XP_ERR_HIP (-2)
#define XP_ERR_STATE (-3)

void xp_set_error(const char* fmt, ...);

#define XP_CHECK_ARG(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            xp_set_error(__VA_ARGS__);          \
            return XP_ERR_ARG;                  \
        }                                       \
    } while (0)

#define XP_HIP(call)                                                                       \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess) {                                                           \
            xp_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return XP_ERR_HIP;                                                             \
        }                                                                                  \
    } while (0)

#define XP_LAUNCH_CHECK()                                                                  \
    do {                                                                                   \
        hipError_t e__ = hipGetLastError();                                                \
        if (e__ != hipSuccess) {                                                           \
            xp_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
            return XP_ERR_HIP;                                                             \
        }                                                                                  \
    } while (0)

// Optional HIP-event timing of one launch (see api.cpp); a no-op unless xp_prof_enable(1).
// flops / bytes are the ALGORITHMIC work of the launch (roofline numerators).
class XpProfScope {
public:
    XpProfScope(const char* tag, hipStream_t s, double flops, double bytes);
    ~XpProfScope();
private:
    bool active_;
    hipStream_t stream_;
    size_t index_ = 0;
};

static inline int xp_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

#ifdef __HIPCC__
// softplus with torch semantics (beta 1, threshold 20): reference csms6s.py:49-50 /
// selective_scan_fwd_kernel_oflex.cuh:124-127.
__device__ __forceinline__ float xp_softplus(float x) { return x <= 20.f ? log1pf(expf(x)) : x; }
__device__ __forceinline__ float xp_silu(float x) { return x / (1.f + expf(-x)); }
__device__ __forceinline__ float xp_gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float xp_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float xp_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
#endif
