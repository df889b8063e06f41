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

// the same check inside a void launch helper: the message is recorded, the launch that follows fails and XP_LAUNCH_CHECK at the caller returns the error
#define XP_HIP_WARN(call)                                                                  \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess) xp_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
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

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property of a kernel: one "done" flag per device ordinal (a process that drives several GPUs
// would otherwise launch with > 64 KB of dynamic LDS on the second device without the opt-in).  `static XpPerDeviceOnce once; if (once.need()) XP_HIP(...)`.
struct XpPerDeviceOnce {
    bool done[64] = {};
    bool need() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;      // unknown device: set the attribute every time
        if (done[d]) return false;
        done[d] = true;
        return true;
    }
};

static inline int xp_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Sticky status bits of xp_xpoint_forward_ex (include/xpoint_hip.h): OR-ed into the caller's device word by the kernels that produce
// the encoder output, the heat map and the descriptor volume.
#define XP_STATUS_ENC 1     /* encoder output: non-finite, or beyond the dense engine's operand range */
#define XP_STATUS_PROB 2    /* heat map: a non-finite logit */
#define XP_STATUS_DESC 4    /* descriptor volume: a non-finite element */
int xp_depth_to_space_nhwc_st(const float* x, float* y, int batch, int H, int W, int C, int bs, float limit, int* status, void* stream);
int xp_softmax_shuffle_st(const float* logits, float* prob, int batch, int Hc, int Wc, int r, int ld, int mode, int* status, void* stream);
int xp_l2norm_rows_st(const float* x, float* y, int64_t rows, int C, float eps, int* status, void* stream);

// Partial products per multiply of the split-bf16 dense kernels (xp_set_dense_products): 6 (default), 3 or 1.
int xp_dense_products_value();
// Dense-layer engine of the fused encoder (xp_set_dense_engine): 0 = x3 (split bf16), 1 = h2 (split fp16, three products).
int xp_dense_engine_value();
// Per-launch override of the split-fp16 engine inside xp_xpoint_forward (xp_set_dense_override): bit i = dense launch i runs on the split-bf16 planes.
unsigned long long xp_dense_override_value();
// Mixed-precision class (xp_set_amp_mode): 1 = every operation autocast would end in a half tensor rounds its output to fp16.
int xp_amp_value();

#ifdef __HIPCC__
// softplus with torch semantics (beta 1, threshold 20): reference csms6s.py:49-50 /
// selective_scan_fwd_kernel_oflex.cuh:124-127.
__device__ __forceinline__ float xp_softplus(float x) { return x <= 20.f ? log1pf(expf(x)) : x; }
// exp(x) on the hardware exp2 unit with the product x*log2(e) carried in two floats, so the relative error stays
// ~2 ulp for any |x| (a plain exp2f(x*log2e) loses |x| * 6e-8): p = 2^t, t = fl(x*L), r = x*L - t (exact via fma).
__device__ __forceinline__ float xp_exp_fast(float x) {
    const float L_hi = 1.44269502162933349609375f, L_lo = 1.925963033500011e-08f;   // log2(e) = hi + lo
    const float t = x * L_hi;
    float r = fmaf(x, L_hi, -t);
    r = fmaf(x, L_lo, r);
    const float p = __builtin_amdgcn_exp2f(t);
    return fmaf(p, r * 0.693147180559945309f, p);
}
// log1p(e) for e >= 0: log(u) with u = fl(1 + e), minus the first-order correction for the rounding of 1 + e
// (Kahan).  v_log_f32 is accurate to ~1 ulp of its result, so the relative error is ~2e-7 down to e -> 0
// (u == 1 gives exactly e).
__device__ __forceinline__ float xp_log1p_fast(float e) {
    const float u = 1.f + e;
    const float l = __builtin_amdgcn_logf(u) * 0.693147180559945309f;
    const float c = ((u - 1.f) - e) * __builtin_amdgcn_rcpf(u);
    return l - c;
}
// One scan step's delta = softplus(x) (torch semantics: beta 1, threshold 20; reference csms6s.py:49-50) and a = exp(delta * A),
// shared by the fused SS2D core and the operator-boundary selective scan.
//   e = e^x straight on the exp2 unit (x <= 20 wherever e is used, so the argument scaling costs at most |x| * 6e-8 relative);
//   e <= 0.14 (the usual case: dt_projs_bias is initialised to softplus^-1 of [1e-3, 0.1], VMamba.py:196-211): ln(1 + e) by its
//     alternating series to e^8 (truncation e^9 / 9 <= 2.3e-9) and a = 2^(A log2(e) delta): two transcendentals;
//   else below the threshold: both from ONE logarithm — delta = ln(1 + e), a = (1 + e)^A = 2^(A log2(1 + e)), the logarithm being
//     v_log_f32 plus the first-order (Kahan) correction for the rounding of 1 + e;
//   above the threshold: delta = x.
// xl2 = x * log2(e) (callers that build x from a dot product fold the factor into the weights and the bias: one multiply per step less),
// Al2 = A * log2(e).
__device__ __forceinline__ void xp_softplus_decay_l2(float xl2, float Al2, float& delta, float& a);
__device__ __forceinline__ void xp_softplus_decay(float x, float A, float& delta, float& a) {
    xp_softplus_decay_l2(x * 1.44269504088896340736f, A * 1.44269504088896340736f, delta, a);
}
__device__ __forceinline__ void xp_softplus_decay_l2(float xl2, float Al2, float& delta, float& a) {
    const float e = __builtin_amdgcn_exp2f(xl2);
    if (e <= 0.14f) {
        float q = fmaf(e, -0.125f, 0.142857142857142857f);
        q = fmaf(q, e, -0.166666666666666667f);
        q = fmaf(q, e, 0.2f);
        q = fmaf(q, e, -0.25f);
        q = fmaf(q, e, 0.333333333333333333f);
        q = fmaf(q, e, -0.5f);
        q = fmaf(q, e, 1.f);
        delta = q * e;
        a = __builtin_amdgcn_exp2f(Al2 * delta);
    } else if (xl2 <= 20.f * 1.44269504088896340736f) {
        const float uu = 1.f + e;
        const float l2 = __builtin_amdgcn_logf(uu);                                   // log2(1 + e), ~1 ulp
        const float cc = ((uu - 1.f) - e) * __builtin_amdgcn_rcpf(uu);                // natural-log units
        delta = l2 * 0.693147180559945309f - cc;
        a = __builtin_amdgcn_exp2f(Al2 * delta);
    } else {
        delta = xl2 * 0.693147180559945309f;
        a = __builtin_amdgcn_exp2f(Al2 * delta);
    }
}
// Branch-free form of the same function for kernels that run one wave per SIMD (every instruction is issue time and a divergent
// branch executes both sides anyway): the logarithm path for every x below the threshold — v_log_f32 plus the Kahan correction keeps
// ~2e-7 relative down to e -> 0 (see xp_log1p_fast) — and a select for the linear tail.  12 vector instructions, 4 of them
// transcendental, no scalar control flow.  Differs from xp_softplus_decay_l2 by rounding only (<= ~3e-7 relative in delta).
__device__ __forceinline__ void xp_softplus_decay_l2_nb(float xl2, float Al2, float& delta, float& a) {
    const float e = __builtin_amdgcn_exp2f(xl2);
    const float uu = 1.f + e;
    const float l2 = __builtin_amdgcn_logf(uu);
    const float cc = ((uu - 1.f) - e) * __builtin_amdgcn_rcpf(uu);
    const float dl = l2 * 0.693147180559945309f - cc;
    delta = xl2 <= 20.f * 1.44269504088896340736f ? dl : xl2 * 0.693147180559945309f;
    a = __builtin_amdgcn_exp2f(Al2 * delta);
}
__device__ __forceinline__ float xp_softplus_fast(float x) { return x <= 20.f ? xp_log1p_fast(xp_exp_fast(x)) : x; }
__device__ __forceinline__ float xp_silu(float x) { return x * __builtin_amdgcn_rcpf(1.f + xp_exp_fast(-x)); }   // ~3 ulp
__device__ __forceinline__ float xp_gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
// GELU(erf) with erfc(|z|) = poly(t) * exp(-z^2), t = 1/(1 + p|z|)  (Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7):
// x >= 0: 0.5 x (2 - erfc), x < 0: 0.5 x erfc — no cancellation on the negative side.  ~15 VALU ops instead of
// libm erff's ~60; absolute error of the result <= ~4e-7 * max(1, |x|).
__device__ __forceinline__ float xp_gelu_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    // exp(-z^2) straight on the exp2 unit: the argument scaling costs |z^2| * 6e-8 relative on a factor that is itself <= erfc,
    // i.e. nothing where erfc matters (checked against erf in fp64: max abs error of the GELU 4.1e-7)
    const float erfc_abs = p * t * __builtin_amdgcn_exp2f(z * z * -1.44269504088896340736f);
    return 0.5f * x * (x >= 0.f ? 2.f - erfc_abs : erfc_abs);
}

// Cross-lane reductions on DPP moves (VALU): hipcc lowers __shfl_xor to ds_bpermute_b32, an LDS crossbar round trip per step.
// row_ror:n rotates inside a row of 16 lanes, so after the four steps every lane of a row holds the row's result.
template <int CTRL>
__device__ __forceinline__ float xp_dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float xp_row16_sum(float v) {
    v += xp_dpp_mov<0x128>(v);      // row_ror:8
    v += xp_dpp_mov<0x124>(v);      // row_ror:4
    v += xp_dpp_mov<0x122>(v);      // row_ror:2
    v += xp_dpp_mov<0x121>(v);      // row_ror:1
    return v;
}
// sum over each aligned group of 8 lanes (every lane of the group gets it)
__device__ __forceinline__ float xp_row8_sum(float v) {
    v += xp_dpp_mov<0xB1>(v);       // quad_perm [1,0,3,2]
    v += xp_dpp_mov<0x4E>(v);       // quad_perm [2,3,0,1]
    v += xp_dpp_mov<0x141>(v);      // row_half_mirror: the other quad of the 8-lane half row
    return v;
}
__device__ __forceinline__ float xp_row16_max(float v) {
    v = fmaxf(v, xp_dpp_mov<0x128>(v));
    v = fmaxf(v, xp_dpp_mov<0x124>(v));
    v = fmaxf(v, xp_dpp_mov<0x122>(v));
    v = fmaxf(v, xp_dpp_mov<0x121>(v));
    return v;
}
__device__ __forceinline__ float xp_lane(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ float xp_wave_sum(float v) {        // every lane gets the sum of the 64 lanes
    v = xp_row16_sum(v);
    return (xp_lane(v, 0) + xp_lane(v, 16)) + (xp_lane(v, 32) + xp_lane(v, 48));
}
__device__ __forceinline__ float xp_wave_max(float v) {
    v = xp_row16_max(v);
    return fmaxf(fmaxf(xp_lane(v, 0), xp_lane(v, 16)), fmaxf(xp_lane(v, 32), xp_lane(v, 48)));
}
#endif
