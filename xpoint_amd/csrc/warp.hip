// Perspective warp of a batch of images: the last step of the reference's registration flow,
//     warped_image = cv2.warpPerspective(im_optical, H_est, im_optical.shape[:2][::-1], borderMode=cv2.BORDER_CONSTANT)
// (predict_align_image_pair.py:308; demo.py:225-249), i.e. flags = INTER_LINEAR, borderValue = 0, M = the FORWARD map src -> dst, which
// OpenCV inverts before the per-pixel inverse mapping.  SURVEY.md 8(f) rank 2.
//
// OpenCV is absent from /root/reference and from this image, so the arithmetic below is the DOCUMENTED scheme of OpenCV's
// imgproc (warpPerspective -> remap, INTER_BITS = 5, INTER_REMAP_COEF_BITS = 15) restated from its published source — "parity unpinned"
// (DESIGN.md section 4); the oracle (oracle/csrc/oracle_kernels.c: xo_warp_perspective_*) states the same scheme in plain C and the GPU
// tests demand bit-equality with it.
//   1. M^-1 by the closed 3x3 cofactor form in double (no LU): t = adj(M) * (1 / det), det == 0 -> the zero matrix.
//   2. per destination pixel (x, y), in double, with the block structure of OpenCV's WarpPerspectiveInvoker (the row base is formed at
//      the first column xb of the pixel's 64-wide block, the in-block offset x1 added afterwards):
//          X0 = m0 * xb + m1 * y + m2,  Y0 = m3 * xb + m4 * y + m5,  W0 = m6 * xb + m7 * y + m8
//          W = W0 + m6 * x1;  W = W ? 32 / W : 0
//          fX = clamp((X0 + m0 * x1) * W, INT_MIN, INT_MAX),  fY likewise;   X = lrint(fX), Y = lrint(fY)        (round half to even)
//          sx = sat16(X >> 5), sy = sat16(Y >> 5),  ax = X & 31, ay = Y & 31
//   3. bilinear taps at (sx, sy), (sx + 1, sy), (sx, sy + 1), (sx + 1, sy + 1); a tap outside the source reads the border value 0.
//        u8 : weights w = 32768 * (1 - ay / 32 | ay / 32) * (1 - ax / 32 | ax / 32) = exact integers (32 - ay | ay) * (32 - ax | ax) * 32,
//             out = (sum w_i * tap_i + 16384) >> 15
//        f32: weights (1 - fy) * (1 - fx), (1 - fy) * fx, fy * (1 - fx), fy * fx with fx = ax / 32 (exact in f32),
//             out = ((t0 * w0 + t1 * w1) + t2 * w2) + t3 * w3      (separate multiplies and adds, left to right; built with -ffp-contract=off)
// One thread per destination pixel (all channels), 64 x 4 pixels per workgroup: consecutive lanes write consecutive pixels; the four taps of
// neighbouring pixels share cache lines.  HBM-bound: a 480 x 640 u8 image is 0.3 MB in, 0.3 MB out.
#include "xp_common.h"
#include "../../include/xpoint_hip.h"

namespace {

struct WarpParams {
    const void* src; void* dst; const double* M;
    int Hs, Ws, Hd, Wd, C, Cd, inverse_map;
};

__device__ __forceinline__ void warp_invert3(const double* __restrict__ S, double (&t)[9]) {
    // OpenCV cv::invert, 3 x 3 double, DECOMP_LU: det3 and the cofactors in this operand order
    double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
    if (d != 0.0) {
        d = 1.0 / d;
        t[0] = (S[4] * S[8] - S[5] * S[7]) * d;
        t[1] = (S[2] * S[7] - S[1] * S[8]) * d;
        t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
        t[3] = (S[5] * S[6] - S[3] * S[8]) * d;
        t[4] = (S[0] * S[8] - S[2] * S[6]) * d;
        t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
        t[6] = (S[3] * S[7] - S[4] * S[6]) * d;
        t[7] = (S[1] * S[6] - S[0] * S[7]) * d;
        t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
    } else {
        for (int k = 0; k < 9; ++k) t[k] = 0.0;
    }
}

__device__ __forceinline__ int warp_sat16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

// source element (channel c) at (sx, sy), 0 outside.  MODE 0: u8 source; 1: f32 source; 2: f32 source quantised on load as the reference
// does before warping: (np.clip(img, 0, 1) * 255.0).astype(np.uint8)  (predict_align_image_pair.py:271: f32 multiply, truncation)
template <int MODE>
__device__ __forceinline__ auto warp_tap(const void* __restrict__ src, int Hs, int Ws, int C, int sx, int sy, int c) {
    const bool in = (unsigned)sx < (unsigned)Ws && (unsigned)sy < (unsigned)Hs;
    const size_t off = ((size_t)(in ? sy : 0) * Ws + (in ? sx : 0)) * C + c;
    if constexpr (MODE == 0) {
        return in ? (int)reinterpret_cast<const uint8_t*>(src)[off] : 0;
    } else if constexpr (MODE == 1) {
        return in ? reinterpret_cast<const float*>(src)[off] : 0.f;
    } else {
        float v = reinterpret_cast<const float*>(src)[off];
        v = fminf(fmaxf(v, 0.f), 1.f) * 255.0f;        // NaN clips to 0 here (numpy would propagate it; a NaN pixel has no u8 value either way)
        return in ? (int)v : 0;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void warp_perspective_kernel(WarpParams p) {
    __shared__ double s_m[9];
    const int b = blockIdx.z;
    if (threadIdx.x == 0) {
        const double* S = p.M + (size_t)b * 9;
        double t[9];
        if (p.inverse_map) { for (int k = 0; k < 9; ++k) t[k] = S[k]; } else warp_invert3(S, t);
        for (int k = 0; k < 9; ++k) s_m[k] = t[k];
    }
    __syncthreads();
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.Wd || y >= p.Hd) return;
    const int bw0 = p.Wd < 64 ? p.Wd : 64;                 // OpenCV's block width: min(BLOCK_SZ * BLOCK_SZ / min(BLOCK_SZ / 2, height), width) = min(64, width) for height >= 16
    const int bw = p.Hd >= 16 ? bw0 : (p.Wd < 1024 / p.Hd ? p.Wd : 1024 / p.Hd);
    const int xb = x / bw * bw, x1 = x - xb;
    const double X0 = s_m[0] * xb + s_m[1] * y + s_m[2];
    const double Y0 = s_m[3] * xb + s_m[4] * y + s_m[5];
    const double W0 = s_m[6] * xb + s_m[7] * y + s_m[8];
    double W = W0 + s_m[6] * x1;
    W = W != 0.0 ? 32.0 / W : 0.0;
    const double fX = fmax(-2147483648.0, fmin(2147483647.0, (X0 + s_m[0] * x1) * W));
    const double fY = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + s_m[3] * x1) * W));
    const int X = __double2int_rn(fX), Y = __double2int_rn(fY);         // NaN (0 * inf) -> 0, as lrint's result is then unspecified in C
    const int sx = warp_sat16(X >> 5), sy = warp_sat16(Y >> 5), ax = X & 31, ay = Y & 31;
    const size_t src_off = (size_t)b * p.Hs * p.Ws * p.C;
    const size_t dst_off = (((size_t)b * p.Hd + y) * p.Wd + x) * p.Cd;
    const void* src = MODE == 0 ? (const void*)(reinterpret_cast<const uint8_t*>(p.src) + src_off) : (const void*)(reinterpret_cast<const float*>(p.src) + src_off);
    for (int c = 0; c < p.Cd; ++c) {
        const int cs = c < p.C ? c : p.C - 1;               // Cd > C: a 1-channel source replicated (cv2.cvtColor(.., COLOR_GRAY2RGB) ahead of the warp)
        const auto t0 = warp_tap<MODE>(src, p.Hs, p.Ws, p.C, sx, sy, cs), t1 = warp_tap<MODE>(src, p.Hs, p.Ws, p.C, sx + 1, sy, cs);
        const auto t2 = warp_tap<MODE>(src, p.Hs, p.Ws, p.C, sx, sy + 1, cs), t3 = warp_tap<MODE>(src, p.Hs, p.Ws, p.C, sx + 1, sy + 1, cs);
        if constexpr (MODE == 1) {
            const float fx = (float)ax * 0.03125f, fy = (float)ay * 0.03125f;
            const float w0 = (1.f - fy) * (1.f - fx), w1 = (1.f - fy) * fx, w2 = fy * (1.f - fx), w3 = fy * fx;
            reinterpret_cast<float*>(p.dst)[dst_off + c] = ((t0 * w0 + t1 * w1) + t2 * w2) + t3 * w3;
        } else {
            const int w0 = (32 - ay) * (32 - ax) * 32, w1 = (32 - ay) * ax * 32, w2 = ay * (32 - ax) * 32, w3 = ay * ax * 32;
            const int v = (t0 * w0 + t1 * w1 + t2 * w2 + t3 * w3 + 16384) >> 15;
            reinterpret_cast<uint8_t*>(p.dst)[dst_off + c] = (uint8_t)(v > 255 ? 255 : v);
        }
    }
}

}  // namespace

extern "C" int xp_warp_perspective(const void* src, void* dst, const double* M, int batch, int Hs, int Ws, int Hd, int Wd, int channels,
                                   int dst_channels, int dtype, int inverse_map, void* stream) {
    XP_CHECK_ARG(src && dst && M, "xp_warp_perspective: null pointer");
    XP_CHECK_ARG(batch > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0, "xp_warp_perspective: bad shape (batch %d, source %d x %d, destination %d x %d)", batch, Hs, Ws, Hd, Wd);
    XP_CHECK_ARG(Hs < 32768 && Ws < 32768 && Hd <= 65535 * 4 && batch <= 65535, "xp_warp_perspective: image too large (source coordinates are 16-bit, as in OpenCV's remap)");
    XP_CHECK_ARG(channels >= 1 && channels <= 4, "xp_warp_perspective: channels must be 1..4, got %d", channels);
    XP_CHECK_ARG(dst_channels == channels || (channels == 1 && dst_channels >= 1 && dst_channels <= 4),
                 "xp_warp_perspective: dst_channels must equal channels, or replicate a 1-channel source (got %d -> %d)", channels, dst_channels);
    XP_CHECK_ARG(dtype == XP_WARP_U8 || dtype == XP_WARP_F32 || dtype == XP_WARP_F32_AS_U8, "xp_warp_perspective: unknown dtype %d", dtype);
    XP_CHECK_ARG(((uintptr_t)M & 7) == 0 && (dtype == XP_WARP_U8 || ((uintptr_t)src & 3) == 0) && (dtype != XP_WARP_F32 || ((uintptr_t)dst & 3) == 0),
                 "xp_warp_perspective: misaligned pointer");
    XP_CHECK_ARG(src != dst, "xp_warp_perspective: in-place warp is not supported");
    WarpParams p{src, dst, M, Hs, Ws, Hd, Wd, channels, dst_channels, inverse_map ? 1 : 0};
    const dim3 grid(xp_cdiv(Wd, 64), xp_cdiv(Hd, 4), batch), block(256);
    const double px = (double)batch * Hd * Wd, eb = dtype == XP_WARP_F32 ? 4.0 : 1.0;
    XpProfScope prof("warp_perspective", (hipStream_t)stream, 0.0, px * dst_channels * eb + (double)batch * Hs * Ws * channels * (dtype == XP_WARP_U8 ? 1.0 : 4.0));
    if (dtype == XP_WARP_U8) hipLaunchKernelGGL(warp_perspective_kernel<0>, grid, block, 0, (hipStream_t)stream, p);
    else if (dtype == XP_WARP_F32) hipLaunchKernelGGL(warp_perspective_kernel<1>, grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(warp_perspective_kernel<2>, grid, block, 0, (hipStream_t)stream, p);
    XP_LAUNCH_CHECK();
    return XP_OK;
}
