// Stand-alone four-route cross scan / cross merge: the operator pair of reference xpoint/models/vmamba_src/csm_triton.py:22-85 (cross_scan_fwd /
// cross_merge_fwd), :88-190 (the one_by_one forms) behind cross_scan_fn / cross_merge_fn (:501-517).  The fused encoder (ss2d.hip) never materialises
// the routes; these kernels exist for the OPERATOR-LEVEL drop-in (a reference SS2D on ROCm keeps its own forward_corev2 and swaps
// cross_scan_fn / selective_scan_fn / cross_merge_fn only) — SURVEY.md §8(b) minimum C-ABI set, row a7.
//
// Semantics (L = H W, pixel p = h W + w).  Route k of `scans`:
//     scans 0 (cross scan):     k = 0 row-major, 1 column-major (l = w H + h), 2 / 3 the reverses of 0 / 1
//     scans 1 (unidirectional): four copies of route 0
//     scans 2 (bidirectional):  0, 1 = route 0; 2, 3 = its reverse
// scan:   y[b, k, c, l] = x[b, (k,) c, pixel_k(l)]
// merge:  out[b, c, p] = (ys0 + ys2) + (ys1 + ys3) with ys_k taken at l = pos_k(p) — exactly the reference's association for scans 0 and 2
//         (csm_triton.py:60-62, :66-67: `y[:, 0:2] + y[:, 2:4].flip(...)`, then `y[:, 0] + y[:, 1]`); scans 1 is torch's `y.sum(1)` over four
//         addends, which ATen evaluates left to right in a float32 accumulator with one final rounding (pinned by tests/golden/g22: random data, bit-exact);
//         one_by_one merge is the inverse permutation per route, no adds.
// Layouts: channel-first x (B,[4,]C,H,W), y (B,4,C,L); channel-last x (B,H,W,[4,]C), y (B,L,4,C); the merge reads ys in the SCAN's out layout
// (`out_channel_first`) and writes the scan's in layout (`in_channel_first`), as the reference names them.
// dtype 0 float32, 1 float16, 2 bfloat16 (the reference under `mixed_precision` hands half tensors to cross_scan_fn); permutations move bits, the merge
// adds round to the tensor's dtype after every add like torch's elementwise kernels do.
//
// CDNA4 mapping: HBM-bound permutations.  (i) channel-first -> channel-first, scans 0 — the reference model's call (VMamba.py:603, :632) — runs on
// 64 x 64 pixel tiles of one (b, c) plane through a padded LDS tile, so the row-major routes AND the column-major routes are read and written in 256-byte
// runs (the Triton kernel it replaces uses 32 x 32 tiles, csm_triton.py:417-428); (ii) every other combination runs a gather kernel with one thread per
// OUTPUT element (stores always coalesced; channel-last <-> channel-last loads are contiguous C-vectors too; the mixed layouts pay strided loads).
#include "xp_common.h"
#include "../../include/xpoint_hip.h"

namespace {

struct CsParams {
    const void* src; void* dst;
    int B, C, H, W;
    int in_cf, out_cf, one_by_one, scans;
};

__device__ __forceinline__ void cs_route(int k, int scans, bool& T, bool& F) { T = scans == 0 && (k & 1); F = scans != 1 && (k >> 1); }
// position l of route k -> pixel index
__device__ __forceinline__ int cs_pixel(int k, int scans, int l, int H, int W) {
    bool T, F; cs_route(k, scans, T, F);
    const int L = H * W, lp = F ? L - 1 - l : l;
    return T ? (lp % H) * W + lp / H : lp;
}
// pixel index -> position in route k
__device__ __forceinline__ int cs_pos(int k, int scans, int p, int H, int W) {
    bool T, F; cs_route(k, scans, T, F);
    const int L = H * W, lp = T ? (p % W) * H + p / W : p;
    return F ? L - 1 - lp : lp;
}

template <typename T> struct CsNum;
template <> struct CsNum<float> { static __device__ __forceinline__ float ld(float v) { return v; } static __device__ __forceinline__ float st(float v) { return v; } };
template <> struct CsNum<_Float16> { static __device__ __forceinline__ float ld(_Float16 v) { return (float)v; } static __device__ __forceinline__ _Float16 st(float v) { return (_Float16)v; } };
struct cs_bf16 { unsigned short u; };
template <> struct CsNum<cs_bf16> {
    static __device__ __forceinline__ float ld(cs_bf16 v) { return __uint_as_float((unsigned)v.u << 16); }
    static __device__ __forceinline__ cs_bf16 st(float f) {            // round to nearest even (NaN kept quiet)
        unsigned u = __float_as_uint(f);
        if ((u & 0x7fffffffu) > 0x7f800000u) return cs_bf16{(unsigned short)((u >> 16) | 0x40u)};
        u += 0x7fffu + ((u >> 16) & 1u);
        return cs_bf16{(unsigned short)(u >> 16)};
    }
};
// the sum of two values of the tensor's dtype, rounded to that dtype (what torch's add kernel returns)
template <typename T> __device__ __forceinline__ T cs_add(T a, T b) { return CsNum<T>::st(CsNum<T>::ld(a) + CsNum<T>::ld(b)); }

// ---- gather form: one thread per output element -----------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void cross_scan_gather_kernel(CsParams p, int64_t total) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= total) return;
    const int L = p.H * p.W, C = p.C;
    int b, k, c, l;
    if (p.out_cf) { l = (int)(o % L); int64_t r = o / L; c = (int)(r % C); r /= C; k = (int)(r & 3); b = (int)(r >> 2); }
    else { c = (int)(o % C); int64_t r = o / C; k = (int)(r & 3); r >>= 2; l = (int)(r % L); b = (int)(r / L); }
    const int px = cs_pixel(k, p.scans, l, p.H, p.W);
    int64_t s;
    if (p.in_cf) s = p.one_by_one ? (((int64_t)(b * 4 + k) * C + c) * L + px) : (((int64_t)b * C + c) * L + px);
    else s = p.one_by_one ? ((((int64_t)b * L + px) * 4 + k) * C + c) : (((int64_t)b * L + px) * C + c);
    static_cast<T*>(p.dst)[o] = static_cast<const T*>(p.src)[s];
}

template <typename T>
__global__ __launch_bounds__(256) void cross_merge_gather_kernel(CsParams p, int64_t total) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= total) return;
    const int L = p.H * p.W, C = p.C;
    const T* ys = static_cast<const T*>(p.src);
    auto src_at = [&](int b, int k, int c, int l) -> T {
        return p.out_cf ? ys[((int64_t)(b * 4 + k) * C + c) * L + l] : ys[(((int64_t)b * L + l) * 4 + k) * C + c];
    };
    if (p.one_by_one) {
        int b, k, c, px;
        if (p.in_cf) { px = (int)(o % L); int64_t r = o / L; c = (int)(r % C); r /= C; k = (int)(r & 3); b = (int)(r >> 2); }
        else { c = (int)(o % C); int64_t r = o / C; k = (int)(r & 3); r >>= 2; px = (int)(r % L); b = (int)(r / L); }
        static_cast<T*>(p.dst)[o] = src_at(b, k, c, cs_pos(k, p.scans, px, p.H, p.W));
        return;
    }
    int b, c, px;
    if (p.in_cf) { px = (int)(o % L); int64_t r = o / L; c = (int)(r % C); b = (int)(r / C); }
    else { c = (int)(o % C); int64_t r = o / C; px = (int)(r % L); b = (int)(r / L); }
    T v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = src_at(b, k, c, cs_pos(k, p.scans, px, p.H, p.W));
    T r;
    if (p.scans == 1) r = CsNum<T>::st(((CsNum<T>::ld(v[0]) + CsNum<T>::ld(v[1])) + CsNum<T>::ld(v[2])) + CsNum<T>::ld(v[3]));   // y.sum(1): left to right in a
                                                                                                                              // float32 accumulator, ONE rounding (pinned by g22)
    else r = cs_add(cs_add(v[0], v[2]), cs_add(v[1], v[3]));                     // csm_triton.py:60-62 / :66-67
    static_cast<T*>(p.dst)[o] = r;
}

// ---- tiled form: channel-first both sides, scans 0, not one_by_one ---------------------------------------------------------------------
constexpr int CS_T = 64;       // tile edge (pixels); 256 threads: 4 tile rows per step, 16 steps

template <typename T>
__global__ __launch_bounds__(256) void cross_scan_tile_kernel(const T* __restrict__ x, T* __restrict__ y, int C, int H, int W, int tiles_w) {
    __shared__ T tile[CS_T][CS_T + 1 + (sizeof(T) == 2)];        // odd dword stride for 4-byte elements; 66 halves = 33 dwords for 2-byte ones
    const int c = blockIdx.y, b = blockIdx.z;
    const int h0 = (blockIdx.x / tiles_w) * CS_T, w0 = (blockIdx.x % tiles_w) * CS_T;
    const int64_t L = (int64_t)H * W;
    const T* xp = x + ((int64_t)b * C + c) * L;
    T* y0 = y + ((int64_t)(b * 4 + 0) * C + c) * L; T* y1 = y + ((int64_t)(b * 4 + 1) * C + c) * L;
    T* y2 = y + ((int64_t)(b * 4 + 2) * C + c) * L; T* y3 = y + ((int64_t)(b * 4 + 3) * C + c) * L;
    const int col = threadIdx.x & 63, r4 = threadIdx.x >> 6;
#pragma unroll 4
    for (int i = 0; i < CS_T / 4; ++i) {
        const int hl = i * 4 + r4, h = h0 + hl, w = w0 + col;
        if (h < H && w < W) {
            const T v = xp[(int64_t)h * W + w];
            tile[hl][col] = v;
            const int64_t l = (int64_t)h * W + w;
            y0[l] = v; y2[L - 1 - l] = v;
        }
    }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < CS_T / 4; ++i) {
        const int wl = i * 4 + r4, w = w0 + wl, h = h0 + col;       // lanes run along h: the column-major routes are written in runs
        if (h < H && w < W) {
            const T v = tile[col][wl];
            const int64_t l = (int64_t)w * H + h;
            y1[l] = v; y3[L - 1 - l] = v;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cross_merge_tile_kernel(const T* __restrict__ ys, T* __restrict__ out, int C, int H, int W, int tiles_w) {
    __shared__ T tile[CS_T][CS_T + 1 + (sizeof(T) == 2)];
    const int c = blockIdx.y, b = blockIdx.z;
    const int h0 = (blockIdx.x / tiles_w) * CS_T, w0 = (blockIdx.x % tiles_w) * CS_T;
    const int64_t L = (int64_t)H * W;
    const T* y0 = ys + ((int64_t)(b * 4 + 0) * C + c) * L; const T* y1 = ys + ((int64_t)(b * 4 + 1) * C + c) * L;
    const T* y2 = ys + ((int64_t)(b * 4 + 2) * C + c) * L; const T* y3 = ys + ((int64_t)(b * 4 + 3) * C + c) * L;
    T* op = out + ((int64_t)b * C + c) * L;
    const int col = threadIdx.x & 63, r4 = threadIdx.x >> 6;
#pragma unroll 4
    for (int i = 0; i < CS_T / 4; ++i) {
        const int wl = i * 4 + r4, w = w0 + wl, h = h0 + col;
        if (h < H && w < W) {
            const int64_t l = (int64_t)w * H + h;
            tile[col][wl] = cs_add(y1[l], y3[L - 1 - l]);
        }
    }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < CS_T / 4; ++i) {
        const int hl = i * 4 + r4, h = h0 + hl, w = w0 + col;
        if (h < H && w < W) {
            const int64_t l = (int64_t)h * W + w;
            op[l] = cs_add(cs_add(y0[l], y2[L - 1 - l]), tile[hl][col]);
        }
    }
}

template <typename T>
int cs_run(bool merge, const CsParams& p, hipStream_t s) {
    const int64_t L = (int64_t)p.H * p.W;
    const bool tiled = p.in_cf && p.out_cf && !p.one_by_one && p.scans == 0 && p.C <= 65535 && p.B <= 65535;
    const int64_t n_routes = (int64_t)p.B * 4 * p.C * L, n_plain = (int64_t)p.B * p.C * L;
    const double bytes = (double)sizeof(T) * (merge ? (p.one_by_one ? 2 * n_routes : n_routes + n_plain) : (p.one_by_one ? 2 * n_routes : n_routes + n_plain));
    XpProfScope prof(merge ? (tiled ? "cross_merge_tile" : "cross_merge_gather") : (tiled ? "cross_scan_tile" : "cross_scan_gather"), s, 0.0, bytes);
    if (tiled) {
        const int tw = xp_cdiv(p.W, CS_T), th = xp_cdiv(p.H, CS_T);
        const dim3 grid(tw * th, p.C, p.B);
        if (merge) hipLaunchKernelGGL(cross_merge_tile_kernel<T>, grid, dim3(256), 0, s, static_cast<const T*>(p.src), static_cast<T*>(p.dst), p.C, p.H, p.W, tw);
        else hipLaunchKernelGGL(cross_scan_tile_kernel<T>, grid, dim3(256), 0, s, static_cast<const T*>(p.src), static_cast<T*>(p.dst), p.C, p.H, p.W, tw);
    } else {
        const int64_t total = merge ? (p.one_by_one ? n_routes : n_plain) : n_routes;
        const unsigned blocks = (unsigned)((total + 255) / 256);
        if (merge) hipLaunchKernelGGL(cross_merge_gather_kernel<T>, dim3(blocks), dim3(256), 0, s, p, total);
        else hipLaunchKernelGGL(cross_scan_gather_kernel<T>, dim3(blocks), dim3(256), 0, s, p, total);
    }
    XP_LAUNCH_CHECK();
    return XP_OK;
}

int cs_entry(bool merge, const void* src, void* dst, int dtype, int B, int C, int H, int W, int in_cf, int out_cf, int one_by_one, int scans, void* stream) {
    const char* who = merge ? "xp_cross_merge" : "xp_cross_scan";
    XP_CHECK_ARG(src && dst, "%s: null pointer", who);
    XP_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0, "%s: bad shape (%d, %d, %d, %d)", who, B, C, H, W);
    XP_CHECK_ARG(scans >= 0 && scans <= 2, "%s: scans must be 0 (cross), 1 (unidirectional) or 2 (bidirectional), got %d", who, scans);
    XP_CHECK_ARG(dtype >= 0 && dtype <= 2, "%s: dtype 0 float32, 1 float16, 2 bfloat16 (got %d)", who, dtype);
    XP_CHECK_ARG((int64_t)H * W < (1ll << 31) && (int64_t)B * 4 * C * H * W < (1ll << 40), "%s: tensor too large", who);
    XP_CHECK_ARG((int64_t)B * 4 < (1ll << 29), "%s: batch too large", who);
    CsParams p{src, dst, B, C, H, W, in_cf != 0, out_cf != 0, one_by_one != 0, scans};
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case 0: return cs_run<float>(merge, p, s);
        case 1: return cs_run<_Float16>(merge, p, s);
        default: return cs_run<cs_bf16>(merge, p, s);
    }
}

}  // namespace

extern "C" int xp_cross_scan(const void* x, void* y, int dtype, int B, int C, int H, int W, int in_channel_first, int out_channel_first,
                             int one_by_one, int scans, void* stream) {
    return cs_entry(false, x, y, dtype, B, C, H, W, in_channel_first, out_channel_first, one_by_one, scans, stream);
}

extern "C" int xp_cross_merge(const void* ys, void* out, int dtype, int B, int C, int H, int W, int in_channel_first, int out_channel_first,
                              int one_by_one, int scans, void* stream) {
    return cs_entry(true, ys, out, dtype, B, C, H, W, in_channel_first, out_channel_first, one_by_one, scans, stream);
}
