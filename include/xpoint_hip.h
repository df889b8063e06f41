/* xpoint_hip.h — C ABI of libxpoint_hip.so: the MI355X (gfx950) XPoint inference hot path.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types cross this boundary.
 *   - every function returns int: 0 = ok, <0 = error (text: xp_last_error(), thread-local),
 *     except the *_bytes / *_numel size queries (size_t) and xp_version / xp_last_error.
 *   - the CALLER owns every buffer; pointers are device pointers (hipMalloc'd or torch
 *     `tensor.data_ptr()`), contiguous float32 unless stated, 16-byte aligned.
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream) and
 *     are stream-ordered; no function synchronises the device unless it says so.
 *   - the library never allocates device memory: weights and workspaces are caller buffers sized by
 *     the *_workspace_bytes queries.  A context (xp_ctx_create) is host-only metadata.
 *   - activations are NHWC (channels contiguous) inside the library.
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
/* The registry of the library's XP_* environment knobs (csrc/xp_knobs.h): tuning / A-B switches only, each read once per process; no knob changes a
 * precision class.  xp_knob_info returns static strings: the variable, the source that reads it, what it does. */
int xp_knob_count(void);
int xp_knob_info(int index, const char** name, const char** where, const char** what);

/* ---------------------------------------------------------------------------------------------
 * Selective-scan forward.  Replaces the pybind op `selective_scan_cuda_oflex.fwd(u, delta, A, B, C,
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

/* The same operator with the reference's other input_t instantiations (cusoflex/selective_scan_core_fwd.cu:6-10) and its
 * second output (selective_scan_oflex.cpp:206-208, kernel selective_scan_fwd_kernel_oflex.cuh:154-162):
 *   itype 0 / 1 / 2: u, delta, B, C are f32 / f16 / bf16 (A, D, delta_bias always f32; the state is always f32);
 *   out_float != 0: out is f32 (the "oflex" form the Python caller uses, csms6s.py:80), else out has the input type;
 *   x_chunks (batch, dim, ceil(seqlen / 2048), 2 * dstate) f32 or NULL: per 2048-element chunk and state n the running
 *   prefix of the scan, x[.., 2n] = prod exp(delta A_n) since the start of the row, x[.., 2n + 1] = h_n at the end of the
 *   chunk (last state = x[:, :, -1, 1::2]).  The 16-bit instantiations evaluate exp(delta A) as exp2 on the hardware unit like
 *   the reference's kernel (reference tolerances f16 rtol 3e-3 / atol 5e-3, bf16 3e-2 / 5e-2: test_selective_scan.py:401-403). */
int xp_selective_scan_fwd_typed(const void* u, const void* delta, const float* A, const void* B, const void* C, const float* D,
                                const float* delta_bias, void* out, float* x_chunks, int itype, int out_float, int batch, int dim,
                                int delta_dim, int seqlen, int dstate, int ngroups, int delta_softplus, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Stand-alone four-route cross scan / cross merge.  Replace `cross_scan_fn` / `cross_merge_fn`
 * (xpoint/models/vmamba_src/csm_triton.py:501-517; torch forms cross_scan_fwd :22-53, cross_merge_fwd :56-85, one_by_one forms :88-180;
 * Triton kernel triton_cross_scan_flex :278-400) for the operator-level drop-in; the fused encoder never materialises the routes
 * (xp_ss2d_core_fwd below).  dtype 0 / 1 / 2 = float32 / float16 / bfloat16 elements.
 *   scans 0: routes row-major, column-major and their reverses; 1: four copies of the row-major route; 2: row-major twice, reversed twice.
 *   xp_cross_scan:  x (B,C,H,W) if in_channel_first else (B,H,W,C)  [one_by_one: (B,4,C,H,W) / (B,H,W,4,C), route k reads its own slice]
 *                   -> y (B,4,C,H*W) if out_channel_first else (B,H*W,4,C).
 *   xp_cross_merge: ys in the scan's OUT layout ((B,4,C,H*W) if out_channel_first else (B,H*W,4,C)) -> out in the scan's IN layout
 *                   ((B,C,H*W) if in_channel_first else (B,H*W,C)): (ys0 + ys2) + (ys1 + ys3) at every pixel, each add rounded to the dtype
 *                   (scans 1: ((ys0 + ys1) + ys2) + ys3 in a float32 accumulator, one rounding — torch's `sum(1)`);
 *                   one_by_one: the inverse permutation per route -> (B,4,C,H*W) / (B,H*W,4,C), no adds.
 * Bit-exact against the reference functions on random data (tests/golden/g22_cross_scan_ops.npz), incl. its own check shape (27,253,57,58). */
int xp_cross_scan(const void* x, void* y, int dtype, int B, int C, int H, int W, int in_channel_first, int out_channel_first,
                  int one_by_one, int scans, void* stream);
int xp_cross_merge(const void* ys, void* out, int dtype, int B, int C, int H, int W, int in_channel_first, int out_channel_first,
                   int one_by_one, int scans, void* stream);

/* Fused SS2D core in pixel layout = cross_scan + dt_proj + selective_scan + cross_merge + out_norm of
 * xpoint/models/vmamba_src/VMamba.py:601-646 (forward_corev2) with csm_triton.py:22-85 (cross scan/merge).
 *   u (batch, H, W, C) = SiLU(dwconv(in_proj(x)));  xdbl (batch*H*W, 4*(R+2)) = u @ x_proj^T with the four
 *   directions stored in the order (0, 2, 1, 3), each [dt_rank values, B, C];  wdt (4, R, C) (= dt_projs_weight transposed: channel-contiguous), dt_bias (4, C),
 *   A (4, C) = -exp(A_logs), Ds (4, C) in the same direction order;  ln_w/ln_b = out_norm;  out (batch, H, W, C).
 *   d_state must be 1 (the XPoint config; general d_state: xp_selective_scan_fwd).
 *   Two internal forms with the same results to f32 rounding: chunked three-pass (long sequences) and sequential per
 *   (image, route, 64 channels) wave (short sequences with many channels: the deep encoder stages); chosen per call
 *   shape, or forced with xp_ss2d_core_set_mode(0 chunked / 1 sequential / -1 automatic) for tests and A/B timing. */
size_t xp_ss2d_core_workspace_bytes(int batch, int H, int W, int C);
int xp_ss2d_core_set_mode(int mode);
int xp_ss2d_core_fwd(const float* u, const float* xdbl, const float* wdt, const float* dt_bias, const float* A,
                     const float* Ds, const float* ln_w, const float* ln_b, float* out, float* workspace,
                     size_t workspace_bytes, int batch, int H, int W, int C, int R, int dstate, float eps,
                     void* stream);
/* The same core with a selectable output format: out_fmt 0 = f32 rows (identical to xp_ss2d_core_fwd), 2 = the P32 image of the (B H W, C) result
 * (see xp_gemm_nt_h2s: the SS2D out_proj then loads it by DMA).  The P32 form exists where the sequential deep-stage form runs:
 * xp_ss2d_core_p32_supported(H, W, C, R) != 0 (per-image quantities only); elsewhere out_fmt = 2 is an argument error. */
int xp_ss2d_core_p32_supported(int H, int W, int C, int R);
int xp_ss2d_core_fwd_ex(const float* u, const float* xdbl, const float* wdt, const float* dt_bias, const float* A,
                        const float* Ds, const float* ln_w, const float* ln_b, void* out, int out_fmt, float* workspace,
                        size_t workspace_bytes, int batch, int H, int W, int C, int R, int dstate, float eps,
                        void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dense layers (nn.Linear / nn.Conv2d of VMamba.py:110-128,649,663,1405-1440 and XPoint.py:112-138).
 *   C[m,n] = (act(sum_k A[m,k] Wt[n,k] + bias[n]) * scale[n] + shift[n]) + res[m,n]
 *   act: 0 none, 1 GELU(erf), 2 ReLU, 3 = ReLU applied after scale/shift (conv -> BN -> ReLU);
 *   bias/scale/shift/res may be NULL (scale and shift together);
 *   K and lda multiples of 4.  xp_conv3x3_nhwc: NHWC input, weight (Co, 3, 3, Ci), pad 1 (zero or
 *   reflection), stride 1 or 2, output NHWC (batch, Ho, Wo, Co). */
int xp_gemm_nt(const float* A, const float* Wt, float* C, const float* bias, const float* scale, const float* shift,
               const float* res, int M, int N, int K, int lda, int ldc, int ldres, int act, void* stream);
int xp_conv3x3_nhwc(const float* x, const float* Wt, float* y, const float* bias, const float* scale,
                    const float* shift, int batch, int Hi, int Wi, int Ci, int Co, int stride, int reflect_pad,
                    int act, void* stream);

/* fp32-accurate variants on the bf16 matrix pipe ("x3": every f32 operand is the exact sum of three bf16
 * values; six bf16 MFMA partial products per multiply, f32 accumulate; error below an f32 FMA chain —
 * DESIGN.md §4).  Same semantics, epilogue and reference call sites as the two entry points above, but
 * the weight matrix is passed pre-split: xp_split_weights_x3 converts a row-major (N, K) f32 matrix
 * (for the conv: (Co, 3, 3, Ci) flattened, K = 9 Ci) into xp_split_weights_x3_bytes(N, K) bytes of
 * slab-major bf16 planes, once per weight upload.  K, lda (resp. Ci) multiples of 4, as above. */
size_t xp_split_weights_x3_bytes(int N, int K);
int xp_split_weights_x3(const float* W, void* out, int N, int K, void* stream);
int xp_gemm_nt_x3(const float* A, const void* Wx3, float* C, const float* bias, const float* scale,
                  const float* shift, const float* res, int M, int N, int K, int lda, int ldc, int ldres, int act,
                  void* stream);
int xp_conv3x3_nhwc_x3(const float* x, const void* Wx3, float* y, const float* bias, const float* scale,
                       const float* shift, int batch, int Hi, int Wi, int Ci, int Co, int stride, int reflect_pad,
                       int act, void* stream);
/* fp32-grade variants on the f16 matrix pipe ("h2": every f32 operand is written as the sum of two fp16 values h0 + h1, which
 * reproduces it to 2^-23 relative in the worst case (fp16 has an 11-bit significand: |x - h0| <= 2^-11 |x|, and the 12-bit remainder
 * loses at most one bit in h1), 2^-25.5 on average; three fp16 MFMA partial products per multiply (a0 b0 + a0 b1 + a1 b0; the dropped
 * a1 b1 <= 2^-22 |ab|), f32 accumulate: per product <= 2^-21 |ab| worst case, ~2^-25 typical — the error class of the f32 FMA chain the
 * reference computes (csrc/gemm_h2_core.h) at half the matrix work of the "x3" kernels, which are exact to 2^-24 per operand).
 * Same semantics, epilogue and reference call sites as
 * xp_gemm_nt / xp_conv3x3_nhwc (VMamba.py:649,663 in/out_proj; :110-128 Mlp; :605 x_proj; :1405-1440 convs;
 * XPoint.py:112-138 head convs).  The weight matrix is passed pre-split: xp_split_weights_h2 converts a row-major (N, K) f32
 * matrix into xp_split_weights_h2_bytes(N, K) bytes — slab-major fp16 planes of the rows scaled by a power of two each
 * (largest element in [2^13, 2^14): keeps the low plane out of fp16's subnormal range) followed by the N inverse scales,
 * which the epilogue applies exactly.  Activations are split unscaled: |A| must stay below 65504 (fp16 range; see
 * xp_xpoint_forward_ex for the guard); elements below 2^-3 carry an absolute error <= 2^-25. */
size_t xp_split_weights_h2_bytes(int N, int K);
int xp_split_weights_h2(const float* W, void* out, int N, int K, void* stream);
int xp_gemm_nt_h2(const float* A, const void* Wh2, float* C, const float* bias, const float* scale, const float* shift,
                  const float* res, int M, int N, int K, int lda, int ldc, int ldres, int act, void* stream);
int xp_conv3x3_nhwc_h2(const float* x, const void* Wh2, float* y, const float* bias, const float* scale, const float* shift,
                       int batch, int Hi, int Wi, int Ci, int Co, int stride, int reflect_pad, int act, void* stream);
/* The same GEMM on PRE-SPLIT activations (round 5; csrc/gemm_ring.hip, csrc/ring_core.h — the "ring" engine; same reference call sites as xp_gemm_nt_h2:
 * VMamba.py:649,663 in/out_proj, :110-128 Mlp).  A is the "P32" image of an (M, K) f32 matrix: [m][k / 32][plane 0..1][32] fp16 with
 * A[m][k] = plane0 + plane1 (the two-way split of csrc/gemm_h2_core.h) — 4 bytes per element, K % 32 == 0, xp_p32_bytes(M, K) bytes.  It is written
 * by the PRODUCER of the tensor: xp_split_activations_h2 (from f32 rows), xp_layernorm_p32, xp_ss2d_core_fwd_ex(out_fmt = 2), or this GEMM itself
 * (out_fmt = 2: C is the P32 image of the (M, N) result, N % 32 == 0, ldc == N).  out_fmt 0: C f32 rows (ldc).  Wh2: xp_split_weights_h2's buffer.
 * Both operands reach the LDS by LDS-DMA and the K loop holds matrix instructions only; same partial products, same summation order and therefore the
 * same bits as xp_gemm_nt_h2's tile kernel on A = plane0 + plane1.  N % 8 == 0; bias / scale / shift / res (f32, ldres) / act as xp_gemm_nt.
 * xp_gemm_nt_h2s_applies(N, K): does xp_xpoint_forward route a layer of this shape here (a per-layer predicate: the producers of its input must know). */
size_t xp_p32_bytes(int64_t M, int K);
int xp_split_activations_h2(const float* x, void* out_p32, int64_t M, int K, int ldx, void* stream);
int xp_gemm_nt_h2s_applies(int N, int K);
int xp_gemm_nt_h2s(const void* A_p32, const void* Wh2, void* C, int out_fmt, const float* bias, const float* scale, const float* shift,
                   const float* res, int M, int N, int K, int ldc, int ldres, int act, void* stream);
/* LayerNorm (as xp_layernorm, gelu = 0) writing the P32 image of its (rows, C) result; C % 32 == 0. */
int xp_layernorm_p32(const float* x, void* y_p32, const float* w, const float* b, int64_t rows, int C, float eps, void* stream);
/* ---------------------------------------------------------------------------------------------
 * fp16-STORAGE dense kernels of the fast mixed-precision class (csrc/gemm_f16.hip): the reference's `mixed_precision: true` deployment
 * (xpoint/models/XPoint.py:182 autocast) with half tensors in HBM.  A (M, lda) and W (N, K) are fp16, K-contiguous; one exact fp16 product per multiply on
 * v_mfma_f32_32x32x16_f16, f32 accumulate;  C = r16( r16( act( r16(acc + bias) ) * scale + shift ) + res )  with r16 = round to nearest fp16 at every
 * point where autocast ends in a half tensor (absent terms drop out with their rounding); res (M, ldres) fp16; C fp16, or f32 holding the fp16-exact values
 * when c_f32 != 0 (the heads' `.to(torch.float)`, XPoint.py:349,363).  act as xp_gemm_nt.  K, lda multiples of 8; buffers 16-byte aligned. */
int xp_f32_to_f16(const float* x, void* y, int64_t n, void* stream);
int xp_gemm_nt_f16(const void* A, const void* W, void* C, int c_f32, const float* bias, const float* scale, const float* shift, const void* res,
                   int M, int N, int K, int lda, int ldc, int ldres, int act, void* stream);
/* 3x3 convolution (stride 1 | 2, zero or reflection pad 1) over NHWC halves as an implicit GEMM of the same kernel; W (Co, 3, 3, Ci) fp16; Ci % 8 == 0. */
int xp_conv3x3_nhwc_f16(const void* x, const void* W, void* y, int y_f32, const float* bias, const float* scale, const float* shift,
                        int batch, int Hi, int Wi, int Ci, int Co, int stride, int reflect_pad, int act, void* stream);

/* Fused MLP of a VSS block in that class (csrc/mlp_f16.hip; VMamba.py:110-128, :1230-1234):  x += fc2(GELU(fc1(a)))  with a = LayerNorm(x) (M, C) fp16 given,
 * x (M, C) fp16 updated in place, W1 (4C, C), W2 (C, 4C) fp16, biases f32; the hidden activation stays on chip; roundings as the two xp_gemm_nt_f16 calls it
 * replaces.  C in {32, 64, 96, 192} (xp_mlp_fused_f16_supported). */
int xp_mlp_fused_f16_supported(int C, int H4);
int xp_mlp_fused_f16(const void* a, void* x, const void* W1, const float* b1, const void* W2, const float* b2, int M, int C, int H4, void* stream);
/* ... with norm2 folded in (a = LayerNorm(x) computed in the kernel's prologue, rounded to fp16 like xp_layernorm_f16's output). */
int xp_ln_mlp_fused_f16(void* x, const float* ln_w, const float* ln_b, float eps, const void* W1, const float* b1, const void* W2, const float* b2,
                        int M, int C, int H4, void* stream);
/* norm + in_proj of a VSS block in one launch (VMamba.py:1225, :649):  y = LayerNorm(x) W^T,  x, y (M, C) fp16, W (C, C) fp16; C in {32, 64, 96, 192}. */
int xp_ln_proj_f16(const void* x, const float* ln_w, const float* ln_b, float eps, const void* W, void* y, int M, int C, void* stream);
/* Glue kernels of the same class (csrc/elementwise_f16.hip): half tensors in HBM, f32 arithmetic, one rounding per autocast boundary (= the store). */
int xp_stem_conv_ln_gelu_f16(const float* img, const float* w9co, const float* bias, const float* ln_w, const float* ln_b, void* y,
                             int batch, int H, int W, int CO, float eps, void* stream);
int xp_layernorm_f16(const void* x, void* y, const float* w, const float* b, int64_t rows, int C, float eps, void* stream);
int xp_dwconv3x3_silu_f16(const void* x, const float* w9c, void* y, float* y_f32_copy, int batch, int H, int W, int C, void* stream);
int xp_depth_to_space_nhwc_f16(const void* x, float* y_f32, void* y_f16, int batch, int H, int W, int C, int bs, int* status, void* stream);
/* Fused SS2D core on half tensors (see xp_ss2d_core_fwd): u, xdbl, out are fp16; scan state / softplus / exp / out_norm in f32 (csms6s.py:47-67,
 * VMamba.py:644-646); dt projection rounded to fp16 before the f32 bias.  u_f32 / xdbl_f32: optional f32 copies enabling the sequential deep-stage form
 * (xp_ss2d_core_f16_wants_f32_copies tells, from per-image quantities only, whether it would be taken). */
int xp_ss2d_core_f16_wants_f32_copies(int H, int W, int C, int R);
int xp_ss2d_core_fwd_f16(const void* u, const void* xdbl, const float* u_f32, const float* xdbl_f32, const float* wdt, const float* dt_bias,
                         const float* A, const float* Ds, const float* ln_w, const float* ln_b, void* out, float* workspace, size_t workspace_bytes,
                         int batch, int H, int W, int C, int R, int dstate, float eps, void* stream);
/* The whole forward in that class (csrc/model.cpp): `weights` = the blob whose autocast-cast tensors were rounded to fp16 (host), w16 = their fp16 copies
 * (xp_f16_weights_bytes / xp_prepare_f16_weights); workspace as xp_forward_workspace_bytes; outputs as xp_xpoint_forward_ex (f32, holding fp16-exact
 * encoder values; prob / desc computed in f32 from the half head outputs, XPoint.py:349,363). */
size_t xp_f16_weights_bytes(void* ctx);
int xp_prepare_f16_weights(void* ctx, const float* weights, void* w16, size_t w16_bytes, void* stream);
int xp_xpoint_forward_f16(void* ctx, const float* weights, const void* w16, const float* images, int batch, int H, int W, void* workspace,
                          size_t workspace_bytes, float* prob, float* desc_nhwc, float* enc_nhwc, float* logits_nhwc, int* status, void* stream);

/* Precision class of the "x3" kernels (xp_gemm_nt_x3, xp_conv3x3_nhwc_x3, xp_mlp_fused_x3 and every dense layer of
 * xp_xpoint_forward with wsplit != NULL), process-wide, read at launch time:
 *   6 (default)  all six partial products of weight >= 2^-16: f32-grade (the class pinned against the reference, 1e-4 bar)
 *   3            a0 b0 + a0 b1 + a1 b0: operands carried to 16 bits, error <= 3 * 2^-16 * sum|a||b| per dot product, half the matrix work
 *   1            a0 b0: bf16 operands, f32 accumulate — the arithmetic class of the reference's `mixed_precision` autocast
 *                (XPoint.py:182); activations, LayerNorm, scan and post-processing stay f32.
 * The same weight buffers serve all three. */
int xp_set_dense_products(int n);
int xp_get_dense_products(void);
/* Engine of the dense layers of xp_xpoint_forward (wsplit != NULL), process-wide, read at launch time:
 *   1 (default)  "h2": xp_gemm_nt_h2 / xp_conv3x3_nhwc_h2 — operands as two fp16 planes, three partial products (f32-grade)
 *   0            "x3": xp_gemm_nt_x3 / xp_conv3x3_nhwc_x3 — three bf16 planes, xp_set_dense_products partial products
 * The fused block kernels (xp_mlp_fused_x3, xp_ln_proj_x3) use the x3 planes under both; the buffer prepared by
 * xp_prepare_split_weights holds both formats. */
int xp_set_dense_engine(int engine);
int xp_get_dense_engine(void);
/* Per-launch engine override inside xp_xpoint_forward(_ex) (round 6; removes the range guard's cliff): with the split-fp16 engine selected, bit i
 * of `mask` sends dense launch i of the forward to the split-bf16 planes (x3: no operand-range limit) while every other launch stays on split fp16.
 * The host finds the launches whose operands leave the fp16 range by bisection over this mask (models.XPoint.handle_status) instead of moving
 * the whole weight set to x3.  Launch numbering (XP_DENSE_LAUNCHES = 47): 0 = patch-embed conv 2; block (stage s, index j), base = 1 + 5 (2 s + j):
 * base + 0 in_proj (with its LayerNorm when fused), + 1 x_proj, + 2 out_proj, + 3 fc1, + 4 fc2 (a fused block tail = one launch: any of + 2 .. + 4
 * sends all three); 41 + s = downsample conv after stage s; 44 = head trunk conv, 45 = detector 1x1, 46 = descriptor 1x1.  Process-wide, like the
 * engine; 0 (default) = no override.  Ignored by the mixed-precision classes. */
#define XP_DENSE_LAUNCHES 47
int xp_set_dense_override(unsigned long long mask);
unsigned long long xp_get_dense_override(void);
/* Mixed-precision class "amp16" — the arithmetic of the reference's `mixed_precision: true` deployment (XPoint.py:182: torch.cuda.amp.autocast
 * around the forward; half is autocast's default dtype): process-wide, read at launch time.
 *   1  every operation that autocast ends in a half tensor rounds its output to fp16 (round to nearest even; the values stay in f32 containers):
 *      the bias add, GELU, eval-BatchNorm affine and residual add of xp_gemm_nt_h2 / xp_conv3x3_nhwc_h2, xp_layernorm, both stages of
 *      xp_dwconv3x3_silu, the three stages of xp_stem_conv_ln_gelu (which also takes the image as half), the dt projection inside
 *      xp_ss2d_core_fwd (rounded before its f32 bias is added, csms6s.py:47-50) and the SS2D output (VMamba.py:646 `y.to(x.dtype)`); the scan,
 *      out_norm, softmax and normalize stay f32, as the reference's do (csms6s.py:52, XPoint.py:349,363).  xp_xpoint_forward(_ex) then runs
 *      every block as separate launches on the split-fp16 engine; the caller passes weights whose convolution / linear tensors were rounded to
 *      fp16 (autocast casts them; models.XPoint does this for gemm_mode "amp16"), so every product is one exact fp16 x fp16 MFMA product.
 *   0  (default) off.
 * Pinned against the real reference run under fp16 CPU autocast (tests/golden/g20); never the headline class. */
int xp_set_amp_mode(int mode);
int xp_get_amp_mode(void);
/* y[i] = (float)(half)x[i] (round to nearest even), n floats, 16-byte aligned buffers; in place allowed.  The `.half()` of the class above. */
int xp_round_f16(const float* x, float* y, int64_t n, void* stream);

/* Fused VSS-block MLP branch, in place:  X <- X + fc2(GELU(fc1(LayerNorm(X)) + b1)) + b2   (reference
 * VMamba.py:1230-1234 VSSBlock.forward second residual, :110-128 Mlp; LayerNorm over C, biased variance, eps;
 * exact-erf GELU).  One launch replaces xp_layernorm + two xp_gemm_nt_x3 calls; the (M, hidden) activation is never
 * written to memory (it goes from the fc1 accumulators to the fc2 operand registers).  Same split-bf16 arithmetic
 * as xp_gemm_nt_x3.
 * Optionally the block's first residual is folded in as well: with T1 != NULL the kernel first does
 * X <- X + T1 W0^T  (VMamba.py:663 SS2D out_proj, bias-free, and :1229 x = x + op(norm(x))), T1 (M, C) not aliasing X.
 * The weight matrices are passed as ONE packed stream in the order the kernel consumes them: xp_mlp_fused_x3_pack
 * converts fc1.weight (hidden, C), fc2.weight (C, hidden) and, if W0x3 != NULL, W0 (C, C) — all already in
 * xp_split_weights_x3 layout — into xp_mlp_fused_x3_pack_bytes(C, hidden, W0x3 != NULL) bytes, once per weight upload.
 * A stream packed with W0 must be used with T1 != NULL and vice versa.
 * Supported shapes: xp_mlp_fused_x3_supported(C, hidden) != 0 (C in {32, 64, 96, 128, 192}, hidden % 32 == 0,
 * 64 <= hidden <= 4096); other shapes return an argument error — callers use the separate entry points. */
int xp_mlp_fused_x3_supported(int C, int hidden);
size_t xp_mlp_fused_x3_pack_bytes(int C, int hidden, int with_proj);
int xp_mlp_fused_x3_pack(const void* W1x3, const void* W2x3, const void* W0x3, void* out, int C, int hidden, void* stream);
int xp_mlp_fused_x3(float* X, const float* T1, const float* ln_w, const float* ln_b, const void* Wpack, const float* b1,
                    const float* b2, int M, int C, int hidden, float eps, void* stream);
/* The same fused kernels on the split-fp16 engine ("h2", csrc/gemm_h2_core.h: two fp16 planes, three products): weight
 * streams packed from xp_split_weights_h2 buffers (W1h2 (hidden, C), W2h2 (C, hidden), W0h2 (C, C) or NULL), whose power-of-two
 * row scales the kernel undoes exactly (fc1: before the GELU; fc2 and the projections: on the accumulators); the same buffers
 * are passed again at launch for those scales.  Shapes as xp_mlp_fused_x3_supported. */
size_t xp_mlp_fused_h2_pack_bytes(int C, int hidden, int with_proj);
int xp_mlp_fused_h2_pack(const void* W1h2, const void* W2h2, const void* W0h2, void* out, int C, int hidden, void* stream);
int xp_mlp_fused_h2(float* X, const float* T1, const float* ln_w, const float* ln_b, const void* Wpack, const void* W1h2, const void* W2h2,
                    const void* W0h2, const float* b1, const float* b2, int M, int C, int hidden, float eps, void* stream);
size_t xp_ln_proj_h2_pack_bytes(int C, int N);
int xp_ln_proj_h2_pack(const void* W0h2, void* out, int C, int N, void* stream);
int xp_ln_proj_h2(const float* X, const float* ln_w, const float* ln_b, const void* Wpack, const void* W0h2, float* Out, int M, int C, int N,
                  float eps, void* stream);
/* The head of the block in the same row-stationary form:  Out (M, N) = LayerNorm(X) W0^T   (VMamba.py:1229 norm + :649
 * in_proj, bias-free), one launch instead of xp_layernorm + xp_gemm_nt_x3, the normalised rows never written to memory.
 * W0 (N, C) is passed as the packed stream built by xp_ln_proj_x3_pack from its xp_split_weights_x3 form
 * (xp_ln_proj_x3_pack_bytes(C, N) bytes; 0 = unsupported: C in {32, 64, 96, 128, 192}, N % 32 == 0).  Out must not alias X. */
size_t xp_ln_proj_x3_pack_bytes(int C, int N);
int xp_ln_proj_x3_pack(const void* W0x3, void* out, int C, int N, void* stream);
int xp_ln_proj_x3(const float* X, const float* ln_w, const float* ln_b, const void* Wpack, float* Out, int M, int C, int N,
                  float eps, void* stream);

/* Glue kernels (HBM-bound). */
int xp_layernorm(const float* x, float* y, const float* w, const float* b, int64_t rows, int C, float eps, int gelu,
                 void* stream);
int xp_dwconv3x3_silu(const float* x, const float* w9c, float* y, int batch, int H, int W, int C, void* stream);
int xp_stem_conv_ln_gelu(const float* img, const float* w9co, const float* bias, const float* ln_w, const float* ln_b,
                         float* y, int batch, int H, int W, int Co, float eps, void* stream);
int xp_depth_to_space_nhwc(const float* x, float* y, int batch, int H, int W, int C, int bs, void* stream);
int xp_softmax_shuffle(const float* logits, float* prob, int batch, int Hc, int Wc, int r, int ld, int mode, void* stream);
int xp_l2norm_rows(const float* x, float* y, int64_t rows, int C, float eps, void* stream);
int xp_nhwc_to_nchw(const float* x, float* y, int batch, int HW, int C, void* stream);
int xp_mul_mask(const float* x, const uint8_t* mask, float* y, int64_t n, void* stream);
/* Batch staging of a step's inputs in ONE launch: images[0 .. n) = optical, images[n .. 2n) = thermal (f32, n = pairs*H*W elements)
   and, when the masks are given (all three or none), masks[0 .. n) = mask_optical, masks[n .. 2n) = mask_thermal (u8).  Replaces the four
   device-to-device copies of the reference's batch assembly (predict_align_image_pair.py:185-190 stacks the pair into one batch;
   ImagePairDataset collate): the runtime's blit kernel moves a 9.8 MB image block at ~130 GB/s (76 us), this kernel at HBM speed. */
int xp_stage_pair_batch(const float* optical, const float* thermal, float* images, const uint8_t* mask_optical,
                        const uint8_t* mask_thermal, uint8_t* masks, int64_t n, void* stream);
int xp_maxpool2_nhwc(const float* x, float* y, int batch, int H, int W, int C, void* stream);
/* RegNet's cost volume followed by its global average pool (reference xpoint/models/RegNet.py:44-52: bmm(x1^T, x2) -> view(N, hw, H', W') ->
   adaptive_avg_pool2d(., 1)), without materialising the (hw, hw) volume:  v[n][p] = a[n][p] . mean_q b[n][q];  a, b (batch, hw, C) row-major
   (channel-normalised feature rows of the two images), v (batch, hw). */
int xp_costvolume_mean(const float* a, const float* b, float* v, int batch, int hw, int C, void* stream);
/* Data ingest (reference datasets/ImagePairDataset.py:199-208 cv2.imread + COLOR_BGR2GRAY + / 255.0, :254-274 crop):
 * src = decoded 8-bit image on the device, (H0, W0, channels) interleaved with channels 1 (gray), 3 (R,G,B) or 4 (R,G,B,A);
 * dst (h, w) f32 = lut256[gray] of the crop at (top, left), gray = (B*1868 + G*9617 + R*4899 + 8192) >> 14 (OpenCV's 8-bit
 * fixed-point BGR2GRAY), lut256[k] = float32(k / 255.0) supplied by the host (double-precision division as numpy's). */
int xp_ingest_u8(const uint8_t* src, int H0, int W0, int channels, int top, int left, int h, int w,
                 const float* lut256, float* dst, void* stream);
/* dst[i] = (float)src[i] / 255 for a batch of 8-bit gray images already cropped (n elements): the device half of an 8-bit upload — the reference loader's
 * `astype(float32) / 255` (xpoint/datasets/ImagePairDataset.py:254-274), same IEEE division, same bits.  src 4-byte, dst 16-byte aligned. */
int xp_u8_to_unit_f32(const uint8_t* src, float* dst, int64_t n, void* stream);
/* Copy `bytes` from device memory to PINNED (device-mapped: hipHostMalloc / torch pin_memory) host memory with a kernel on `stream` instead of a copy
 * engine: small result lists of a streaming step, so that they never queue in front of the next step's image upload (predict.PairPipeline.download_async).
 * Valid on the host once an event recorded on `stream` after the call has completed.  16-byte aligned pointers. */
int xp_copy_to_mapped_host(const void* src_dev, void* dst_host, size_t bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Model forward.  Replaces xpoint.models.XPoint.forward_impl (xpoint/models/XPoint.py:283-323) with the
 * VMamba encoder (vmamba_src/VMamba.py:1507-1525) and the detector / descriptor heads (XPoint.py:348-371).
 * Device-format parameters live in ONE caller-owned float blob; xp_param_info enumerates
 * (name, offset, numel) so the host can fill it from a reference state_dict (xpoint_amd/models.py). */
typedef struct xp_model_cfg {
    int embed_dim;       /* MODEL.VSSM.EMBED_DIM (96) */
    int n_stages;        /* len(DEPTHS) */
    int depths[4];
    int d_state;         /* SSM_D_STATE (1) */
    int dt_rank;         /* 0 = "auto" = ceil(dim/16) */
    float mlp_ratio;     /* 4.0 */
    int head_channels;   /* 256 */
    int desc_size;       /* descriptor_size (256) */
    int det_channels;    /* 65 */
} xp_model_cfg;

int xp_ctx_create(const xp_model_cfg* cfg, void** ctx);
int xp_ctx_destroy(void* ctx);
int xp_param_count(void* ctx);
size_t xp_weights_numel(void* ctx);
int xp_param_info(void* ctx, int index, char* name, int name_len, size_t* offset, size_t* numel);
int xp_forward_shapes(void* ctx, int batch, int H, int W, int* Hc, int* Wc, int* enc_channels);
size_t xp_forward_workspace_bytes(void* ctx, int batch, int H, int W);
/* Split-bf16 copies of the GEMM / conv weights (xp_split_weights_x3 layout) for the fp32-accurate bf16-matrix-pipe
 * kernels: a second caller-owned device buffer of xp_split_weights_bytes(ctx) bytes, derived from the f32 blob by
 * xp_prepare_split_weights after every weight upload (the f32 blob stays the single source of truth, e.g. for the
 * RCCL broadcast). */
size_t xp_split_weights_bytes(void* ctx);
int xp_prepare_split_weights(void* ctx, const float* weights, void* wsplit, size_t wsplit_bytes, void* stream);
/* images (batch,1,H,W) in [0,1]; outputs: prob (batch,H,W) or NULL; desc_nhwc (batch,Hc,Wc,desc_size) or NULL;
 * enc_nhwc (batch,Hc,Wc,embed_dim/2) required; logits_nhwc (batch,Hc,Wc,65) or NULL.
 * wsplit: the buffer prepared by xp_prepare_split_weights -> dense layers run on xp_gemm_nt_x3 / xp_conv3x3_nhwc_x3;
 * NULL -> they run on the exact-f32 MFMA kernels (xp_gemm_nt / xp_conv3x3_nhwc).  Same results to f32 rounding.
 * With the split-fp16 engine selected (xp_set_dense_engine(1), the default of the Python host) every dense-layer INPUT must stay
 * below 65504 in magnitude; beyond it that layer's output rows turn into NaN, and because the heads apply ReLU (max(NaN, 0) = 0 on the
 * GPU) `prob` may then look finite.  xp_xpoint_forward_ex reports it: `status` (device int, caller-zeroed, may be NULL) receives, OR-ed in by
 * the kernels that write the outputs anyway (no extra pass, no host synchronisation):
 *   XP_STATUS_ENC  (1)  the encoder output (= the residual stream every dense layer of the encoder writes into, and the operand of the head
 *                       convolution) holds a non-finite element, or — split-fp16 engine only — one with |x| >= 65504
 *   XP_STATUS_PROB (2)  a heat-map cell had a non-finite logit        XP_STATUS_DESC (4)  a descriptor row had a non-finite element
 * A caller that reads a non-zero status re-runs the forward on engine 0 (split-bf16: no range limit) — the Python host does so
 * automatically and stays on that engine for the weight set (models.XPoint, PairPipeline.verify). */
int xp_xpoint_forward(void* ctx, const float* weights, const void* wsplit, const float* images, int batch, int H, int W,
                      void* workspace, size_t workspace_bytes, float* prob, float* desc_nhwc, float* enc_nhwc,
                      float* logits_nhwc, void* stream);
#define XP_STATUS_ENC 1
#define XP_STATUS_PROB 2
#define XP_STATUS_DESC 4
int xp_xpoint_forward_ex(void* ctx, const float* weights, const void* wsplit, const float* images, int batch, int H, int W,
                         void* workspace, size_t workspace_bytes, float* prob, float* desc_nhwc, float* enc_nhwc,
                         float* logits_nhwc, int* status, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Post-processing.
 * xp_box_nms replaces xpoint.utils.box_nms (xpoint/utils/utils.py:148-192; torchvision.ops.nms /
 * batched_nms): prob (batch,H,W) -> out (batch,H,W) with surviving scores, zeros elsewhere.
 *   size: box side (multiple of 0.5, <= 16); keep_top_k > 0 keeps the k best survivors per image (needs cap >=
 *   survivors).  Exact greedy NMS as a fixed point (round 3): a local-maximum pass, one suppression round and an in-launch finisher per
 *   image (three launches, NO sweep count) whenever an image fits the finisher's LDS as one band (480 x 640 and smaller); larger images
 *   (1024 x 1024) are finished band by band and the finisher launch is repeated — launches past convergence exit at once.
 *   max_sweeps_async == 0: synchronous like the reference (returns a finished tensor; banded images: XP_ERR_STATE if bands are still
 *   undecided after all counter slots); > 0: stream-ordered, at most that many finisher launches for banded images, no host
 *   synchronisation — verify later with xp_box_nms_check (0 undecided = exact result; one-band images always are).  Box sizes the
 *   fixed-point form does not cover fall back to tile sweeps with the same two modes. */
size_t xp_box_nms_workspace_bytes(int batch, int H, int W, int cap);
int xp_box_nms(const float* prob, float* out, void* workspace, size_t workspace_bytes, int batch, int H, int W,
               float size, float min_prob, float iou, int keep_top_k, int cap, int max_sweeps_async,
               int* converged_host, void* stream);
int xp_box_nms_check(const void* workspace, int batch, int H, int W, int* undecided_tiles, void* stream);

/* torch.nonzero((prob > thr) [* mask]) per image (predict_align_image_pair.py:242-243, predict_keypoints.py:213-215):
 * kp (batch, cap, 2) int32 (y, x) in row-major order, counts (batch) int32 (may exceed cap: truncated list). */
size_t xp_extract_keypoints_workspace_bytes(int batch, int H, int W);
int xp_extract_keypoints(const float* prob, const uint8_t* mask, float thr, int* kp, int* counts, int batch, int H,
                         int W, int cap, void* workspace, size_t workspace_bytes, void* stream);

/* xpoint.utils.interpolate_descriptors (utils.py:229-238): bilinear grid_sample (align_corners=True) of the
 * NHWC descriptor volume (batch,Hc,Wc,D) at the keypoints + L2 normalisation -> out (batch, cap, D). */
int xp_sample_descriptors(const int* kp, const int* counts, const float* desc_nhwc, float* out, int batch, int cap,
                          int Hc, int Wc, int D, int H, int W, void* stream);

/* xpoint.utils.get_matches(d1, d2, 'bfmatcher', False, crossCheck=True) (xpoint/utils/matching.py:4-36; OpenCV
 * BFMatcher NORM_L2) for `pairs` independent pairs.  d1 (pairs,cap1,D), d2 (pairs,cap2,D); the number of valid
 * rows of pair i is counts[i*cnt_stride + which1] / [.. + which2] (counts NULL: all cap rows).
 * mode 0 = strict mutual nearest neighbour (primary), 1 = legacy cross-check.  Outputs: idx12/dist12 (pairs,cap1),
 * idx21/dist21 (pairs,cap2), matches (pairs,cap1) as (queryIdx, trainIdx, distance) ascending in queryIdx,
 * match_count (pairs).  Index results are those of exact arithmetic (first index wins exact ties): the fp16 matrix pass only nominates, the
 * nominated pairs are re-evaluated in f32 direct form and the survivors of that window in fp64 (csrc/match.hip). */
size_t xp_match_workspace_bytes(int pairs, int cap1, int cap2, int D);
int xp_match_mnn(const float* d1, const float* d2, const int* counts, int cnt_stride, int which1, int which2, int pairs,
                 int cap1, int cap2, int D, int mode, int* idx12, float* dist12, int* idx21, float* dist21,
                 int* match_q, int* match_t, float* match_d, int* match_count, void* workspace, size_t workspace_bytes,
                 void* stream);
/* `get_matches(..., knn_matches=True)` (matching.py:20-27): cv2.BFMatcher(NORM_L2).knnMatch(d1, d2, k = 2) — the two nearest targets of every query in
 * exact arithmetic (ties -> lower index); the caller applies Lowe's ratio test.  idx2 / dist2 (pairs, cap1, 2); a pair with fewer than two targets gets
 * -1 / +inf in the missing slots.  Same workspace as xp_match_mnn. */
int xp_match_knn2(const float* d1, const float* d2, const int* counts, int cnt_stride, int which1, int which2, int pairs, int cap1, int cap2,
                  int D, int* idx2, float* dist2, void* workspace, size_t workspace_bytes, void* stream);
/* `ThresholdMatcher.match` (matching.py:77-102): every (q, t) with sqrt(2 - 2 clip(<a_q, b_t>, -1, 1)) < threshold, decided in fp64 (the matrix pass only
 * nominates).  out_pairs (out_cap, 3) = (pair, query, target) in NO particular order (the reference lists them row-major: the host wrapper sorts),
 * out_dist (out_cap); out_count: EIGHT ints on the device, [0] = accepted pairs (> out_cap: list truncated), [1] = nominated pairs (> hit_cap: repeat with a
 * larger `hits` scratch of hit_cap x 2 ints — accepted pairs may be missing).  0 <= threshold <= 2. */
int xp_match_threshold(const float* d1, const float* d2, const int* counts, int cnt_stride, int which1, int which2, int pairs, int cap1, int cap2,
                       int D, double threshold, int* hits, int hit_cap, int* out_pairs, float* out_dist, int* out_count, int out_cap,
                       void* workspace, size_t workspace_bytes, void* stream);
/* Candidate-list statistics of the latest xp_match_mnn call on `workspace` (same pairs / cap1 / cap2 / D / counts): out4 = 4 x uint64 on the device =
 * [sum of the nomination-list lengths over live rows and columns, the longest list, rows + columns whose list overflowed the inline capacity
 * (xp_match_cand_cap(); finished by the parallel overflow pass), live rows + columns].  bench.py reports them as match_candidates_per_row /
 * match_overflow_rows: the matcher's run time depends on them (clustered descriptors nominate more), its result never does. */
int xp_match_stats(void* workspace, const int* counts, int cnt_stride, int which1, int which2, int pairs, int cap1, int cap2, int D,
                   unsigned long long* out4, void* stream);
int xp_match_cand_cap(void);

/* ---------------------------------------------------------------------------------------------
 * Evaluation-harness helper: out[i] = min_j |a_i - b_j| (Euclidean, 2-D points (y, x)); +inf when
 * nb == 0.  a is f64 (warped keypoints), b f32 (pixel keypoints): the difference is formed in f64 and cast to f32, the
 * norm is taken in f32 — the arithmetic of benchmark_evaluation.py:441-450 (repeatability) and :652-659 (correct-match
 * matrix, reduced over one axis) without materialising the N x M matrix. */
int xp_points_min_dist(const double* a, int na, const float* b, int nb, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Robust homography from point correspondences, batched over pairs: the device-side stand-in
 * for cv2.findHomography(src, dst, cv2.USAC_MAGSAC, ransacReprojThreshold, confidence, maxIters) as called from
 * predict_align_image_pair.py:291-303 and benchmark_evaluation.py:796-812.  Same contract (H maps src -> dst, h33 = 1;
 * inlier mask at the reprojection threshold; no model below 4 correspondences -> n_inliers = 0, H = identity), a
 * different — deterministic — estimator (hash-seeded 4-point DLT hypotheses, MSAC score, least-squares refinement):
 * not bit-comparable with OpenCV.
 *   src, dst (pairs, cap, 2) f32 (x, y); counts (pairs) int32 or NULL; H (pairs, 9) f64; mask (pairs, cap) u8. */
/* Correspondences of the mutual matches for xp_find_homography (counts = match_count): kp (2*pairs, cap, 2) int32 (y, x),
 * optical images first; src / dst (pairs, cap, 2) f32 (x, y) — the optical_pts / thermal_pts lists of
 * predict_align_image_pair.py:283-284 built on the device. */
int xp_gather_match_points(const int* kp, const int* match_q, const int* match_t, const int* match_count, int pairs, int cap,
                           float* src, float* dst, void* stream);
size_t xp_find_homography_workspace_bytes(int pairs);
int xp_find_homography(const float* src, const float* dst, const int* counts, int pairs, int cap, float reproj_thr,
                       int max_iters, unsigned seed, double* H, uint8_t* mask, int* n_inliers, void* workspace,
                       size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Perspective warp — the reference's registration output:
 *     cv2.warpPerspective(im_optical, H_est, im_optical.shape[:2][::-1], borderMode=cv2.BORDER_CONSTANT)
 * (predict_align_image_pair.py:308, demo.py:225-249): INTER_LINEAR, border value 0, M (batch, 9) f64 ON THE DEVICE = the forward
 * map src -> dst (what xp_find_homography writes), inverted on the device by the closed 3 x 3 form unless inverse_map != 0
 * (cv2.WARP_INVERSE_MAP).  Arithmetic = OpenCV's documented fixed-point scheme (source coordinates rounded to 1/32 pixel in double,
 * u8: 15-bit integer weights; f32: the same 1/32 fractions as float weights) — OpenCV is absent here: parity unpinned, the oracle
 * (oracle/csrc/oracle_kernels.c: xo_warp_perspective_u8 / _f32) states the same scheme and the GPU tests demand equality with it.
 *   src (batch, Hs, Ws, channels), dst (batch, Hd, Wd, dst_channels), channel-interleaved; dst_channels == channels, or a
 *   1-channel source replicated into dst_channels (the reference warps cv2.cvtColor(gray, COLOR_GRAY2RGB)).
 *   dtype XP_WARP_U8: u8 -> u8;  XP_WARP_F32: f32 -> f32;  XP_WARP_F32_AS_U8: f32 source in [0, 1] quantised on load as
 *   (np.clip(img, 0, 1) * 255.0).astype(np.uint8) (predict_align_image_pair.py:271) -> u8.  Hs, Ws < 32768. */
#define XP_WARP_U8 0
#define XP_WARP_F32 1
#define XP_WARP_F32_AS_U8 2
int xp_warp_perspective(const void* src, void* dst, const double* M, int batch, int Hs, int Ws, int Hd, int Wd, int channels,
                        int dst_channels, int dtype, int inverse_map, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline leg; replaces the
 * reference's wall-clock brackets, benchmark_evaluation.py:12-37).  Off by default.  xp_prof_filter(tag)
 * restricts recording to one kernel tag (NULL/"" = all).  xp_prof_count / xp_prof_get synchronise on the
 * recorded events and return, per tag: total ms, launches, algorithmic FLOPs and bytes of those launches. */
int xp_prof_enable(int on);
int xp_prof_filter(const char* tag);
int xp_prof_reset(void);
int xp_prof_count(void);
int xp_prof_get(int index, char* tag, int tag_len, double* total_ms, int* launches, double* flops, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* XPOINT_HIP_H */
