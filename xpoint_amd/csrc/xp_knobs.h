// The ONE registry of the library's XP_* environment knobs ("30+ knobs read through function-local statics").  Every knob is a tuning / A-B
// switch: the defaults are the measured best and no knob changes a result's precision class (those are API calls: xp_set_dense_engine, xp_set_dense_products,
// xp_set_amp_mode).  Each is read ONCE per process, at the first call that needs it.  tests/test_cpu_host.py::test_knob_registry_is_complete greps the sources
// for getenv("XP_...") / os.environ["XP_..."] and fails when a knob is missing here (or listed here but read nowhere).  xp_knob_count / xp_knob_info enumerate it.
#pragma once

struct XpKnob { const char* name; const char* where; const char* what; };

static const XpKnob kXpKnobs[] = {
    // ---- dense engines
    {"XP_DENSE_ENGINE", "api.cpp", "initial value of xp_set_dense_engine: h2 (default) | x3"},
    {"XP_DENSE_PRODUCTS", "api.cpp", "initial value of xp_set_dense_products: 6 (default) | 3 | 1"},
    {"XP_RING", "gemm_ring.hip", "0: keep every layer on the round-4 GEMM kernels (default 1: ring engine for N, K >= 256 layers of the split class)"},
    {"XP_RING_TILE", "gemm_ring.hip", "force the ring tile: 0 = 256x256, 1 = 256x128, 2 = 128x128"},
    {"XP_H2P", "gemm_h2p.hip", "ping-pong split-fp16 GEMM: 0 off, 1 (default) K >= 768 and N >= 384, 2 widened"},
    {"XP_H2_ENGINE", "gemm_h2.hip", "rs | lds: force the row-stationary / tile variant of the split-fp16 GEMM"},
    {"XP_H2_TILE", "gemm_h2.hip", "force the split-fp16 tile (0..4)"},
    {"XP_H2_NO64", "gemm_h2.hip", "64x128 tile rule: 0 old, 1 never, 2 (default) only below 128 tiles"},
    {"XP_H2_NGROUP", "gemm_h2.hip", "column tiles per group of the tile order (0 = N-fastest)"},
    {"XP_X3_TILE", "gemm_x3.hip", "force the split-bf16 tile"},
    {"XP_GEMM_STAGGER", "gemm.hip", "exact-f32 GEMM: start-up delay (cycles) of the second workgroup slot of every CU"},
    {"XP_GEMM_STAMPS", "gemm.hip", "exact-f32 GEMM: s_memtime stamps per workgroup (debug)"},
    {"XP_F16_TILE", "gemm_f16.hip", "force the fp16-class tile (0..5)"},
    {"XP_F16_BK", "gemm_f16.hip", "force the fp16-class slab depth: 32 | 64"},
    // ---- fused block kernels
    {"XP_NO_FUSED_MLP", "model.cpp", "1: stages 0-1 as separate launches instead of xp_ln_proj + xp_mlp_fused"},
    {"XP_FUSE_MAXC", "model.cpp", "blocks wider than this many channels run unfused"},
    {"XP_FUSED_X3", "model.cpp", "1: fused block kernels on the x3 planes under the h2 engine"},
    {"XP_NO_LN_PROJ_F16", "model.cpp", "fp16 class: LayerNorm and in_proj as two launches"},
    {"XP_NO_FUSED_MLP_F16", "model.cpp", "fp16 class: 1 two-GEMM MLP everywhere, 2 fused MLP behind a separate LayerNorm"},
    {"XP_MLP_TAIL", "mlp_fused.hip", "0: no separate 4-wave launch for a last round less than half full (C = 192)"},
    {"XP_MLP_NW8", "mlp_fused.hip", "1: 8-wave workgroups in the x3 fused tail (C <= 96)"},
    {"XP_MLP_H2_NW4", "mlp_fused.hip", "1: 4-wave workgroups in the h2 fused tail at C = 192"},
    // ---- scans
    {"XP_SS2D_SEQ", "ss2d.hip", "0 / 1: force the chunked / sequential form of the fused SS2D core (also xp_ss2d_core_set_mode)"},
    {"XP_SS2D_SEQ_MAXL", "ss2d.hip", "L bound of the automatic choice of the sequential form (default 8192)"},
    {"XP_SS2D_SEQ_V1", "ss2d.hip", "1: first sequential kernel (dt projection on the vector ALU)"},
    {"XP_SS2D_SEQ_NW", "ss2d.hip", "waves per route of the pipelined sequential kernel: 1 | 2 | 4"},
    {"XP_SS2D_TBUDGET", "ss2d.hip", "chunk-length budget of the chunked passes (default 6144 pixel-channels)"},
    {"XP_SS2D_T", "ss2d.hip", "chunk length per channel count, e.g. \"96:32,192:16,384:16\""},
    {"XP_SS2D_THREADS", "ss2d.hip", "threads per workgroup of the chunked passes (default 192)"},
    {"XP_SCAN_V1", "selective_scan.hip", "force the first d_state = 1 operator-boundary scan kernel"},
    {"XP_SCAN_V2", "selective_scan.hip", "force the second d_state = 1 operator-boundary scan kernel"},
    {"XP_SCAN_OLD_GEN", "selective_scan.hip", "1: the round-1 generic-N operator-boundary scan kernel"},
    // ---- glue / post-processing
    {"XP_NMS_SCHED", "postproc.hip", "NMS local-iteration schedule"},
    {"XP_NMS_SWEEP", "postproc.hip", "NMS sweep count"},
    {"XP_NMS_WIDE_ROUNDS", "postproc.hip", "wide suppression rounds ahead of the NMS finisher"},
    // ---- profiling
    {"XP_PROF_SHAPES", "*.hip", "1: per-shape tags in the HIP-event breakdown (xp_prof_*)"},
    // ---- host side (Python)
    {"XP_GEMM_MODE", "models.py / bench.py", "default gemm_mode of models.XPoint: h2 | x3 | f32 | x2 | bf16 | amp16 | amp16f"},
    {"XP_HONOR_MIXED_PRECISION", "models.py", "1: take the precision class from the config's mixed_precision flag"},
    {"XP_LIB_PATH", "_lib.py", "load another build of libxpoint_hip.so"},
    {"XP_EXTRA_HIPCC_FLAGS", "build.py", "extra hipcc flags of `python -m xpoint_amd.build`"},
    {"XP_D2H_COPY_ENGINE", "predict.py", "1: result lists to the host through the runtime's copy engines instead of xp_copy_to_mapped_host (A/B)"},
    {"XP_C5_HEAD_STREAM", "streaming.py", "caller: the RegNet head of the streaming step on the caller's stream (round-4 placement) instead of the detection / matching stream (A/B)"},
    {"XP_RANK_CPUS", "affinity.py", "pin every rank to this cpulist instead of its GPU's NUMA CPUs"},
    {"XP_CPU_THREADS", "bench.py", "threads of the CPU baseline"},
    {"XP_BENCH_DEPTH", "bench.py", "steps in flight of the alternating-encoder schedule (default 3)"},
    {"XP_BENCH_NO_PIN", "bench.py", "1: do not pin the rank to its GPU's NUMA CPUs"},
    {"XP_BENCH_NO_RCCL", "bench.py", "1: world 1 without a live RCCL communicator"},
    {"XP_BENCH_REHEARSE_ON_ONE_GPU", "bench.py", "1: --gpus N on a 1-GPU box: the N ranks share GPU 0, collectives over gloo (code-path rehearsal of the N-rank line; rates meaningless)"},
    {"XP_BENCH_PCIE_PARTS", "bench.py", "1: print the parts of the streaming loop (profiles/r5_streaming_parts.txt)"},
};
static const int kXpKnobCount = (int)(sizeof(kXpKnobs) / sizeof(kXpKnobs[0]));
