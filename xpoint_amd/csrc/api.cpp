// Error plumbing + version for the C ABI (include/xpoint_hip.h).
#include <stdarg.h>
#include <string.h>

#include "xp_common.h"
#include "xp_knobs.h"

static thread_local char g_err[1024] = "";

void xp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* xp_last_error(void) { return g_err; }
extern "C" int xp_version(void) { return 100; }  // 0.1.0

// the XP_* environment knob registry (xp_knobs.h)
extern "C" int xp_knob_count(void) { return kXpKnobCount; }
extern "C" int xp_knob_info(int index, const char** name, const char** where, const char** what) {
    if (index < 0 || index >= kXpKnobCount || !name || !where || !what) { xp_set_error("xp_knob_info: bad arguments"); return XP_ERR_ARG; }
    *name = kXpKnobs[index].name; *where = kXpKnobs[index].where; *what = kXpKnobs[index].what;
    return XP_OK;
}

extern "C" int xp_device_info(int device, int* cu_count, int* wave_size, char* arch, int arch_len) {
    hipDeviceProp_t p;
    XP_HIP(hipGetDeviceProperties(&p, device));
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (wave_size) *wave_size = p.warpSize;
    if (arch && arch_len > 0) {
        strncpy(arch, p.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return XP_OK;
}

// ------------------------------------------------------------------------------------------------
// In-library kernel timing with HIP events, recorded on the stream each kernel is launched on
// (bench.py's roofline leg).  Off by default: when disabled a scope costs one branch.
// ------------------------------------------------------------------------------------------------
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct ProfRec { std::string tag; double flops, bytes; hipEvent_t e0, e1; };
struct ProfAcc { double ms = 0, flops = 0, bytes = 0; int count = 0; };
std::mutex g_pm;
bool g_prof_on = false;
std::string g_prof_filter;
std::vector<ProfRec> g_pending;
std::vector<hipEvent_t> g_free_events;
std::map<std::string, ProfAcc> g_acc;

hipEvent_t get_event() {
    if (!g_free_events.empty()) { hipEvent_t e = g_free_events.back(); g_free_events.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
void collect_locked() {
    for (auto& r : g_pending) {
        (void)hipEventSynchronize(r.e1);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            ProfAcc& a = g_acc[r.tag];
            a.ms += ms; a.flops += r.flops; a.bytes += r.bytes; a.count += 1;
        }
        g_free_events.push_back(r.e0); g_free_events.push_back(r.e1);
    }
    g_pending.clear();
}
}  // namespace

XpProfScope::XpProfScope(const char* tag, hipStream_t s, double flops, double bytes) : active_(false), stream_(s) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_pm);
    if (!g_prof_filter.empty() && g_prof_filter != tag) return;
    active_ = true;
    ProfRec r{tag, flops, bytes, get_event(), get_event()};
    (void)hipEventRecord(r.e0, s);
    g_pending.push_back(r);
    index_ = g_pending.size() - 1;
}
XpProfScope::~XpProfScope() {
    if (!active_) return;
    std::lock_guard<std::mutex> lk(g_pm);
    if (index_ < g_pending.size()) (void)hipEventRecord(g_pending[index_].e1, stream_);
}

extern "C" int xp_prof_enable(int on) { std::lock_guard<std::mutex> lk(g_pm); g_prof_on = on != 0; return XP_OK; }
extern "C" int xp_prof_filter(const char* tag) { std::lock_guard<std::mutex> lk(g_pm); g_prof_filter = tag ? tag : ""; return XP_OK; }
extern "C" int xp_prof_reset(void) { std::lock_guard<std::mutex> lk(g_pm); collect_locked(); g_acc.clear(); return XP_OK; }
extern "C" int xp_prof_count(void) { std::lock_guard<std::mutex> lk(g_pm); collect_locked(); return (int)g_acc.size(); }
extern "C" int xp_prof_get(int index, char* tag, int tag_len, double* total_ms, int* launches, double* flops, double* bytes) {
    std::lock_guard<std::mutex> lk(g_pm);
    collect_locked();
    if (index < 0 || index >= (int)g_acc.size()) { xp_set_error("xp_prof_get: index out of range"); return XP_ERR_ARG; }
    auto it = g_acc.begin(); std::advance(it, index);
    if (tag && tag_len > 0) { strncpy(tag, it->first.c_str(), tag_len - 1); tag[tag_len - 1] = 0; }
    if (total_ms) *total_ms = it->second.ms; if (launches) *launches = it->second.count;
    if (flops) *flops = it->second.flops; if (bytes) *bytes = it->second.bytes;
    return XP_OK;
}

// ---- precision class of the split-bf16 dense kernels -------------------------------------------------------------------
#include <atomic>
static std::atomic<int> g_dense_products{[] { const char* e = getenv("XP_DENSE_PRODUCTS"); const int v = e ? atoi(e) : 6; return (v == 1 || v == 3) ? v : 6; }()};
int xp_dense_products_value() { return g_dense_products.load(); }
extern "C" int xp_get_dense_products(void) { return g_dense_products.load(); }
// Dense-layer engine of xp_xpoint_forward with wsplit != NULL: 1 = "h2" (two fp16 planes, three products; gemm_h2_core.h),
// 0 = "x3" (three bf16 planes, xp_set_dense_products products; gemm_x3_core.h).  The fused block kernels (xp_mlp_fused_x3,
// xp_ln_proj_x3) are x3 in both.  XP_DENSE_ENGINE=x3|h2 sets the initial value.
static std::atomic<int> g_dense_engine{[] { const char* e = getenv("XP_DENSE_ENGINE"); return (e && std::string(e) == "x3") ? 0 : 1; }()};
int xp_dense_engine_value() { return g_dense_engine.load(); }
extern "C" int xp_get_dense_engine(void) { return g_dense_engine.load(); }
extern "C" int xp_set_dense_engine(int engine) {
    XP_CHECK_ARG(engine == 0 || engine == 1, "xp_set_dense_engine: 0 (x3: split bf16) or 1 (h2: split fp16); got %d", engine);
    g_dense_engine.store(engine);
    return XP_OK;
}

// Per-launch engine override of xp_xpoint_forward(_ex) under the split-fp16 engine (round 6): bit i set = dense launch i of the forward (numbering:
// include/xpoint_hip.h, xp_set_dense_override) runs on the split-bf16 planes (no operand-range limit) instead.  Process-wide like the engine itself; the
// Python host sets it for the duration of one (host-synchronous) enqueue.
static std::atomic<unsigned long long> g_dense_override{0ull};
unsigned long long xp_dense_override_value() { return g_dense_override.load(); }
extern "C" unsigned long long xp_get_dense_override(void) { return g_dense_override.load(); }
extern "C" int xp_set_dense_override(unsigned long long mask) { g_dense_override.store(mask); return XP_OK; }
// Mixed-precision class ("amp16"): the arithmetic of the reference's `mixed_precision: true` deployment (XPoint.py:182, autocast):
// convolutions / linear layers on half operands with half outputs, LayerNorm / GELU / SiLU / BatchNorm / residual adds returning half tensors,
// the scan, out_norm, softmax and normalize in f32.  Process-wide, read at launch time by xp_gemm_nt_h2 / xp_conv3x3_nhwc_h2 (epilogue rounding),
// xp_layernorm, xp_dwconv3x3_silu, xp_stem_conv_ln_gelu, xp_ss2d_core_fwd and xp_xpoint_forward(_ex).
static std::atomic<int> g_amp_mode{0};
int xp_amp_value() { return g_amp_mode.load(); }
extern "C" int xp_get_amp_mode(void) { return g_amp_mode.load(); }
extern "C" int xp_set_amp_mode(int mode) {
    XP_CHECK_ARG(mode == 0 || mode == 1, "xp_set_amp_mode: 0 (off) or 1 (fp16 rounding at the autocast boundaries); got %d", mode);
    g_amp_mode.store(mode);
    return XP_OK;
}
extern "C" int xp_set_dense_products(int n) {
    XP_CHECK_ARG(n == 6 || n == 3 || n == 1, "xp_set_dense_products: 6 (f32-grade, default), 3 (two-plane operands) or 1 (plain bf16 operands); got %d", n);
    g_dense_products.store(n);
    return XP_OK;
}
