// cpflow_amd host side of the C ABI (include/cpflow_amd.h): plan construction
// (gate program, embedded target, penalty table, Adam bias-correction table),
// kernel variant selection and launches.  gfx950 only.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <set>
#include <vector>

#include "../../include/cpflow_amd.h"
#include "cpf_device.h"
#include "cpf_blocks.h"
#include "cpf_layered.h"
#include "cpf_split.h"
#include "cpf_dense.h"
#include "cpf_dense_aot.h"
#include "cpf_zyz.h"
#include "cpf_zyzl.h"
#include "cpf_zyz3.h"
#include "cpf_zyz3_aot.h"
#include "cpf_zyz_aot.h"
#include "cpf_jit.h"

namespace {

thread_local char g_err[512] = "";

// Kernel-selection switches (two-ended vs one-wave, stored states, dense blocks, relabelling ...) are set through
// cpf_set_tuning (include/cpflow_amd.h) by the tests and the A/B tools.  The release library reads no CPF_* environment
// variable for them; a build with -DCPF_ENV_TUNING falls back to the environment (tools/).
std::map<std::string, std::string>& tuning_map() { static std::map<std::string, std::string> m; return m; }
std::mutex& tuning_mutex() { static std::mutex m; return m; }
// a switch's value, copied out under the lock (cpf_set_tuning may run on another thread): tests as false when unset
struct TuneVal {
    std::string s; bool set = false;
    explicit operator bool() const { return set; }
    char operator[](size_t i) const { return i < s.size() ? s[i] : '\0'; }
};
TuneVal tuning(const char* name) {
    TuneVal v;
    {
        std::lock_guard<std::mutex> lock(tuning_mutex());
        auto& m = tuning_map();
        auto it = m.find(name);
        if (it != m.end()) { v.s = it->second; v.set = true; return v; }
    }
#ifdef CPF_ENV_TUNING
    if (const char* e = getenv(name)) { v.s = e; v.set = true; }
#endif
    return v;
}
// switches that exist in a -DCPF_DIAGNOSTICS build only (the release library does not even hold their names)
#ifdef CPF_DIAGNOSTICS
#define CPF_DIAG_TUNING(name) tuning(name)
#else
#define CPF_DIAG_TUNING(name) TuneVal()
#endif

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define CPF_HIP(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail((int)e_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

}  // namespace

struct cpf_plan {
    int device = 0;
    int n = 0, K = 0, n_rot = 0, nb = 0, P = 0, G = 0, T = 0, NQ = 0;
    int entangler = 0;
    std::string rot;
    // A labelling = the names the kernels give the problem's qubits: [0] the problem's own, [1] most-used qubits last
    // (fill_plan_shape; only 4 and 5 qubits).  Each has its own placements, gate program and target; a launch picks one
    // (select_labelling) and the three vectors below are the picked labelling's.
    struct Labelling {
        std::vector<int> qperm, h_place, layer_ud;
        DevBuf d_prog, d_pgate, d_target;
    };
    Labelling lab[2];
    bool has_relabel = false;
    std::vector<int> layer_ud;       // (up, dn, up, dn, ...) of the shortest period of the placements
    std::vector<int> h_place;        // (up, dn) of every block
    std::vector<int> qperm;          // kernel qubit of problem qubit q
    std::map<int, DevBuf> dense_tab; // per waves-per-sample variant of the dense kernel: its per-block table (cpf_dense.h)
    // Per-step scratch of the kernels that read per-block tables through the scalar cache (cpf_dense.h: [samples][K][32] floats,
    // cpf_zyz.h: [samples][K][16]), rebuilt by the kernel every step.  One buffer per stream: launches of one plan on different
    // streams may overlap, launches on one stream are ordered.
    struct Scratch { DevBuf buf; size_t bytes = 0; };
    std::map<hipStream_t, Scratch> scratch;
    std::mutex scratch_mutex;
    DevBuf zyz_sides[2];             // static structure of the circuit for the ZYZ-form kernels (cpf_zyz.h ZyzSideInfo), per labelling
    int dense_surf_mask[4] = {0, 0, 0, 0};   // ... and the qubits whose surface gates its sweeps still apply (not folded into a block)
    bool use_jit = true;             // CPF_JIT=0 disables runtime specialisation
    std::string jit_note;            // why the last launch did not use the specialised kernel (diagnostics)
    std::string variant;             // kernel variant of the last launch
    bool force_generic = false;      // CPF_FORCE_GENERIC=1: always use the generic per-gate kernel (A/B testing)
    std::vector<uint8_t> h_pkind;
    std::vector<int4> h_prog;
    DevBuf d_pkind, d_bias, d_ibias;
    cpf::KArgs base{};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    int num_cu = 256;
    int64_t lds_per_cu = 160 * 1024;   // bytes of LDS per CU (device property; 160 KiB on gfx950)
    ~cpf_plan() {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
    }
};

namespace {

// labelling of a launch: relabelled for the optimiser on the one-wave layouts, the problem's own otherwise (and for
// cpf_unitary_device, which returns U in the problem's labels)
int labelling_for(const cpf_plan* plan, int mode, int lw) { return (plan->has_relabel && lw == 0 && mode != cpf::MODE_UNITARY) ? 1 : 0; }
void select_labelling(cpf_plan* plan, int which) {
    const cpf_plan::Labelling& lb = plan->lab[which];
    plan->qperm = lb.qperm; plan->h_place = lb.h_place; plan->layer_ud = lb.layer_ud;
}
void apply_labelling(const cpf_plan* plan, int which, cpf::KArgs& a) {
    const cpf_plan::Labelling& lb = plan->lab[which];
    a.prog = lb.d_prog.as<int4>(); a.pgate = lb.d_pgate.as<int32_t>(); a.target = lb.d_target.as<float2>();
}

// Penalty table: the reference builds slopes / intercepts / break points from Python floats
// (double) and they meet the float32 angle as float32 scalars (penalty.py:14-15, 44-71).
void build_pieces(const cpf_penalty& po, cpf::Piece pc[10]) {
    const double pi = 3.141592653589793, xm = po.xmax, ym = po.ymax, p0 = po.plato_0, p1 = po.plato_1, p2 = po.plato_2;
    const double X[11] = {0, p0, xm - p2, xm + p2, pi - p1, pi + p1, pi + xm - p2, pi + xm + p2, 2 * pi - p0, 2 * pi, 3 * pi};
    const double Y[11] = {0, 0, ym, ym, 1, 1, ym, ym, 0, 0, 0};
    for (int i = 0; i < 9; i++) {
        const double x0 = X[i], y0 = Y[i], x1 = X[i + 1], y1 = Y[i + 1];
        pc[i].lo = (float)x0;
        pc[i].hi = (float)x1;
        pc[i].slope = (float)((y1 - y0) / (x1 - x0));
        pc[i].icpt = (float)((x0 * y1 - x1 * y0) / (x0 - x1));
    }
    pc[9].lo = (float)X[9]; pc[9].hi = (float)X[10]; pc[9].slope = 0.f; pc[9].icpt = 1.f;
}

int validate(const cpf_problem* pb) {
    if (!pb) return fail(-1, "problem is NULL");
    if (pb->n_qubits < CPF_MIN_QUBITS || pb->n_qubits > CPF_MAX_QUBITS)
        return fail(-2, "n_qubits=%d outside [%d, %d]", pb->n_qubits, CPF_MIN_QUBITS, CPF_MAX_QUBITS);
    if (pb->num_blocks < 0) return fail(-3, "num_blocks=%d < 0", pb->num_blocks);
    if (pb->num_blocks > 0 && !pb->placements) return fail(-3, "placements is NULL");
    if (pb->n_rot < 0 || pb->n_rot > 8 || (pb->n_rot > 0 && !pb->rot_axes)) return fail(-4, "bad rotation_gates");
    for (int i = 0; i < pb->n_rot; i++)
        if (pb->rot_axes[i] != 'x' && pb->rot_axes[i] != 'y' && pb->rot_axes[i] != 'z')
            return fail(-4, "rotation gate '%c' not in {x,y,z}", pb->rot_axes[i]);
    if (pb->entangler != CPF_ENTANGLER_CP && pb->entangler != CPF_ENTANGLER_CZ && pb->entangler != CPF_ENTANGLER_CX)
        return fail(-5, "entangler %d not supported (cp, cz, cx)", pb->entangler);
    if (pb->loss_kind != CPF_LOSS_HST && pb->loss_kind != CPF_LOSS_STATE && pb->loss_kind != CPF_LOSS_RELATIVE_PHASE)
        return fail(-6, "loss_kind %d not supported", pb->loss_kind);
    if (!pb->target) return fail(-7, "target is NULL");
    for (int b = 0; b < pb->num_blocks; b++) {
        const int p0 = pb->placements[2 * b], p1 = pb->placements[2 * b + 1];
        if (p0 < 0 || p1 < 0 || p0 >= pb->n_qubits || p1 >= pb->n_qubits || p0 == p1)
            return fail(-3, "placement %d = [%d, %d] invalid for %d qubits", b, p0, p1, pb->n_qubits);
    }
    if (pb->num_iterations < 0) return fail(-8, "num_iterations < 0");
    if (pb->penalty.kind != CPF_PENALTY_LINEAR && pb->penalty.kind != CPF_PENALTY_L1) return fail(-9, "penalty kind");
    return 0;
}

template <int NQ, int LW>
int launch_variant(const cpf::KArgs& a, int64_t S, hipStream_t stream) {
    using L = cpf::Lay<NQ, LW>;
    const size_t lds = cpf::lds_floats(a.P, a.G, L::WPS) * sizeof(float);
    if (lds > 160 * 1024) return fail(-20, "P=%d needs %zu B of LDS (> 160 KiB)", a.P, lds);
    if (lds > 64 * 1024)
        CPF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&cpf::k_cp_template<NQ, LW>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((cpf::k_cp_template<NQ, LW>), dim3((unsigned)S), dim3(L::THREADS), lds, stream, a);
    CPF_HIP(hipGetLastError());
    return 0;
}

// More than 64 KiB of dynamic LDS for a kernel loaded from a code object: ROCm 7 takes the hipFunction_t in hipFuncSetAttribute
// (and launches such a kernel even without it; measured with a 100 KB kernel on the MI355X boxes).  False: keep to 64 KiB.
bool module_lds_opt_in(hipFunction_t fn, size_t bytes) {
    if (bytes <= 64 * 1024) return true;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}

// Ahead-of-time instances of the layered kernel: 3 qubits, the reference's two layer constructors
// (chain_layer(3), connected_layer(3); topology.py:7-12), 'xyz' / 'xz', optimiser state for <= 128 / 256 angles.
// Stored phases cost 512 (K + 1) bytes of LDS per sample: use them while that does not cut the number of
// resident workgroups the launch needs (S / CUs, at most ~5 per SIMD by registers).
bool phases_fit(const cpf_plan* plan, int64_t S, size_t lds_bytes_with_phases) {
    const int num_cu = plan->num_cu;
    { const TuneVal e = tuning("CPF_MSTORE"); if (e) return e[0] == '1'; }          // A/B and tests
    // same-box A/B (ms, store vs rebuild): config 2 S=1024 10.06/10.53, S=4096 24.45/26.15, S=16384 96.1/93.5;
    // config 3 S=1024 12.89/13.75, S=4096 38.5/35.4 -> store only while LDS still admits the workgroups the launch
    // wants per CU (registers cap that at ~20)
    const int64_t want = std::min<int64_t>((S + num_cu - 1) / num_cu, 20);
    return (int64_t)(plan->lds_per_cu / (int64_t)lds_bytes_with_phases) >= want;
}
// Two waves per sample (cpf_split.h) while the one-wave kernel would leave SIMDs with a single wave: the step is
// then one exposed dependent chain, and cutting it in two is worth more than the second wave's issue slots.
// With the block count a compile-time constant (kspec) the two-ended kernel also wins once the SIMDs are full
// (config 2 shape, same box: S = 2048 15.1 vs 17.8 ms, 16384 84.2 vs 93.7 ms; config 3 shape S = 4096 28.8 vs 33.8 ms).
bool want_split(const cpf_plan* plan, const cpf::KArgs& a, int K, int LL, int rot_code, int64_t S, bool kspec) {
    if (!cpf::layered_merge(3, 0, rot_code) || !cpf::split_supported(K, LL) || a.loss_kind == 2) return false;
    { const TuneVal e = tuning("CPF_SPLIT"); if (e) return e[0] == '1'; }           // A/B and tests
    return kspec || S <= (int64_t)plan->num_cu * 4;
}
bool kspec_enabled() { const TuneVal e = tuning("CPF_KSPEC"); return !(e && e[0] == '0'); }   // A/B and tests: 0 = K from the plan
// Stored states (cpf_split.h, MSTORE = 2) cost 512 (NI K + 2) more bytes of LDS per sample (25 KB at config 2): use
// them while LDS still admits the workgroups the launch wants on a CU.
// cap: 64 KiB for the kernels built into the library (launched without an LDS opt-in), the CU's LDS for run-time built ones
bool states_fit(const cpf_plan* plan, int64_t S, int rot_code, size_t lds_bytes_with_states, size_t cap = 64 * 1024) {
    const int num_cu = plan->num_cu;
    if (cpf::rot_nr(rot_code) < 2 || lds_bytes_with_states > cap) return false;
    { const TuneVal e = tuning("CPF_PSTORE"); if (e) return e[0] == '1'; }          // A/B and tests
    // only while every workgroup of the launch is resident at once (same box, config 2 shape: S = 1024 6.95 vs 7.15 ms
    // with / without states, S = 2048 17.5 vs 15.1, S = 4096 32.1 vs 22.8)
    return (int64_t)(plan->lds_per_cu / (int64_t)lds_bytes_with_states) >= (S + num_cu - 1) / num_cu;
}
// K-specialised two-ended kernels exist ahead of time for the BASELINE 3-qubit configurations
// (chain_layer(3) with 12 blocks, connected_layer(3) with 16; 'xyz'); KC = 0: K is read from the plan.
template <int ROT, int PPT, int KC, int... UD>
int launch_split3(cpf_plan* plan, const cpf::KArgs& a, int K, int64_t S, hipStream_t stream) {
    const size_t lds2 = (size_t)cpf::SplitLds::total(K, 2, cpf::inner_gates<ROT>()) * sizeof(float);
    const char* tag = KC ? " K" : "";
    if constexpr (cpf::rot_nr(ROT) >= 2 && KC > 0) {
        // Launches that leave at most one wave of this kernel on a SIMD (one or two samples per CU): the cut that balances the
        // two waves' chains (MSTORE bit 17), and with two samples per CU both in ONE workgroup (bit 16) - consecutive two-wave
        // workgroups start one SIMD apart, (0, 1) then (1, 2), which books one SIMD twice and leaves one idle
        // (tools/micro/hwid.py).  Config 3's share (512 samples): 6.88 -> 6.03 ms; config 2's shape 5.84 -> 5.09 (512
        // samples), 5.63 -> 4.83 (256).  With four samples per CU neither helps (6.03 -> 6.35 ms at config 2).
        const int64_t per_cu = (S + plan->num_cu - 1) / plan->num_cu;
        int pairs = per_cu == 2 ? 2 : (per_cu == 1 ? 1 : 0);
        { const TuneVal e = tuning("CPF_PAIRS"); if (e) pairs = e[0] == '2' ? 2 : (e[0] == '1' ? 1 : 0); }      // A/B and tests: 0 = the saturated form
        if (pairs == 2 && (S < 2 || ((S + 1) / 2 + plan->num_cu - 1) / plan->num_cu * (int64_t)(2 * lds2) > plan->lds_per_cu)) pairs = 0;
        if (pairs == 1 && !states_fit(plan, S, ROT, lds2)) pairs = 0;
        if (pairs == 2) {
            auto kern = &cpf::k_cp_split<ROT, PPT, 2 + 4 * KC + (3 << 16), UD...>;
            CPF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds2)));
            hipLaunchKernelGGL(kern, dim3((unsigned)((S + 1) / 2)), dim3(256), 2 * lds2, stream, a);
            CPF_HIP(hipGetLastError());
            plan->variant = "split/aot waves=2 states K pairs=2";
            return 0;
        }
        if (pairs == 1) {
            hipLaunchKernelGGL((cpf::k_cp_split<ROT, PPT, 2 + 4 * KC + (2 << 16), UD...>), dim3((unsigned)S), dim3(128), lds2, stream, a);
            CPF_HIP(hipGetLastError());
            plan->variant = "split/aot waves=2 states K lone";
            return 0;
        }
    }
    if constexpr (cpf::rot_nr(ROT) >= 2) {
        if (states_fit(plan, S, ROT, lds2)) {
            hipLaunchKernelGGL((cpf::k_cp_split<ROT, PPT, 2 + 4 * KC, UD...>), dim3((unsigned)S), dim3(128), lds2, stream, a);
            CPF_HIP(hipGetLastError());
            plan->variant = std::string("split/aot waves=2 states") + tag;
            return 0;
        }
    }
    const size_t lds = (size_t)cpf::SplitLds::total(K, 1) * sizeof(float);
    hipLaunchKernelGGL((cpf::k_cp_split<ROT, PPT, 1 + 4 * KC, UD...>), dim3((unsigned)S), dim3(128), lds, stream, a);
    CPF_HIP(hipGetLastError());
    plan->variant = std::string("split/aot waves=2") + tag;
    return 0;
}
template <int ROT, int PPT, int... UD>
int launch_layered3(const cpf_plan* plan_c, const cpf::KArgs& a, int K, int64_t S, hipStream_t stream) {
    cpf_plan* plan = const_cast<cpf_plan*>(plan_c);
    plan->variant = "layered/aot waves=1";
    constexpr int XYZ = cpf::rot_pack(3, cpf::OP_RX, cpf::OP_RY, cpf::OP_RZ);
    constexpr int LL = (int)sizeof...(UD) / 2;
    const bool kspec = kspec_enabled() && ROT == XYZ && PPT == 2 && ((LL == 2 && K == 12) || (LL == 3 && K == 16));
    if (want_split(plan, a, K, LL, ROT, S, kspec)) {
        if constexpr (ROT == XYZ && PPT == 2 && LL == 2) { if (kspec) return launch_split3<ROT, PPT / 2, 12, UD...>(plan, a, K, S, stream); }
        if constexpr (ROT == XYZ && PPT == 2 && LL == 3) { if (kspec) return launch_split3<ROT, PPT / 2, 16, UD...>(plan, a, K, S, stream); }
        return launch_split3<ROT, PPT / 2, 0, UD...>(plan, a, K, S, stream);
    }
    const bool merge = cpf::layered_merge(3, 0, ROT);
    const size_t lds1 = (size_t)cpf::LayeredLds<3>::total(K, 1, merge) * sizeof(float);
    const size_t lds0 = (size_t)cpf::LayeredLds<3>::total(K, 1, false) * sizeof(float);
    if (merge && lds1 <= 64 * 1024 && phases_fit(plan, S, lds1))
        hipLaunchKernelGGL((cpf::k_cp_layered<3, 0, ROT, PPT, 1, UD...>), dim3((unsigned)S), dim3(64), lds1, stream, a);
    else
        hipLaunchKernelGGL((cpf::k_cp_layered<3, 0, ROT, PPT, 0, UD...>), dim3((unsigned)S), dim3(64), lds0, stream, a);
    CPF_HIP(hipGetLastError());
    return 0;
}
template <int ROT, int PPT>
bool launch_layered3_layer(const cpf_plan* plan, const cpf::KArgs& a, int64_t S, hipStream_t stream, int* rc) {
    const std::vector<int>& ud = plan->layer_ud;
    if (ud == std::vector<int>{0, 1, 1, 2}) { *rc = launch_layered3<ROT, PPT, 0, 1, 1, 2>(plan, a, plan->K, S, stream); return true; }
    if (ud == std::vector<int>{0, 1, 0, 2, 1, 2}) { *rc = launch_layered3<ROT, PPT, 0, 1, 0, 2, 1, 2>(plan, a, plan->K, S, stream); return true; }
    return false;
}
bool try_layered3_aot(const cpf_plan* plan, const cpf::KArgs& a, int64_t S, hipStream_t stream, int* rc) {
    { const TuneVal e = CPF_DIAG_TUNING("CPF_NO_AOT3"); if (e && e[0] == '1') return false; }      // A/B: force the runtime build
    if (plan->force_generic || a.mode != cpf::MODE_OPTIMIZE || plan->NQ != 3 || plan->n != 3 ||
        plan->entangler != CPF_ENTANGLER_CP || plan->K < 1)
        return false;
    const int ppt = (a.P + 63) / 64;
    constexpr int XYZ = cpf::rot_pack(3, cpf::OP_RX, cpf::OP_RY, cpf::OP_RZ);
    constexpr int XZ = cpf::rot_pack(2, cpf::OP_RX, cpf::OP_RZ, 0);
    if (plan->rot == "xyz" && ppt <= 2) return launch_layered3_layer<XYZ, 2>(plan, a, S, stream, rc);
    if (plan->rot == "xyz" && ppt <= 4) return launch_layered3_layer<XYZ, 4>(plan, a, S, stream, rc);
    if (plan->rot == "xz" && ppt <= 2) return launch_layered3_layer<XZ, 2>(plan, a, S, stream, rc);
    if (plan->rot == "xz" && ppt <= 4) return launch_layered3_layer<XZ, 4>(plan, a, S, stream, rc);
    return false;
}

// Ahead-of-time instances for the two larger BASELINE configurations, so that they do not depend on a compiler
// being present at run time: 4 qubits chain_layer(4) (config 4: K = 24, 180 angles, 1 wave) and 6 qubits on a
// ring (config 5: K = 60, 438 angles, 4 waves).
// One-wave layered kernels of 4 and more qubits: samples per workgroup (cpf_layered.h layered_body, KArgs::group).  Four waves of
// one workgroup land on the CU's four SIMDs, so a launch of up to 1024 samples runs one wave per SIMD and one of 2048 two, whatever
// the dispatcher would have done with single-wave workgroups (same box, 4-qubit star, 23 blocks: 800 / 900 samples 43 ms as
// single-wave workgroups, 26 ms grouped; 5-qubit chain, 36 blocks, 1000 samples: 118 or 190 ms from run to run, 118 grouped).
int layered_group(const cpf_plan* plan, int lw, int64_t S, size_t lds_per_sample) {
    int g = 1;
    // not for launches that fill every SIMD twice anyway (config 4, 2048 samples: 45.6 ms single, 45.8 grouped)
    if (lw == 0 && plan->NQ >= 4 && S > plan->num_cu && S < 8 * (int64_t)plan->num_cu) g = 4;
    { const TuneVal e = tuning("CPF_GROUP"); if (e && lw == 0 && plan->NQ >= 4) g = e[0] == '4' ? 4 : (e[0] == '2' ? 2 : 1); }      // A/B and tests
    while (g > 1 && (int64_t)g * (int64_t)lds_per_sample > plan->lds_per_cu) g >>= 1;        // above 64 KiB per workgroup: opted in at the launch
    return g;
}
template <int NQ, int LW, int ROT, int PPT, int... UD>
int launch_layered_aot(const cpf_plan* plan, const cpf::KArgs& a_in, int K, int64_t S, hipStream_t stream) {
    const size_t lds = (size_t)cpf::LayeredLds<NQ>::total(K, 1 << LW, false) * sizeof(float);
    if (lds > 64 * 1024) return fail(-20, "LDS footprint above 64 KiB");
    cpf::KArgs a = a_in;
    a.group = layered_group(plan, LW, S, lds);
    if (lds * a.group > 64 * 1024)
        CPF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&cpf::k_cp_layered<NQ, LW, ROT, PPT, 0, UD...>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds * a.group)));
    hipLaunchKernelGGL((cpf::k_cp_layered<NQ, LW, ROT, PPT, 0, UD...>), dim3((unsigned)((S + a.group - 1) / a.group)), dim3((64 << LW) * a.group),
                       lds * a.group, stream, a);
    CPF_HIP(hipGetLastError());
    return 0;
}
bool try_layered_big_aot(const cpf_plan* plan, const cpf::KArgs& a, int lw, int64_t S, hipStream_t stream, int* rc) {
    { const TuneVal e = CPF_DIAG_TUNING("CPF_NO_AOT3"); if (e && e[0] == '1') return false; }
    if (plan->force_generic || a.mode != cpf::MODE_OPTIMIZE || plan->n != plan->NQ || plan->entangler != CPF_ENTANGLER_CP ||
        plan->rot != "xyz")
        return false;
    constexpr int XYZ = cpf::rot_pack(3, cpf::OP_RX, cpf::OP_RY, cpf::OP_RZ);
    const int ppt = (a.P + (64 << lw) - 1) / (64 << lw);
    if (plan->NQ == 4 && lw == 0 && ppt == 3 && plan->layer_ud == std::vector<int>{0, 2, 2, 3, 3, 1}) {      // chain_layer(4) relabelled (fill_plan_shape)
        *rc = launch_layered_aot<4, 0, XYZ, 3, 0, 2, 2, 3, 3, 1>(plan, a, plan->K, S, stream);
        return true;
    }
    if (plan->NQ == 6 && lw == 2 && ppt == 2 && plan->layer_ud == std::vector<int>{0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 0}) {
        *rc = launch_layered_aot<6, 2, XYZ, 2, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 0>(plan, a, plan->K, S, stream);
        return true;
    }
    return false;
}


// The scratch buffer of (plan, stream), at least `bytes` large.  Growing it waits for the stream's earlier launches (they
// read the old buffer) and frees it, which synchronises the device - a one-off per plan and stream for the largest batch.
int scratch_for(cpf_plan* plan, hipStream_t stream, size_t bytes, float** out) {
    std::lock_guard<std::mutex> lock(plan->scratch_mutex);
    cpf_plan::Scratch& sc = plan->scratch[stream];
    if (bytes > sc.bytes) {
        CPF_HIP(hipStreamSynchronize(stream));
        sc.bytes = 0;
        if (sc.buf.p) { (void)hipFree(sc.buf.p); sc.buf.p = nullptr; }
        CPF_HIP(sc.buf.alloc(bytes));
        sc.bytes = bytes;
    }
    *out = sc.buf.as<float>();
    return 0;
}

// ---- dense-block kernel (cpf_dense.h): 4..6 qubits, any layer / rotation string / entangler.  In order of preference:
// a layer-specialised instance built ahead of time (cpf_dense_aot.hip), one built at run time (cpf_jit.h), the
// table-dispatched instance (no compiler needed, ~1.5x slower: a scalar branch tree per block).
template <class L>
int dense_table(cpf_plan* plan, DevBuf& tab) {
    if (tab.p) return 0;
    const int K = plan->K, per_block = 1 + 2 * plan->n_rot;
    std::vector<cpf::DenseBlock> h((size_t)K);
    // the same layer-aware rule as the layer-specialised kernels (cpf_dense.h dense_layer_entry): pair positions must agree
    const int LLn = (int)plan->layer_ud.size() / 2;
    for (int b = 0; b < K; b++) h[b] = cpf::dense_layer_entry<L>(plan->layer_ud.data(), LLn, b % LLn, 3 * plan->n + per_block * b);
    // surface gates: folded into the first block that touches the qubit (cpf_dense_math.h); plan->dense_surf_mask keeps
    // the qubits no block touches - their gates stay in the sweeps
    int unfolded = (1 << plan->n) - 1;
    const bool fold = !(tuning("CPF_DENSE_FOLD") && tuning("CPF_DENSE_FOLD")[0] == '0');
    for (int b = 0; b < K; b++) {
        const int up = plan->h_place[2 * b], dn = plan->h_place[2 * b + 1];
        h[b].z |= (up << 12) | (dn << 16);
        if (fold && ((unfolded >> up) & 1)) { h[b].z |= 512; unfolded &= ~(1 << up); }
        if (fold && ((unfolded >> dn) & 1)) { h[b].z |= 1024; unfolded &= ~(1 << dn); }
    }
    plan->dense_surf_mask[L::LW] = unfolded;
    CPF_HIP(tab.alloc(h.size() * sizeof(cpf::DenseBlock)));
    CPF_HIP(hipMemcpy(tab.p, h.data(), h.size() * sizeof(cpf::DenseBlock), hipMemcpyHostToDevice));
    return 0;
}
// layers with an ahead-of-time specialised instance (cpf_dense_aot.hip)
// rotation code of the plan for the dense kernel's compile-time walks (cpf_blocks.h), 0 when it has more than 3 letters
int dense_rot_code(const cpf_plan* plan) {
    if (plan->n_rot > 3) return 0;
    int kinds[3] = {0, 0, 0};
    for (int i = 0; i < plan->n_rot; i++) kinds[i] = plan->rot[i] == 'x' ? cpf::OP_RX : (plan->rot[i] == 'y' ? cpf::OP_RY : cpf::OP_RZ);
    return cpf::rot_with_ent(cpf::rot_pack(plan->n_rot, kinds[0], kinds[1], kinds[2]), plan->entangler == CPF_ENTANGLER_CZ ? 1 : (plan->entangler == CPF_ENTANGLER_CX ? 2 : 0));
}
bool dense_has_aot(int nq, int lw, int rot, const std::vector<int>& ud) {
    if (rot != CPF_DENSE_AOT_ROT) return false;
    static const std::vector<std::pair<int, std::vector<int>>> aot = {
        {6 * 8 + 1, {0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 0}}, {6 * 8 + 2, {0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 0}},
        {5 * 8 + 0, {0, 1, 1, 2, 2, 3, 3, 4}}, {4 * 8 + 0, {0, 1, 1, 2, 2, 3}}};
    for (const auto& kv : aot) if (kv.first == nq * 8 + lw && kv.second == ud) return true;
    return false;
}
cpf_jit::Spec dense_jit_spec(int nq, int lw, int rot, const std::vector<int>& ud) {
    cpf_jit::Spec spec{"cpf_dense.h", cpf_jit::body_expr("dense_body", {nq, lw, rot}, ud), 64 << lw};
    spec.arg = "cpf::DenseArgs";
    spec.min_waves = 2;                       // layer-specialised sweeps fit 256 registers (two waves per SIMD)
    spec.flags = {CPF_DENSE_MLLVM_FLAGS};
    return spec;
}
bool dense_wanted(const cpf_plan* plan, int mode) {
    if (plan->force_generic || mode != cpf::MODE_OPTIMIZE || plan->n != plan->NQ || plan->NQ < 4 || plan->K < 1 || plan->n_rot > 8) return false;
    const TuneVal e = tuning("CPF_DENSE");
    // default: 6 qubits; and 4 / 5 qubits with more than three rotation letters, which the gate-by-gate kernels only serve through the
    // generic per-gate kernel (same box, 'zyxz': 4 qubits, 12 blocks, 2048 samples 142.2 -> 26.9 ms; 5 qubits, 16 blocks, 1000 samples
    // 258.6 -> 42.5; 4 qubits, 100 samples 112.9 -> 20.2)
    return e ? e[0] != '0' : (plan->NQ == 6 || plan->n_rot > 3);
}
int dense_lw(const cpf_plan* plan, int64_t S) {
    const TuneVal w = tuning("CPF_DENSE_LW");
    const int64_t simds = (int64_t)plan->num_cu * 4;
    switch (plan->NQ) {
        // 6 qubits: two waves per sample (32 amplitudes per thread) once the samples fill every SIMD twice, four waves (16 per
        // thread) below: same box, ZYZ-form kernel, config 5's shape at 64 / 256 / 512 samples 116.9 / 118.2 / 206.3 ms on four waves
        // against 174.2 / 175.1 / 300.7 on two; 768 samples 300.1 against 301.4; 1024 samples 392.8 against 332.3
        case 6: return w ? (w[0] == '2' ? 2 : 1) : (2 * S <= simds ? 2 : 1);
        case 5: return (w ? w[0] == '1' : 2 * S < simds) ? 1 : 0;
        default: return 0;
    }
}
bool dense_specialised(const cpf_plan* plan) {
    const TuneVal sp = tuning("CPF_DENSE_SPECIALISED");               // A/B and tests: 0 = table-dispatched instance only
    return !(sp && sp[0] == '0') && !plan->layer_ud.empty() && plan->layer_ud.size() <= 32;
}

int launch_dense(cpf_plan* plan, const cpf::KArgs& a, int nq, int lw, int64_t S, hipStream_t stream) {
    DevBuf& tab = plan->dense_tab[lw];
    int rc = 0;
    switch (nq * 8 + lw) {
        case 6 * 8 + 1: rc = dense_table<cpf::DLay<6, 1>>(plan, tab); break;
        case 6 * 8 + 2: rc = dense_table<cpf::DLay<6, 2>>(plan, tab); break;
        case 5 * 8 + 0: rc = dense_table<cpf::DLay<5, 0>>(plan, tab); break;
        case 5 * 8 + 1: rc = dense_table<cpf::DLay<5, 1>>(plan, tab); break;
        case 4 * 8 + 0: rc = dense_table<cpf::DLay<4, 0>>(plan, tab); break;
        default: return fail(-2, "no dense layout for %d qubits, %d waves", nq, 1 << lw);
    }
    if (rc) return rc;
    const int wps = 1 << lw, threads = 64 << lw;
    const size_t lds = (size_t)cpf::DenseLds(a.P, a.G, plan->K, nq, wps).total * sizeof(float);
    if (lds > (size_t)plan->lds_per_cu) return fail(-20, "dense kernel: %zu B of LDS per sample", lds);
    // global scratch of the scalar-register path: one 128-byte matrix per block and sample, rebuilt every step by the kernel
    // (layer-specialised instances only: the table-dispatched one keeps the matrices in LDS)
    const bool specialised = dense_specialised(plan);
    float* gmat = nullptr;
    if (specialised) { if (int rc2 = scratch_for(plan, stream, (size_t)S * (size_t)plan->K * 32 * sizeof(float), &gmat)) return rc2; }
    const cpf::DenseArgs da{a, tab.as<cpf::DenseBlock>(), plan->K, plan->n_rot, plan->dense_surf_mask[lw], gmat};
    // 1. run-time specialised (shipped cache / compile) unless an ahead-of-time instance of this very layer exists
    hipError_t e = hipSuccess;
    const std::string waves = " waves=" + std::to_string(wps);
    if (specialised) {
        const int rot = dense_rot_code(plan);
        const bool have_aot = dense_has_aot(nq, lw, rot, plan->layer_ud);
        if (!have_aot && plan->use_jit) {
            const cpf_jit::Spec spec = dense_jit_spec(nq, lw, rot, plan->layer_ud);
            const cpf_jit::Kernel& k = cpf_jit::get(plan->device, spec);
            if (k.fn && module_lds_opt_in(k.fn, lds)) {
                cpf::DenseArgs args = da;
                void* params[] = {&args};
                e = hipModuleLaunchKernel(k.fn, (unsigned)S, 1, 1, (unsigned)threads, 1, 1, (unsigned)lds, stream, params, nullptr);
                plan->variant = "dense/jit(" + k.backend + ")" + waves;
                return e == hipSuccess ? 0 : fail((int)e, "hipModuleLaunchKernel failed: %s", hipGetErrorString(e));
            }
            plan->jit_note = k.error;
            if (k.error.find("forbidden") != std::string::npos) return fail(-30, "%s", k.error.c_str());
        }
    }
    const int kind = cpf_dense_launch_aot(nq, lw, dense_rot_code(plan), plan->layer_ud, specialised, da, S, lds, stream, &e);
    if (kind == 0) return fail(-2, "no dense kernel for %d qubits, %d waves", nq, wps);
    if (e != hipSuccess) return fail((int)e, "dense kernel launch failed: %s", hipGetErrorString(e));
    plan->variant = std::string(kind == 1 ? "dense/aot" : "dense/table") + waves + (kind == 2 && !plan->jit_note.empty() ? " (" + plan->jit_note + ")" : "");
    return 0;
}
// Default for 6 qubits (2 waves per sample, 32 amplitudes per thread; 4 waves on request): config 5 runs 756 ms against
// 896 ms on the gate-by-gate kernel.  With 16 or fewer amplitudes per thread the fixed costs of a block - the 32-value
// environment reduction, the exchanges of lane qubits, the 128-byte matrix load - weigh as much as its arithmetic: 5 qubits
// 150 vs 144 ms, 4 qubits (config 4) 122 vs 58 ms, so there the gate-by-gate kernels stay the default (CPF_DENSE=1 selects the
// dense one: 5 qubits 1 wave per sample while the samples fill the SIMDs, else 2).
bool try_dense(cpf_plan* plan, const cpf::KArgs& a, int64_t S, hipStream_t stream, int* rc) {
    if (!dense_wanted(plan, a.mode)) return false;
    *rc = launch_dense(plan, a, plan->NQ, dense_lw(plan, S), S, stream);
    return true;
}


// ---- ZYZ-form kernel (cpf_zyz.h): the shapes the dense kernel serves whose blocks are CP or CZ, layer-specialised only
// (ahead of time for the BASELINE layers, at run time otherwise).  Anything else - CX blocks, no compiler and no cached
// object, more than ZYZ_MAX_BLOCKS blocks - stays on the dense kernel.  CPF_ZYZ=0 switches it off (tests, A/B).
int zyz_sides(cpf_plan* plan, int which) {          // `which`: the labelling select_labelling() has put into plan->h_place
    DevBuf& buf = plan->zyz_sides[which];
    if (buf.p) return 0;
    const int K = plan->K, n = plan->n;
    std::vector<cpf::ZyzSideInfo> h((size_t)(2 * K + n));
    std::vector<int> last((size_t)n, -1);
    for (auto& e : h) { e.prev = -1; e.qubit = 0; e.pad0 = e.pad1 = 0; for (auto& w : e.later) w = 0; }
    for (int b = 0; b < K; b++)
        for (int s = 0; s < 2; s++) {
            const int w = 2 * b + s, q = plan->h_place[w];
            h[w].prev = last[q]; h[w].qubit = q;
            for (int p = last[q]; p >= 0; p = h[p].prev) h[p].later[w >> 6] |= 1ull << (w & 63);       // every earlier side on q sees w as a later one
            h[2 * K + q].later[w >> 6] |= 1ull << (w & 63);
            last[q] = w;
        }
    for (int q = 0; q < n; q++) { h[2 * K + q].prev = last[q]; h[2 * K + q].qubit = q; }
    CPF_HIP(buf.alloc(h.size() * sizeof(cpf::ZyzSideInfo)));
    CPF_HIP(hipMemcpy(buf.p, h.data(), h.size() * sizeof(cpf::ZyzSideInfo), hipMemcpyHostToDevice));
    return 0;
}
cpf_jit::Spec zyz_jit_spec(int nq, int lw, int rot, const std::vector<int>& ud) {
    cpf_jit::Spec spec{"cpf_zyz.h", cpf_jit::body_expr("zyz_body", {nq, lw, rot}, ud), 64 << lw};
    spec.arg = "cpf::ZyzArgs";
    spec.min_waves = 2;
    spec.flags = {CPF_DENSE_MLLVM_FLAGS};
    return spec;
}
bool zyz_wanted(const cpf_plan* plan, int mode) {
    const TuneVal e = tuning("CPF_ZYZ");
    if (e && e[0] == '0') return false;
    return dense_wanted(plan, mode) && dense_specialised(plan) && plan->entangler != CPF_ENTANGLER_CX && plan->K <= cpf::ZYZ_MAX_BLOCKS;
}
bool try_zyz(cpf_plan* plan, const cpf::KArgs& a, int64_t S, hipStream_t stream, int* rc) {
    if (!zyz_wanted(plan, a.mode)) return false;
    const int nq = plan->NQ, lw = dense_lw(plan, S), wps = 1 << lw, threads = 64 << lw, rot = dense_rot_code(plan);
    const bool have_aot = cpf_zyz_has_aot(nq, lw, rot, plan->layer_ud);
    const cpf_jit::Kernel* jk = nullptr;
    if (!have_aot) {
        if (!plan->use_jit) return false;
        jk = &cpf_jit::get(plan->device, zyz_jit_spec(nq, lw, rot, plan->layer_ud));
        if (!jk->fn) { plan->jit_note = jk->error; return false; }      // the dense kernel's own chain reports a forbidden compile
    }
    const size_t lds = (size_t)cpf::ZyzLds(a.P, a.G, plan->K, nq, wps).total * sizeof(float);
    if (lds > (size_t)plan->lds_per_cu) return false;
    DevBuf& tab = plan->dense_tab[lw];
    switch (nq * 8 + lw) {
        case 6 * 8 + 1: *rc = dense_table<cpf::DLay<6, 1>>(plan, tab); break;
        case 6 * 8 + 2: *rc = dense_table<cpf::DLay<6, 2>>(plan, tab); break;
        case 5 * 8 + 0: *rc = dense_table<cpf::DLay<5, 0>>(plan, tab); break;
        case 5 * 8 + 1: *rc = dense_table<cpf::DLay<5, 1>>(plan, tab); break;
        case 4 * 8 + 0: *rc = dense_table<cpf::DLay<4, 0>>(plan, tab); break;
        default: return false;
    }
    if (*rc) return true;
    if ((*rc = zyz_sides(plan, 0))) return true;
    float* scratch = nullptr;
    if ((*rc = scratch_for(plan, stream, (size_t)S * (size_t)plan->K * cpf::ZYZ_TAB_FLOATS * sizeof(float), &scratch))) return true;
    const cpf::ZyzArgs za{a, tab.as<cpf::DenseBlock>(), plan->zyz_sides[0].as<cpf::ZyzSideInfo>(), plan->K, plan->n_rot, scratch};
    const std::string waves = " waves=" + std::to_string(wps);
    hipError_t e = hipSuccess;
    if (have_aot) {
        cpf_zyz_launch_aot(nq, lw, rot, plan->layer_ud, za, S, lds, stream, &e);
        plan->variant = "zyz/aot" + waves;
    } else {
        if (!module_lds_opt_in(jk->fn, lds)) { plan->jit_note = "dynamic LDS opt-in refused for the ZYZ-form kernel"; return false; }
        cpf::ZyzArgs args = za;
        void* params[] = {&args};
        e = hipModuleLaunchKernel(jk->fn, (unsigned)S, 1, 1, (unsigned)threads, 1, 1, (unsigned)lds, stream, params, nullptr);
        plan->variant = "zyz/jit(" + jk->backend + ")" + waves;
    }
    *rc = e == hipSuccess ? 0 : fail((int)e, "ZYZ-form kernel launch failed: %s", hipGetErrorString(e));
    return true;
}

// ---- ZYZ form on the gate-by-gate layout (cpf_zyzl.h): one wave per sample, 4 and 5 qubits, CP or CZ blocks,
// layer-specialised (ahead of time for config 4's relabelled chain, at run time otherwise).  CPF_ZYZL=0|1 switches it.
int zyzl_slots(const cpf_plan* plan, int lw) { const int th = 64 << lw; return (2 * plan->K + plan->n + th - 1) / th; }      // items (sides + qubits) per lane
cpf_jit::Spec zyzl_jit_spec(int nq, int lw, int rot, int nslot, const std::vector<int>& ud) {
    cpf_jit::Spec spec{"cpf_zyzl.h", cpf_jit::body_expr("zyzl_body", {nq, lw, rot, nslot}, ud), 256};      // up to four one-wave samples per workgroup
    spec.arg = "cpf::ZyzlArgs";
    spec.min_waves = 1;
    spec.flags = {CPF_DENSE_MLLVM_FLAGS};
    return spec;
}
bool zyzl_wanted(const cpf_plan* plan, int mode, int lw) {
    if (plan->force_generic || mode != cpf::MODE_OPTIMIZE || lw < 0 || lw > 2 || plan->n != plan->NQ || plan->NQ < 3 || plan->NQ > 5) return false;
    if (lw > 0 && plan->NQ == 3) return false;          // two / four waves per sample: 4 and 5 qubits (the columns split over the waves)
    if (2 * plan->K + plan->n > 256) return false;          // side / qubit items: four 64-bit words of flip bits and of "later sides" masks
    if (plan->K < 1 || zyzl_slots(plan, lw) > 4 || plan->n_rot < 1 || plan->n_rot > 3 || plan->entangler == CPF_ENTANGLER_CX) return false;
    if (plan->layer_ud.empty() || plan->layer_ud.size() > 32) return false;
    const TuneVal e = tuning("CPF_ZYZL");
    // default: one wave per sample on 4 and 5 qubits; four waves per sample on 5 qubits (same box, 100 / 256 samples: chain 36 blocks
    // 33.4 / 33.5 ms against 42.3 / 42.7 on the gate-by-gate kernel, 86 blocks 67.6 / 68.1 against 93.8 / 94.5; the 4-qubit star,
    // every row qubit on lanes, 17.4 against 16.2: stays).  3 qubits: on request (the A/B of DESIGN.md section 4)
    return e ? e[0] != '0' : (lw == 0 ? plan->NQ >= 4 : (lw == 2 && plan->NQ == 5));
}
// Layers whose blocks' input states the forward sweep keeps in registers (cpf_zyzl.h SAVE: the backward sweep does not un-apply those
// blocks from psi) - CPF_ZYZL_AOT_SAVE layers, what the registers hold - and, in front of them, CPF_ZYZL_AOT_SAVEL layers in LDS.
// Shipped instances only - config 4's shape with at least that many whole layers - so that no run-time built shape gains a variant;
// CPF_ZYZL_SAVE=0 selects the instance without (parity tests run both).
int zyzl_saved_layers(const cpf_plan* plan, int lw, int rot, int nslot) {
    const int LL = (int)plan->layer_ud.size() / 2;
    if (lw != 0 || plan->NQ != 4 || nslot != 1 || LL < 1 || plan->K % LL != 0 || plan->K / LL < CPF_ZYZL_AOT_SAVE + CPF_ZYZL_AOT_SAVEL) return 0;
    { const TuneVal e = tuning("CPF_ZYZL_SAVE"); if (e && e[0] == '0') return 0; }
    return cpf_zyzl_has_aot(plan->NQ, rot, CPF_ZYZL_AOT_KEPT_CODE, plan->layer_ud) ? CPF_ZYZL_AOT_SAVE : 0;
}
bool try_zyzl(cpf_plan* plan, const cpf::KArgs& a_in, int lw, int64_t S, hipStream_t stream, int* rc) {
    if (!zyzl_wanted(plan, a_in.mode, lw)) return false;
    const int nq = plan->NQ, rot = dense_rot_code(plan), which = labelling_for(plan, a_in.mode, lw);
    const int nslot1 = zyzl_slots(plan, lw), save = zyzl_saved_layers(plan, lw, rot, nslot1), nslot = save ? CPF_ZYZL_AOT_KEPT_CODE : nslot1;
    const int keptl_states = save ? CPF_ZYZL_AOT_SAVEL * (int)(plan->layer_ud.size() / 2) : 0;
    const bool have_aot = lw == 0 && cpf_zyzl_has_aot(nq, rot, nslot, plan->layer_ud);
    const cpf_jit::Kernel* jk = nullptr;
    if (!have_aot) {
        if (!plan->use_jit) return false;
        jk = &cpf_jit::get(plan->device, zyzl_jit_spec(nq, lw, rot, nslot, plan->layer_ud));
        if (!jk->fn) {
            plan->jit_note = jk->error;
            if (jk->error.find("forbidden") != std::string::npos) { *rc = fail(-30, "%s", jk->error.c_str()); return true; }
            return false;
        }
    }
    const size_t lds1 = (size_t)cpf::ZyzlLds(plan->K, nq, 1 << lw, keptl_states).total * sizeof(float);
    if (lds1 > (size_t)plan->lds_per_cu) return false;
    cpf::KArgs a = a_in;
    a.group = lw == 0 ? layered_group(plan, lw, S, lds1) : 1;
    if (nq == 3 && S > plan->num_cu) a.group = 4;          // four waves of a workgroup land on the CU's four SIMDs (cpf_layered.h)
    const size_t lds = lds1 * a.group;
    if ((*rc = zyz_sides(plan, which))) return true;
    const cpf::ZyzlArgs za{a, plan->zyz_sides[which].as<cpf::ZyzSideInfo>(), plan->K, plan->n_rot};
    const unsigned grid = (unsigned)((S + a.group - 1) / a.group), block = lw == 0 ? 64u * a.group : 64u << lw;
    hipError_t e = hipSuccess;
    const std::string tail = " waves=" + std::to_string(1 << lw) + (a.group > 1 ? " group=" + std::to_string(a.group) : "");
    if (have_aot) {
        cpf_zyzl_launch_aot(nq, rot, nslot, plan->layer_ud, za, grid, block, lds, stream, &e);
        plan->variant = "zyzl/aot" + tail + (save ? " kept=" + std::to_string(save) + (CPF_ZYZL_AOT_SAVEL ? "+" + std::to_string(CPF_ZYZL_AOT_SAVEL) : std::string()) : "");
    } else {
        if (!module_lds_opt_in(jk->fn, lds)) { plan->jit_note = "dynamic LDS opt-in refused for the ZYZ-form kernel"; return false; }
        cpf::ZyzlArgs args = za;
        void* params[] = {&args};
        e = hipModuleLaunchKernel(jk->fn, grid, 1, 1, block, 1, 1, (unsigned)lds, stream, params, nullptr);
        plan->variant = "zyzl/jit(" + jk->backend + ")" + tail;
    }
    *rc = e == hipSuccess ? 0 : fail((int)e, "ZYZ-form kernel launch failed: %s", hipGetErrorString(e));
    return true;
}

// ---- runtime-specialised kernels (cpf_jit.h).  Which kernels a (plan, S, lw) launch would try, in order of
// preference: used by the launch path and by cpf_jit_precompile (which builds them without a GPU).
struct JitCandidate { cpf_jit::Spec spec; size_t lds = 0; std::string tag; int group = 1, block = 0; };   // block: threads per workgroup of the launch (0: spec.threads)

int ppt_bucket(int ppt) {          // angles per owner thread, rounded up to a few sizes so that adaptive searches over K share kernels
    for (int b : {1, 2, 3, 4, 6, 8, 12}) if (ppt <= b) return b;
    return ppt;
}

// k_mode: 0 = block count from the plan, 1 = one build per block count, -1 = as CPF_JIT_K says (opt-in);
// only_split: the two-ended 3-qubit candidate alone.
std::vector<JitCandidate> jit_candidates(cpf_plan* plan, int P, int loss_kind, int lw, int64_t S, int k_mode = -1, bool only_split = false) {
    std::vector<JitCandidate> out;
    plan->jit_note.clear();
    // CP blocks; CZ blocks as CP(pi) with a constant record slot; CX blocks as a row permutation in the gate-by-gate sweeps
    // (cpf_blocks.h rot_ent)
    if ((plan->n != plan->NQ && plan->NQ != 3) || plan->K < 1) return out;     // n < NQ: a 2-qubit problem on the 3-qubit kernels
    if (plan->n_rot < 1 || plan->n_rot > 3 || plan->layer_ud.empty() || plan->layer_ud.size() > 32) return out;
    const int threads = 64 << lw, ppt = ppt_bucket((P + threads - 1) / threads);
    if (ppt > 12) { plan->jit_note = "too many angles per thread"; return out; }
    int kinds[3] = {0, 0, 0};
    for (int i = 0; i < plan->n_rot; i++) kinds[i] = plan->rot[i] == 'x' ? cpf::OP_RX : (plan->rot[i] == 'y' ? cpf::OP_RY : cpf::OP_RZ);
    const int rot_code = cpf::rot_with_ent(cpf::rot_pack(plan->n_rot, kinds[0], kinds[1], kinds[2]), plan->entangler == CPF_ENTANGLER_CZ ? 1 : (plan->entangler == CPF_ENTANGLER_CX ? 2 : 0));
    cpf::KArgs a{};
    a.loss_kind = loss_kind;
    const TuneVal jk = tuning("CPF_JIT_K");                                        // opt-in: one build per block count
    const bool jit_k = k_mode < 0 ? (jk && jk[0] == '1') : k_mode == 1;
    if (plan->NQ == 3 && lw == 0 && want_split(plan, a, plan->K, (int)plan->layer_ud.size() / 2, rot_code, S, jit_k)) {
        const int sppt = ppt_bucket((P + 127) / 128);
        const size_t slds2 = (size_t)cpf::SplitLds::total(plan->K, 2, 2 * (plan->n_rot - 1)) * sizeof(float);
        const bool states = states_fit(plan, S, rot_code, slds2, (size_t)plan->lds_per_cu);
        int smstore = (states ? 2 : 1) + (jit_k ? 4 * plan->K : 0);
        // one / two samples per CU: the K-specialised kernel with the balanced cut (MSTORE bit 17), at two per CU with both
        // samples in one workgroup (bit 16) - the forms of launch_split3, same choice
        const int64_t per_cu = (S + plan->num_cu - 1) / plan->num_cu;
        int pairs = !jit_k ? 0 : (per_cu == 2 ? 2 : (per_cu == 1 ? 1 : 0));
        { const TuneVal e = tuning("CPF_PAIRS"); if (e && jit_k) pairs = e[0] == '2' ? 2 : (e[0] == '1' ? 1 : 0); }      // A/B and tests: 0 = the saturated form
        if (pairs == 2 && (S < 2 || cpf::rot_nr(rot_code) < 2 ||
                           ((S + 1) / 2 + plan->num_cu - 1) / plan->num_cu * (int64_t)(2 * slds2) > plan->lds_per_cu)) pairs = per_cu == 1 ? 1 : 0;
        if (pairs == 1 && !states) pairs = 0;
        JitCandidate c;
        if (pairs == 2) {
            smstore = 2 + 4 * plan->K + (3 << 16);
            c.group = 2; c.block = 256; c.lds = 2 * slds2;
        } else {
            if (pairs == 1) smstore += 2 << 16;
            c.lds = states ? slds2 : (size_t)cpf::SplitLds::total(plan->K, 1) * sizeof(float);
        }
        c.spec = {"cpf_split.h", cpf_jit::body_expr("split_body", {rot_code, sppt, smstore}, plan->layer_ud), pairs == 2 ? 256 : 128};
        c.tag = std::string((smstore & 3) == 2 ? "split/jit(%s) waves=2 states" : "split/jit(%s) waves=2") + (jit_k ? " K" : "") +
                (pairs == 2 ? " pairs=2" : (pairs == 1 ? " lone" : ""));
        if ((int64_t)c.lds <= plan->lds_per_cu) out.push_back(c);
    }
    if (only_split) return out;
    int mstore = 0;
    if (plan->NQ == 3 && cpf::layered_merge(3, lw, rot_code)) {
        const size_t with = (size_t)cpf::LayeredLds<3>::total(plan->K, 1, true) * sizeof(float);
        mstore = (with <= 64 * 1024 && phases_fit(plan, S, with)) ? 1 : 0;
    }
    size_t lds = 0;
    switch (plan->NQ) {
        case 3: lds = cpf::LayeredLds<3>::total(plan->K, 1 << lw, mstore != 0); break;
        case 4: lds = cpf::LayeredLds<4>::total(plan->K, 1 << lw, false); break;
        case 5: lds = cpf::LayeredLds<5>::total(plan->K, 1 << lw, false); break;
        case 6: lds = cpf::LayeredLds<6>::total(plan->K, 1 << lw, false); break;
        default: return out;
    }
    lds *= sizeof(float);
    if (lds > 64 * 1024) { plan->jit_note = "LDS footprint above 64 KiB"; return out; }
    JitCandidate c;
    c.group = layered_group(plan, lw, S, lds);
    c.lds = lds * c.group;
    c.block = threads * c.group;
    c.spec = {"cpf_layered.h", cpf_jit::body_expr("layered_body", {plan->NQ, lw, rot_code, ppt, mstore}, plan->layer_ud), lw == 0 ? 256 : threads};   // launch bound: up to four one-wave samples per workgroup
    c.tag = std::string("layered/jit(%s) waves=") + (lw ? "4" : "1");
    out.push_back(c);
    return out;
}

std::string tag_with_backend(const std::string& tag, const std::string& backend) {
    std::string t = tag;
    const size_t k = t.find("%s");
    if (k != std::string::npos) t.replace(k, 2, backend);
    return t;
}

// cpf_jit_precompile: an object that was built must have been written under $CPF_JIT_CACHE when that is set (a rejected
// directory would otherwise leave the shipped cache empty while the build reports success)
int precompile_landed(const cpf_jit::Code& code, const std::string& body) {
    const char* want = getenv("CPF_JIT_CACHE");
    if (!code.compiled || !want) return 0;
    if (code.path.rfind(std::string(want) + "/", 0) == 0) return 0;
    return fail(-32, "precompile of %s: object written to %s, not under CPF_JIT_CACHE=%s%s", body.c_str(), code.path.c_str(), want, code.note.c_str());
}

int launch_jit(cpf_plan* plan, const cpf::KArgs& a, const JitCandidate& c, const cpf_jit::Kernel& k, int64_t S, hipStream_t stream) {
    cpf::KArgs args = a;
    args.group = c.group;
    if (!module_lds_opt_in(k.fn, c.lds)) return fail(-20, "%zu B of dynamic LDS refused for a run-time built kernel", c.lds);
    void* params[] = {&args};
    const hipError_t e = hipModuleLaunchKernel(k.fn, (unsigned)((S + c.group - 1) / c.group), 1, 1, (unsigned)(c.block ? c.block : c.spec.threads), 1, 1,
                                               (unsigned)c.lds, stream, params, nullptr);
    plan->variant = tag_with_backend(c.tag, k.backend);
    return e == hipSuccess ? 0 : fail((int)e, "hipModuleLaunchKernel failed: %s", hipGetErrorString(e));
}

// ---- two-ended ZYZ form, 3 qubits (cpf_zyz3.h): block count, cut, samples per workgroup and layer are compile-time constants of
// every instance.  The cut M = blocks swept by wave A, which also owns the surface gates and the penalty; wave B owns the blocks'
// angles.  Same box, ms by cut (config 2's shape, 12 blocks, unless named; profiles/r05_cfg2_levers_ab.txt):
//   one sample per CU      256 samples: 4.12 / 3.93 / 3.80 at 4 / 5 / 6; 100 samples of 14 blocks 4.39 / 4.12 / 4.11 at 5 / 6 / 7
//                                                                                                        -> K / 2
//   two samples per CU     (round 6's kernel: one wave on every SIMD, nothing shares it) config 2's shape at 512 samples 3.88 / 3.94 / 4.09 at
//                          3 / 4 / 5; config 3's shape (16 blocks) at 512: 4.94 / 4.73 / 4.31 / 4.35 at 2 / 3 / 4 / 5 (768 samples of
//                          config 2's shape: 3.93 / 3.95 at 3 / 4)                                          -> 0.25 K, rounded
//   more, all resident     (round 6: 1024 samples 4.06 / 4.04 / 4.00 at 2 / 3 / 4; config 3's shape at 1024: 4.48 / 4.44 / 4.48 at 4 / 5 / 6)
//                          round 5 - 1024: 4.31 / 4.23 at 3 / 4 (before wave A's tracking moved behind its gradient half: 4.48 / 4.37 / 4.46 /
//                          4.57 at 2 / 3 / 4 / 5); config 3's shape (16 blocks) 512: 4.66 / 4.71 at 5 / 6     -> 0.33 K, rounded
//   in rounds              2048: 10.72 / 10.54 at 4 / 5; (before the two-lane owner layout) 4096: 22.70 / 21.75 at 4 / 5, config 3's
//                          shape 4096: 23.2 / 22.6 at 6 / 7                                              -> 0.43 K
// Two samples per workgroup sharing the owner of the block angles (PAIRS = 2) paid at two samples per CU while an owner lane carried
// all four angle slots of its side (4.35 against 4.70 ms); with two lanes per side (cpf_zyz3.h S2, one sample per workgroup only) it
// loses (4.46 against 4.28) and is a switch only.
// CPF_ZYZ3_M overrides the cut, CPF_ZYZ3_PAIRS the samples per workgroup, CPF_ZYZ3_RES the workgroups resident per CU.
int64_t zyz3_per_cu(const cpf_plan* plan, int64_t S) { return (S + plan->num_cu - 1) / plan->num_cu; }
int64_t zyz3_max_resident(const cpf_plan* plan) { return std::min<int64_t>(16, plan->lds_per_cu / ((int64_t)cpf::zyz3_lds_floats(plan->K) * 4)); }
bool zyz3_all_resident(const cpf_plan* plan, int64_t S) { return zyz3_per_cu(plan, S) <= zyz3_max_resident(plan); }
int zyz3_pairs(const cpf_plan* plan, int64_t S) {
    (void)S;
    int pairs = 1;
    { const TuneVal e = tuning("CPF_ZYZ3_PAIRS"); if (e) pairs = (e[0] == '2' && 2 * plan->K <= 32) ? 2 : 1; }
    if ((int64_t)2 * cpf::zyz3_lds_floats(plan->K) * 4 > plan->lds_per_cu) pairs = 1;
    return pairs;
}
int zyz3_cut(const cpf_plan* plan, int64_t S) {
    const int K = plan->K;
    const int64_t per_cu = zyz3_per_cu(plan, S);
    int m = per_cu <= 1 ? K / 2 : (zyz3_pairs(plan, S) == 2 ? K / 2 : (per_cu == 2 ? (25 * K + 50) / 100 : (zyz3_all_resident(plan, S) ? (33 * K + 50) / 100 : (43 * K + 50) / 100)));
    { const TuneVal e = tuning("CPF_ZYZ3_M"); if (e) m = atoi(e.s.c_str()); }
    return std::max(1, std::min(K - 1, m));
}
// Dynamic LDS per workgroup: what the kernel needs.  CPF_ZYZ3_RES=r (A/B) asks for more, so that only r workgroups are resident per
// CU: evening the residency out over the rounds of an oversubscribed launch was measured and lost to simply keeping every slot full
// (config 2's shape, 7 workgroups fit per CU: 2048 samples 10.3 ms at 7 + 1 against 12.1 at 4 + 4; 4096 samples 18.4 at 7 + 7 + 2
// against 18.6 at 6 + 6 + 4 and 23.4 at 4 x 4).
size_t zyz3_lds_request(const cpf_plan* plan, int64_t S, int pairs) {
    (void)S;
    const int64_t lds1 = (int64_t)cpf::zyz3_lds_floats(plan->K) * 4 * pairs;
    int64_t res = 0;
    { const TuneVal e = CPF_DIAG_TUNING("CPF_ZYZ3_RES"); if (e) res = atoi(e.s.c_str()); }
    if (res <= 0 || res * lds1 > plan->lds_per_cu) return (size_t)lds1;
    return (size_t)std::max<int64_t>(lds1, (plan->lds_per_cu / res - 1024) & ~(int64_t)255);      // (not the whole share: a CU keeps a little LDS back)
}
cpf_jit::Spec zyz3_jit_spec(int rot, int K, int M, int pairs, const std::vector<int>& ud) {
    cpf_jit::Spec spec{"cpf_zyz3.h", cpf_jit::body_expr("zyz3_body", {rot, K, M, pairs}, ud), 128 * pairs};
#ifdef CPF_DIAGNOSTICS
    { const TuneVal e = CPF_DIAG_TUNING("CPF_ZYZ3_STAMPS"); if (e && e[0] == '1') spec.flags = {"-DCPF_DIAGNOSTICS", "-DCPF_STAMPS"}; }      // per-phase cycle sums (tools/gpu.py stamps)
    {   // A/B builds: extra -D flags, separated by spaces
        const TuneVal e = CPF_DIAG_TUNING("CPF_ZYZ3_DEFS");
        size_t i = 0;
        if (e && spec.flags.empty()) spec.flags.push_back("-DCPF_DIAGNOSTICS");
        while (e && i < e.s.size()) {
            size_t j = e.s.find(' ', i);
            if (j == std::string::npos) j = e.s.size();
            if (j > i) spec.flags.push_back(e.s.substr(i, j - i));
            i = j + 1;
        }
    }
#endif
    return spec;
}
// The kernel serves 3-qubit problems with 1..3 rotation letters, CP or CZ blocks, the |T|^2 losses and up to 32 blocks, and is the
// default for all of them (29 KB of LDS per sample at 12 blocks, 37 KB at 16: config 2's shape at 2048 / 4096 samples 11.2 / 21.5 ms
// against 12.3 / 24.3 on the gate-by-gate two-ended kernel, config 3's at 1024 / 4096 samples 5.55 / 26.7 against 7.55 / 29.8).
// A launch that selects another 3-qubit family through its switches (CPF_SPLIT, CPF_PSTORE, ... - the A/B tools and the tests of
// those kernels) keeps that family.
bool zyz3_serves(const cpf_plan* plan, int mode, int loss_kind) {          // what the kernel can run
    if (plan->force_generic || mode != cpf::MODE_OPTIMIZE || plan->NQ != 3 || plan->n != 3 || loss_kind == 2) return false;
    if (!cpf::zyz3_supported(plan->K, plan->n_rot) || plan->entangler == CPF_ENTANGLER_CX) return false;
    if (plan->layer_ud.empty() || plan->layer_ud.size() > 32) return false;
    return (int64_t)cpf::zyz3_lds_floats(plan->K) * 4 <= plan->lds_per_cu;
}
bool zyz3_selected() {                                                     // what the tuning switches say
    { const TuneVal e = tuning("CPF_ZYZ3"); if (e) return e[0] == '1'; }
    for (const char* other : {"CPF_SPLIT", "CPF_PSTORE", "CPF_MSTORE", "CPF_KSPEC", "CPF_PAIRS", "CPF_LONE", "CPF_ZYZL", "CPF_JIT_K"})
        if (tuning(other)) return false;
    if (CPF_DIAG_TUNING("CPF_NO_AOT3")) return false;
    return true;
}
bool zyz3_wanted(const cpf_plan* plan, const cpf::KArgs& a) { return zyz3_serves(plan, a.mode, a.loss_kind) && zyz3_selected(); }
bool try_zyz3(cpf_plan* plan, const cpf::KArgs& a, int64_t S, hipStream_t stream, int* rc) {
    if (!zyz3_wanted(plan, a)) return false;
    const int M = zyz3_cut(plan, S), rot = dense_rot_code(plan), pairs = zyz3_pairs(plan, S);
    const size_t lds = zyz3_lds_request(plan, S, pairs);
    const std::string tail = " waves=2 K cut=" + std::to_string(M) + (pairs == 2 ? " pairs=2" : "");
    const TuneVal defs = CPF_DIAG_TUNING("CPF_ZYZ3_DEFS"), stamps = CPF_DIAG_TUNING("CPF_ZYZ3_STAMPS");
    cpf::KArgs args = a;
    args.prio = zyz3_all_resident(plan, S);                  // every workgroup resident: wave B gets the higher issue priority
    { const TuneVal p = tuning("CPF_PRIO"); if (p) args.prio = p[0] == '1'; }
    // ... and from three samples per CU on (waves of different samples share SIMDs) the workgroups start their step loops four-fold out
    // of phase (cpf_zyz3.h CPF_Z3_STAGGER: 2048 waves in lock-step through dense and sparse phases make the clock dip; sustained means of
    // 150 launches: config 2 4.06 -> 3.98 ms; 1536 samples, and 512 / 256 where the bit is not set, unchanged; config 3's 512: +0.6 % if set)
    if (zyz3_all_resident(plan, S) && zyz3_per_cu(plan, S) >= 3) args.prio |= 2;
    { const TuneVal p = tuning("CPF_PRIO"); if (p) args.prio = atoi(p.s.c_str()) & 3; }      // A/B: the whole mask (1 = priority only, 2 = stagger only, 3 = both)
    if (!defs && !stamps) {
        hipError_t e = hipSuccess;
        if (cpf_zyz3_launch_aot(rot, plan->K, M, pairs, plan->layer_ud, args, S, lds, stream, &e)) {
            plan->variant = "zyz3/aot" + tail;
            *rc = e == hipSuccess ? 0 : fail((int)e, "two-ended ZYZ-form kernel launch failed: %s", hipGetErrorString(e));
            return true;
        }
    }
    if (!plan->use_jit) return false;
    // an optional kernel: where compiling is forbidden (CPF_JIT_FORBID_COMPILE=1) a shape that is not in a cache is a miss, and the
    // gate-by-gate kernels follow
    const char* forbid = getenv("CPF_JIT_FORBID_COMPILE");
    const cpf_jit::Kernel& k = cpf_jit::get(plan->device, zyz3_jit_spec(rot, plan->K, M, pairs, plan->layer_ud), forbid && forbid[0] == '1');
    if (!k.fn) { if (k.error != "not cached") plan->jit_note = k.error; return false; }
    if (!module_lds_opt_in(k.fn, lds)) { plan->jit_note = "dynamic LDS opt-in refused for the two-ended ZYZ-form kernel"; return false; }
    void* params[] = {&args};
    const hipError_t e = hipModuleLaunchKernel(k.fn, (unsigned)((S + pairs - 1) / pairs), 1, 1, 128 * pairs, 1, 1, (unsigned)lds, stream, params, nullptr);
    plan->variant = "zyz3/jit(" + k.backend + ")" + tail;
    *rc = e == hipSuccess ? 0 : fail((int)e, "two-ended ZYZ-form kernel launch failed: %s", hipGetErrorString(e));
    return true;
}

// 3 qubits: a two-ended kernel specialised on the block count, when cpflow_amd/jit_cache/ (or the private cache) holds
// one for this shape - python -m cpflow_amd.build --precompile ships them for the block counts an adaptive search over
// the reference's 3-qubit layers visits.  Nothing is compiled here: a miss falls through to the kernels that read the
// block count from the plan.  The tutorial's shape (chain_layer(3), 14 blocks, 100 samples): see DESIGN.md section 4.
bool try_split3_cached_k(cpf_plan* plan, const cpf::KArgs& a, int64_t S, hipStream_t stream, int* rc) {
    if (!plan->use_jit || plan->force_generic || a.mode != cpf::MODE_OPTIMIZE || plan->NQ != 3 || plan->n != 3 || !kspec_enabled()) return false;
    { const TuneVal e = CPF_DIAG_TUNING("CPF_NO_AOT3"); if (e && e[0] == '1') return false; }      // A/B: the plain runtime build
    const int LL = (int)plan->layer_ud.size() / 2;
    if (plan->rot == "xyz" && plan->entangler == CPF_ENTANGLER_CP && ((LL == 2 && plan->K == 12) || (LL == 3 && plan->K == 16)) &&
        (plan->layer_ud == std::vector<int>{0, 1, 1, 2} || plan->layer_ud == std::vector<int>{0, 1, 0, 2, 1, 2}))
        return false;                      // built into the library (launch_layered3), with its two-samples-per-workgroup form
    for (const JitCandidate& c : jit_candidates(plan, a.P, a.loss_kind, 0, S, 1, true)) {
        const cpf_jit::Kernel& k = cpf_jit::get(plan->device, c.spec, true);
        if (!k.fn) { if (k.error != "not cached") plan->jit_note = k.error; return false; }
        // an optional kernel: a runtime that refuses its LDS opt-in is a miss, the kernels that read K from the plan follow
        if (!module_lds_opt_in(k.fn, c.lds)) { plan->jit_note = "dynamic LDS opt-in refused for a shipped K-specialised kernel"; return false; }
        *rc = launch_jit(plan, a, c, k, S, stream);
        return true;
    }
    return false;
}

// lw = log2(waves per sample).  Returns true when a specialised kernel was launched (or a hard error is in *rc).
bool try_layered_jit(cpf_plan* plan, const cpf::KArgs& a, int lw, int64_t S, hipStream_t stream, int* rc) {
    if (!plan->use_jit || plan->force_generic || a.mode != cpf::MODE_OPTIMIZE) return false;
    for (const JitCandidate& c : jit_candidates(plan, a.P, a.loss_kind, lw, S)) {
        const cpf_jit::Kernel& k = cpf_jit::get(plan->device, c.spec);
        if (!k.fn) {
            plan->jit_note = k.error;
            if (k.error.find("forbidden") != std::string::npos) { *rc = fail(-30, "%s", k.error.c_str()); return true; }
            continue;          // next candidate, then the generic kernel
        }
        if (!module_lds_opt_in(k.fn, c.lds)) { plan->jit_note = "dynamic LDS opt-in refused for a run-time built kernel"; continue; }
        *rc = launch_jit(plan, a, c, k, S, stream);
        return true;
    }
    return false;
}

// waves per sample: 1 when there are enough samples to fill the SIMDs, else 4 (columns split over waves)
int choose_lw(const cpf_plan* plan, int64_t S) {
    const int64_t simds = (int64_t)plan->num_cu * 4;
    int lw = 0;
    // measured on MI355X: n=4 S=2048: 1 wave 9.9 ms vs 4 waves 17.0 ms; n=5 S=1000: 1 wave 22.4 ms vs 4 waves 29.6 ms
    switch (plan->NQ) {
        case 4: lw = (2 * S > simds) ? 0 : 2; break;       // round 3 (shears, merged Rz pairs, grouped launches): 520-900 samples 26-27 ms on one wave against 35-47 on four
        case 5: lw = (2 * S >= simds) ? 0 : 2; break;
        case 6: lw = 2; break;
        default: lw = 0;
    }
    // round 4: where the one-wave ZYZ-form kernel serves the shape (cpf_zyzl.h) one wave per sample wins from one sample per CU on
    // (same box, 4-qubit star, 23 blocks: 300 / 500 samples 20.3 ms against 24.5 / 25.1 on four waves; 200 samples 20.0 against 16.4;
    // 5-qubit chain, 36 blocks: 300 / 500 samples 65.4 against 73.0 / 74.3; 200 samples 65.0 against 42.6)
    if (plan->NQ == 4 && S > plan->num_cu && zyzl_wanted(plan, cpf::MODE_OPTIMIZE, 0)) lw = 0;
    // 5 qubits on the ZYZ form: four waves per sample up to two waves per SIMD's worth of samples (chain, 36 blocks, 512 samples:
    // 51.4 ms on four waves against 65.8 on one and 71.5 on two; 1000 samples 97.2 against 66.7), one wave above
    if (plan->NQ == 5 && zyzl_wanted(plan, cpf::MODE_OPTIMIZE, 0) && zyzl_wanted(plan, cpf::MODE_OPTIMIZE, 2)) lw = 2 * S <= simds ? 2 : 0;
    { const TuneVal e = tuning("CPF_LW"); if (e && plan->NQ > 3) lw = (e[0] == '0') ? 0 : ((e[0] == '1' && plan->NQ < 6 && zyzl_wanted(plan, cpf::MODE_OPTIMIZE, 1)) ? 1 : 2); }
    if (plan->NQ == 6) lw = 2;
    return lw;
}

// Variant choice: one workgroup per sample; more waves per sample when there are too few
// samples to fill 256 CUs x 4 SIMDs and the matrix is large enough to split by columns.
int launch(const cpf_plan* plan_c, const cpf::KArgs& a_in, int64_t S, hipStream_t stream) {
    cpf_plan* plan = const_cast<cpf_plan*>(plan_c);
    if (S <= 0) return 0;
    if (S > 0x7fffffffLL) return fail(-21, "num_samples too large");
    int rc = 0;
    const int lw = choose_lw(plan, S);
    cpf::KArgs a = a_in;
    a.nsamples = (int)S;
    a.lone = S <= plan->num_cu ? 1 : 0;
    { const TuneVal e = tuning("CPF_LONE"); if (e) a.lone = e[0] == '1'; }      // A/B and tests
    {
        const int which = labelling_for(plan, a.mode, lw);
        select_labelling(plan, which);
        apply_labelling(plan, which, a);
    }
    // 0. a shipped 3-qubit kernel specialised on the block count; 1. ahead-of-time layered kernel (3 qubits, the
    // reference's own layers); 2. runtime-specialised layered kernel; 3. generic per-gate kernel
    { const TuneVal e = tuning("CPF_ZYZL"); if (e && e[0] == '1' && plan->NQ == 3 && try_zyzl(plan, a, lw, S, stream, &rc)) return rc; }
    if (try_zyz3(plan, a, S, stream, &rc)) return rc;
    if (try_split3_cached_k(plan, a, S, stream, &rc)) return rc;
    if (try_layered3_aot(plan, a, S, stream, &rc)) return rc;
    if (try_zyz(plan, a, S, stream, &rc)) return rc;
    if (try_dense(plan, a, S, stream, &rc)) return rc;
    if (try_zyzl(plan, a, lw, S, stream, &rc)) return rc;
    if (try_layered_big_aot(plan, a, lw, S, stream, &rc)) { plan->variant = std::string("layered/aot waves=") + (lw ? "4" : "1"); return rc; }
    if (try_layered_jit(plan, a, lw, S, stream, &rc)) return rc;
    plan->variant = std::string("generic/aot waves=") + (lw ? "4" : "1") + (plan->jit_note.empty() ? "" : " (" + plan->jit_note + ")");
    switch (plan->NQ) {
        case 3: return launch_variant<3, 0>(a, S, stream);
        case 4: return lw == 0 ? launch_variant<4, 0>(a, S, stream) : launch_variant<4, 2>(a, S, stream);
        case 5: return lw == 0 ? launch_variant<5, 0>(a, S, stream) : launch_variant<5, 2>(a, S, stream);
        case 6: return launch_variant<6, 2>(a, S, stream);
        default: return fail(-2, "unsupported kernel width %d", plan->NQ);
    }
}

// The part of a plan that needs no device: sizes, rotation string, kernel width, the layer (shortest period of the
// placement sequence) the specialised kernels are built around.
void fill_plan_shape(const cpf_problem* pb, cpf_plan* pl) {
    pl->n = pb->n_qubits; pl->K = pb->num_blocks; pl->n_rot = pb->n_rot; pl->entangler = pb->entangler;
    pl->nb = 2 * pb->n_rot + (pb->entangler == CPF_ENTANGLER_CP ? 1 : 0);
    pl->P = 3 * pl->n + pl->nb * pl->K;
    pl->T = pb->num_iterations;
    pl->rot.assign(pb->rot_axes ? pb->rot_axes : "", (size_t)pb->n_rot);
    { const TuneVal fg = CPF_DIAG_TUNING("CPF_FORCE_GENERIC"); pl->force_generic = fg && fg[0] == '1'; }
    { const TuneVal fj = tuning("CPF_JIT"); pl->use_jit = !(fj && fj[0] == '0'); }
    pl->NQ = pl->n < 3 ? 3 : pl->n;     // 2-qubit problems run as U (x) I on the 3-qubit kernel
    // Qubit relabelling (4 and 5 qubits, gate-by-gate kernel, one wave per sample).  That layout keeps the kernel's LAST
    // row qubits in registers (cpf_device.h Lay: a rotation there is 2 packed instructions per amplitude, on a lane qubit 3,
    // and a CP touches only the amplitudes whose register bits are set), so the qubits the template uses most go last:
    // problem qubit q becomes kernel qubit qperm[q], stable by usage.  Everything that names a qubit (placements, the gate
    // program, the rows and columns of the target) is relabelled; parameters keep their order (a surface gate's record slot
    // is found through pgate).  chain_layer(4): (0, 1, 2, 3) -> (0, 2, 3, 1), config 4 56.5 -> 52.1 ms.  With four waves per
    // sample every row qubit sits on lanes and the problem's own labels are kept (the 4-qubit star at 500 samples: 25.2 ms
    // against 28.6 relabelled), hence the choice per launch.
    pl->has_relabel = (pl->NQ == 4 || pl->NQ == 5) && pl->n == pl->NQ;
    { const TuneVal e = tuning("CPF_RELABEL"); if (e && e[0] == '0') pl->has_relabel = false; }      // A/B and tests
    { const TuneVal e = tuning("CPF_DENSE"); if (e ? e[0] == '1' : pl->n_rot > 3) pl->has_relabel = false; }        // the dense layout has its own qubit moves
    for (int which = 0; which < 2; which++) {
        cpf_plan::Labelling& lb = pl->lab[which];
        lb.qperm.resize((size_t)pl->n);
        for (int q = 0; q < pl->n; q++) lb.qperm[q] = q;
        if (which == 1 && pl->has_relabel) {
            std::vector<int> use((size_t)pl->n, 0), order((size_t)pl->n);
            for (int q : pl->lab[0].layer_ud) use[q]++;       // over one layer, so that the labels do not depend on the block count
            for (int q = 0; q < pl->n; q++) order[q] = q;
            std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return use[x] < use[y]; });
            for (int k = 0; k < pl->n; k++) lb.qperm[order[k]] = k;
        }
        lb.h_place.resize(2 * (size_t)pl->K);
        for (size_t i = 0; i < lb.h_place.size(); i++) lb.h_place[i] = lb.qperm[pb->placements[i]];
        lb.layer_ud.clear();
        for (int L = 1; L <= pl->K && lb.layer_ud.empty(); L++) {
            bool periodic = true;
            for (int b = L; b < pl->K && periodic; b++)
                periodic = lb.h_place[2 * b] == lb.h_place[2 * (b % L)] && lb.h_place[2 * b + 1] == lb.h_place[2 * (b % L) + 1];
            if (periodic) lb.layer_ud.assign(lb.h_place.begin(), lb.h_place.begin() + 2 * L);
        }
    }
    if (pl->has_relabel && pl->lab[1].qperm == pl->lab[0].qperm) pl->has_relabel = false;          // nothing to gain (e.g. a ring)
    select_labelling(pl, 0);
}

}  // namespace

extern "C" {

int32_t cpf_version(void) { return CPF_VERSION; }

int32_t cpf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* cpf_last_error(void) { return g_err; }

int32_t cpf_set_tuning(const char* name, const char* value) {
    // The release library knows the kernel-FAMILY switches only - what the parity tests need to put two families of one shape side by
    // side.  Timing experiments and instrumented builds (phase stamps, extra -D flags handed to the run-time compiler, residency and
    // forced-generic switches) exist in a library built with -DCPF_DIAGNOSTICS only (tools/gpu.py, cpflow_amd.build.build_variant).
    static const char* const known[] = {"CPF_SPLIT", "CPF_PSTORE", "CPF_MSTORE", "CPF_KSPEC", "CPF_PAIRS", "CPF_LONE", "CPF_PRIO", "CPF_LW", "CPF_GROUP",
                                        "CPF_DENSE", "CPF_DENSE_LW", "CPF_DENSE_SPECIALISED", "CPF_DENSE_FOLD", "CPF_ZYZ", "CPF_ZYZL", "CPF_ZYZL_SAVE", "CPF_RELABEL",
                                        "CPF_JIT", "CPF_JIT_K", "CPF_ZYZ3", "CPF_ZYZ3_M", "CPF_ZYZ3_PAIRS",
#ifdef CPF_DIAGNOSTICS
                                        "CPF_FORCE_GENERIC", "CPF_NO_AOT3", "CPF_ZYZ3_RES", "CPF_ZYZ3_STAMPS", "CPF_ZYZ3_DEFS",
#endif
    };
    if (!name) return fail(-1, "cpf_set_tuning: name is NULL");
    for (const char* k : known)
        if (!strcmp(k, name)) {
            std::lock_guard<std::mutex> lock(tuning_mutex());
            if (value) tuning_map()[name] = value; else tuning_map().erase(name);
            return 0;
        }
    return fail(-1, "cpf_set_tuning: unknown switch '%s'", name);
}

int32_t cpf_plan_create(const cpf_problem* pb, int32_t device, cpf_plan** out_plan) {
    if (!out_plan) return fail(-1, "out_plan is NULL");
    *out_plan = nullptr;
    if (int rc = validate(pb)) return rc;
    CPF_HIP(hipSetDevice(device));
    cpf_plan* pl = new cpf_plan();
    pl->device = device;
    fill_plan_shape(pb, pl);
    const int n = pl->n, NQ = pl->NQ, N = 1 << n, D = 1 << NQ, P = pl->P;

    // ---- gate program (main.py:106-146): surface Rz Rx Rz per qubit, then per block
    //      entangler followed by R_g(up) R_g(down) for each letter g (main.py:69-82)
    pl->h_pkind.assign(P, 0);
    // perm[q] = the kernel's label of problem qubit q.  Surface gates are emitted in KERNEL qubit order, so that gate
    // 3 k + g acts on kernel qubit k (the specialised kernels rely on it) while carrying the parameter 3 q + g of the problem
    // qubit q it stands for; gates on different qubits commute.
    auto build_prog = [&](const std::vector<int>& perm, std::vector<int4>& prog, std::vector<int32_t>& pgate) {
        prog.clear();
        std::vector<int> inv((size_t)n);
        for (int q = 0; q < n; q++) inv[perm[q]] = q;
        for (int k = 0; k < n; k++) {
            const int q = inv[k];
            prog.push_back(make_int4(cpf::OP_RZ, k, 0, 3 * q + 0));
            prog.push_back(make_int4(cpf::OP_RX, k, 0, 3 * q + 1));
            prog.push_back(make_int4(cpf::OP_RZ, k, 0, 3 * q + 2));
        }
        for (int b = 0; b < pl->K; b++) {
            const int up = perm[pb->placements[2 * b]], dn = perm[pb->placements[2 * b + 1]], base = 3 * n + pl->nb * b;
            const int lo = up < dn ? up : dn, hi = up < dn ? dn : up;
            if (pb->entangler == CPF_ENTANGLER_CP) {
                prog.push_back(make_int4(cpf::OP_CP, lo, hi, base + pl->nb - 1));
                pl->h_pkind[base + pl->nb - 1] = 1;
            } else if (pb->entangler == CPF_ENTANGLER_CZ) {
                prog.push_back(make_int4(cpf::OP_CZ, lo, hi, 0));
            } else {
                prog.push_back(make_int4(cpf::OP_CX, up, dn, 0));       // control = up, target = down
            }
            for (int i = 0; i < pb->n_rot; i++) {
                const int kind = pb->rot_axes[i] == 'x' ? cpf::OP_RX : (pb->rot_axes[i] == 'y' ? cpf::OP_RY : cpf::OP_RZ);
                prog.push_back(make_int4(kind, up, 0, base + 2 * i));
                prog.push_back(make_int4(kind, dn, 0, base + 2 * i + 1));
            }
        }
        pgate.assign((size_t)P, 0);
        for (int g = 0; g < (int)prog.size(); g++)
            if (prog[g].x != cpf::OP_CZ && prog[g].x != cpf::OP_CX) pgate[prog[g].w] = g;
    };
    const int d = NQ - n;
    std::vector<int4> prog[2];
    std::vector<int32_t> pgate[2];
    std::vector<float2> tgt[2];
    for (int which = 0; which < (pl->has_relabel ? 2 : 1); which++) {
        const std::vector<int>& perm = pl->lab[which].qperm;
        build_prog(perm, prog[which], pgate[which]);
        // target, used as complex64, rows and columns in the kernel's qubit labels, embedded as V (x) I when the kernel is
        // wider than the problem
        tgt[which].assign((size_t)D * D, make_float2(0.f, 0.f));
        auto relabelled = [&](int r) {                        // qubit 0 = most significant bit (main.py:119)
            int k = 0;
            for (int q = 0; q < n; q++) k |= ((r >> (n - 1 - q)) & 1) << (n - 1 - perm[q]);
            return k;
        };
        for (int r = 0; r < N; r++)
            for (int c = 0; c < N; c++)
                for (int x = 0; x < (1 << d); x++)
                    tgt[which][(size_t)((relabelled(r) << d) | x) * D + ((relabelled(c) << d) | x)] =
                        make_float2((float)pb->target[2 * ((size_t)r * N + c)], (float)pb->target[2 * ((size_t)r * N + c) + 1]);
    }
    pl->h_prog = prog[0];
    pl->G = (int)prog[0].size();

    // ---- Adam bias corrections 1 - b^t in float32 (optax 0.1.1 bias_correction)
    std::vector<float2> bias((size_t)(pl->T > 0 ? pl->T : 1));
    const float b1f = (float)pb->b1, b2f = (float)pb->b2;
    for (int t = 1; t <= pl->T; t++)
        bias[t - 1] = make_float2(1.0f - powf(b1f, (float)t), 1.0f - powf(b2f, (float)t));
    // ... and their float32 reciprocals (the two-ended 3-qubit kernel multiplies: m / (1 - b1^t) as m * (1 / (1 - b1^t)), the
    // reciprocal correctly rounded here instead of v_rcp_f32's 1 ulp on the device)
    std::vector<float2> ibias(bias.size(), make_float2(1.0f, 1.0f));
    for (int t = 1; t <= pl->T; t++) ibias[t - 1] = make_float2(1.0f / bias[t - 1].x, 1.0f / bias[t - 1].y);

    hipError_t e = hipSuccess;
    auto up = [&](DevBuf& b, const void* src, size_t bytes) {
        if (e != hipSuccess) return;
        e = b.alloc(bytes);
        if (e == hipSuccess && bytes) e = hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice);
    };
    for (int which = 0; which < (pl->has_relabel ? 2 : 1); which++) {
        up(pl->lab[which].d_prog, prog[which].data(), prog[which].size() * sizeof(int4));
        up(pl->lab[which].d_pgate, pgate[which].data(), pgate[which].size() * sizeof(int32_t));
        up(pl->lab[which].d_target, tgt[which].data(), tgt[which].size() * sizeof(float2));
    }
    up(pl->d_pkind, pl->h_pkind.data(), (size_t)P);
    up(pl->d_bias, bias.data(), bias.size() * sizeof(float2));
    up(pl->d_ibias, ibias.data(), ibias.size() * sizeof(float2));
    if (e == hipSuccess) e = hipEventCreate(&pl->ev0);
    if (e == hipSuccess) e = hipEventCreate(&pl->ev1);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        delete pl;
        return fail((int)e, "plan upload failed: %s", hipGetErrorString(e));
    }
    pl->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (prop.maxSharedMemoryPerMultiProcessor >= 64 * 1024) pl->lds_per_cu = (int64_t)prop.maxSharedMemoryPerMultiProcessor;

    cpf::KArgs& a = pl->base;
    apply_labelling(pl, 0, a);
    a.G = pl->G; a.P = P; a.T = pl->T; a.mode = cpf::MODE_OPTIMIZE;
    a.n_actual = n;
    a.pkind = pl->d_pkind.as<uint8_t>();
    a.bias = pl->d_bias.as<float2>();
    a.ibias = pl->d_ibias.as<float2>();
    build_pieces(pb->penalty, a.pc);
    a.pen_kind = pb->penalty.kind;
    a.prio = 0;
    a.use_reg = (pb->use_reg && pb->entangler == CPF_ENTANGLER_CP) ? 1 : 0;
    a.r = (float)pb->r;
    a.lr = (float)pb->lr; a.b1 = b1f; a.b2 = b2f; a.eps = (float)pb->eps;
    a.omb1 = (float)(1.0 - pb->b1); a.omb2 = (float)(1.0 - pb->b2);
    a.thr_cp = (float)pb->threshold_cp; a.entry_loss = (float)pb->entry_loss;
    // cost_HST: 1 - |T|^2 / D^2.  State preparation: 1 - |<psi|U|0>|^2 is the same |T|^2 form with the state in
    // column 0 of the target (other columns zero) and unit scale.  Relative phase: 1 - sum |V|^2 |U|^2 / D.
    a.loss_kind = pb->loss_kind == CPF_LOSS_RELATIVE_PHASE ? 2 : 0;
    a.inv_dim2 = pb->loss_kind == CPF_LOSS_HST ? 1.0f / ((float)D * (float)D)
               : (pb->loss_kind == CPF_LOSS_STATE ? 1.0f / (float)(1 << (2 * d)) : 1.0f / (float)D);
    *out_plan = pl;
    return 0;
}

int32_t cpf_plan_destroy(cpf_plan* plan) {
    if (!plan) return 0;
    (void)hipSetDevice(plan->device);
    delete plan;
    return 0;
}

int32_t cpf_plan_num_params(const cpf_plan* plan) { return plan ? plan->P : fail(-1, "plan is NULL"); }
int32_t cpf_plan_num_gates(const cpf_plan* plan) { return plan ? plan->G : fail(-1, "plan is NULL"); }

int32_t cpf_plan_cp_mask(const cpf_plan* plan, uint8_t* out_mask) {
    if (!plan || !out_mask) return fail(-1, "NULL argument");
    memcpy(out_mask, plan->h_pkind.data(), (size_t)plan->P);
    return 0;
}

double cpf_plan_flops_per_sample_step(const cpf_plan* plan) {
    if (!plan) return 0.0;
    // F = 3 N^2 (32 K + 16 n) + 20 P
    const double N = (double)(1 << plan->n);
    return 3.0 * N * N * (32.0 * plan->K + 16.0 * plan->n) + 20.0 * plan->P;
}

int32_t cpf_initial_angles_device(const cpf_plan* plan, uint32_t seed, int64_t num_samples_total, int64_t first,
                                  int64_t count, int32_t cp_zero, float* d_angles, void* stream) {
    if (!plan) return fail(-1, "plan is NULL");
    if (count == 0) return 0;
    if (!d_angles) return fail(-1, "d_angles is NULL");
    if (first < 0 || count < 0 || first + count > num_samples_total || num_samples_total >= 0x7fffffffLL)
        return fail(-10, "bad sample range first=%lld count=%lld total=%lld", (long long)first, (long long)count,
                    (long long)num_samples_total);
    CPF_HIP(hipSetDevice(plan->device));
    const uint64_t total = (uint64_t)count * plan->P;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(cpf::k_initial_angles, dim3(blocks), dim3(256), 0, (hipStream_t)stream, seed,
                       (uint32_t)num_samples_total, (uint32_t)first, (uint32_t)count, (uint32_t)plan->P,
                       plan->d_pkind.as<uint8_t>(), cp_zero, d_angles);
    CPF_HIP(hipGetLastError());
    return 0;
}

int32_t cpf_optimize_device(const cpf_plan* plan_c, int64_t S, const float* d_init, const uint8_t* d_frozen,
                            float* d_params, float* d_regloss, float* d_loss, float* d_reg, int32_t* d_cz,
                            uint8_t* d_pass, float* d_hist_params, float* d_hist_loss, void* stream) {
    cpf_plan* plan = const_cast<cpf_plan*>(plan_c);
    if (!plan) return fail(-1, "plan is NULL");
    if (S == 0) return 0;
    if (!d_init || !d_params || !d_regloss || !d_loss || !d_reg || !d_cz || !d_pass) return fail(-1, "NULL buffer");
    CPF_HIP(hipSetDevice(plan->device));
    cpf::KArgs a = plan->base;
    a.mode = cpf::MODE_OPTIMIZE;
    { const TuneVal e = tuning("CPF_PRIO"); a.prio = e ? (e[0] == '1') : (S <= (int64_t)plan->num_cu * 8); }   // A/B: CPF_PRIO=0|1
    a.init = d_init; a.frozen = d_frozen;
    a.out_params = d_params; a.out_regloss = d_regloss; a.out_loss = d_loss; a.out_reg = d_reg;
    a.out_cz = d_cz; a.out_pass = d_pass;
    a.hist_params = d_hist_params; a.hist_loss = d_hist_loss;
    CPF_HIP(hipEventRecord(plan->ev0, (hipStream_t)stream));
    if (int rc = launch(plan, a, S, (hipStream_t)stream)) return rc;
    CPF_HIP(hipEventRecord(plan->ev1, (hipStream_t)stream));
    plan->timed = true;
    return 0;
}

int32_t cpf_plan_last_variant(const cpf_plan* plan, char* buf, size_t size) {
    if (!plan || !buf || !size) return fail(-1, "NULL argument");
    snprintf(buf, size, "%s", plan->variant.c_str());
    return 0;
}

int32_t cpf_plan_last_kernel_ms(cpf_plan* plan, float* out_ms) {
    if (!plan || !out_ms) return fail(-1, "NULL argument");
    if (!plan->timed) return fail(-11, "no optimize launch recorded on this plan");
    CPF_HIP(hipEventSynchronize(plan->ev1));
    CPF_HIP(hipEventElapsedTime(out_ms, plan->ev0, plan->ev1));
    return 0;
}

int32_t cpf_loss_grad_device(const cpf_plan* plan, int64_t S, const float* d_theta, float* d_loss, float* d_reg,
                             float* d_regloss, float* d_grad, void* stream) {
    if (!plan) return fail(-1, "plan is NULL");
    if (S == 0) return 0;
    if (!d_theta || !d_loss || !d_reg || !d_regloss) return fail(-1, "NULL buffer");
    CPF_HIP(hipSetDevice(plan->device));
    cpf::KArgs a = plan->base;
    a.mode = cpf::MODE_LOSS_GRAD;
    a.init = d_theta; a.frozen = nullptr;
    a.out_loss = d_loss; a.out_reg = d_reg; a.out_regloss = d_regloss; a.out_grad = d_grad;
    return launch(plan, a, S, (hipStream_t)stream);
}

int32_t cpf_unitary_device(const cpf_plan* plan, int64_t S, const float* d_theta, float* d_unitary, void* stream) {
    if (!plan) return fail(-1, "plan is NULL");
    if (S == 0) return 0;
    if (!d_theta || !d_unitary) return fail(-1, "NULL buffer");
    CPF_HIP(hipSetDevice(plan->device));
    cpf::KArgs a = plan->base;
    a.mode = cpf::MODE_UNITARY;
    a.init = d_theta; a.frozen = nullptr;
    a.out_U = reinterpret_cast<float2*>(d_unitary);
    return launch(plan, a, S, (hipStream_t)stream);
}

int32_t cpf_reg_device(const cpf_plan* plan, int64_t count, const float* d_theta, float* d_reg, void* stream) {
    if (!plan) return fail(-1, "plan is NULL");
    if (count == 0) return 0;
    if (!d_theta || !d_reg || count < 0) return fail(-1, "NULL buffer");
    CPF_HIP(hipSetDevice(plan->device));
    hipLaunchKernelGGL(cpf::k_reg_rows, dim3((unsigned)((count + 127) / 128)), dim3(128), 0, (hipStream_t)stream,
                       plan->base, d_theta, count, d_reg);
    CPF_HIP(hipGetLastError());
    return 0;
}

int32_t cpf_jit_compiles(void) { return cpf_jit::compile_counter().load(); }

int32_t cpf_jit_precompile(const cpf_problem* pb, int64_t num_samples, char* buf, size_t size) {
    if (int rc = validate(pb)) return rc;
    cpf_plan pl;                       // shape only: nothing here touches a device
    fill_plan_shape(pb, &pl);
    std::string report;
    int built = 0;
    const int loss_kind = pb->loss_kind == CPF_LOSS_RELATIVE_PHASE ? 2 : 0;
    if (dense_wanted(&pl, cpf::MODE_OPTIMIZE)) {          // the ZYZ-form / dense-block kernels serve this shape: their layer-specialised instances
        const int lw = dense_lw(&pl, num_samples);
        if (zyz_wanted(&pl, cpf::MODE_OPTIMIZE) && !cpf_zyz_has_aot(pl.NQ, lw, dense_rot_code(&pl), pl.layer_ud)) {
            const cpf_jit::Spec spec = zyz_jit_spec(pl.NQ, lw, dense_rot_code(&pl), pl.layer_ud);
            const cpf_jit::Code code = cpf_jit::build(spec);
            if (code.bytes.empty()) return fail(-31, "precompile of %s failed: %s", spec.body.c_str(), code.error.c_str());
            if (int rc = precompile_landed(code, spec.body)) return rc;
            report = "zyz/jit(" + code.backend + ") waves=" + std::to_string(1 << lw) + (code.compiled ? " [built] " : " [cached] ") + spec.body + "\n";
            built = 1;
            if (buf && size) snprintf(buf, size, "%s", report.c_str());
            return built;
        }
        if (dense_specialised(&pl) && !dense_has_aot(pl.NQ, lw, dense_rot_code(&pl), pl.layer_ud)) {
            const cpf_jit::Spec spec = dense_jit_spec(pl.NQ, lw, dense_rot_code(&pl), pl.layer_ud);
            const cpf_jit::Code code = cpf_jit::build(spec);
            if (code.bytes.empty()) return fail(-31, "precompile of %s failed: %s", spec.body.c_str(), code.error.c_str());
            if (int rc = precompile_landed(code, spec.body)) return rc;
            report = "dense/jit(" + code.backend + ") waves=" + std::to_string(1 << lw) + (code.compiled ? " [built] " : " [cached] ") + spec.body + "\n";
            built = 1;
        }
        if (buf && size) snprintf(buf, size, "%s", report.c_str());
        return built;
    }
    const int lw = choose_lw(&pl, num_samples);
    select_labelling(&pl, labelling_for(&pl, cpf::MODE_OPTIMIZE, lw));
    if (zyzl_wanted(&pl, cpf::MODE_OPTIMIZE, lw) && !(lw == 0 && cpf_zyzl_has_aot(pl.NQ, dense_rot_code(&pl), zyzl_slots(&pl, lw), pl.layer_ud))) {
        const cpf_jit::Spec spec = zyzl_jit_spec(pl.NQ, lw, dense_rot_code(&pl), zyzl_slots(&pl, lw), pl.layer_ud);
        const cpf_jit::Code code = cpf_jit::build(spec);
        if (code.bytes.empty()) return fail(-31, "precompile of %s failed: %s", spec.body.c_str(), code.error.c_str());
        if (int rc = precompile_landed(code, spec.body)) return rc;
        report += "zyzl/jit(" + code.backend + ") waves=" + std::to_string(1 << lw) + (code.compiled ? " [built] " : " [cached] ") + spec.body + "\n";
        built++;
    }
    // 3 qubits: the two-ended ZYZ-form kernel (cpf_zyz3.h), the default at every number of samples - then nothing else
    // is built for the shape: the gate-by-gate kernels that read the block count from the plan are part of the library
    if (zyz3_serves(&pl, cpf::MODE_OPTIMIZE, loss_kind) && zyz3_selected()) {
        // every launch regime's cut (zyz3_cut: K / 2 at one sample per CU, 0.25 K at two, 0.33 K while all workgroups are resident, 0.43 K in rounds),
        // whatever `num_samples` says: which regime a launch falls into depends on the device's CU count and LDS size, and a
        // cache built for one sample count must not make another one compile on the launch path.  CPF_ZYZ3_M, when
        // set, names the one cut to build.
        const int rot = dense_rot_code(&pl), pairs = zyz3_pairs(&pl, num_samples), K = pl.K;
        std::vector<int> cuts = {K / 2, (25 * K + 50) / 100, (33 * K + 50) / 100, (43 * K + 50) / 100};
        if (tuning("CPF_ZYZ3_M")) cuts = {zyz3_cut(&pl, num_samples)};
        std::set<int> seen;
        for (int m : cuts) {
            const int M = std::max(1, std::min(K - 1, m));
            if (!seen.insert(M).second || cpf_zyz3_has_aot(rot, K, M, pairs, pl.layer_ud)) continue;
            const cpf_jit::Spec spec = zyz3_jit_spec(rot, K, M, pairs, pl.layer_ud);
            const cpf_jit::Code code = cpf_jit::build(spec);
            if (code.bytes.empty()) return fail(-31, "precompile of %s failed: %s", spec.body.c_str(), code.error.c_str());
            if (int rc = precompile_landed(code, spec.body)) return rc;
            report += "zyz3/jit(" + code.backend + ") waves=2 K cut=" + std::to_string(M) + (pairs == 2 ? " pairs=2" : "") + (code.compiled ? " [built] " : " [cached] ") + spec.body + "\n";
            built++;
        }
        if (buf && size) snprintf(buf, size, "%s", report.c_str());
        return built;
    }
    for (const JitCandidate& c : jit_candidates(&pl, pl.P, loss_kind, lw, num_samples)) {
        const cpf_jit::Code code = cpf_jit::build(c.spec);
        if (code.bytes.empty()) return fail(-31, "precompile of %s failed: %s", c.spec.body.c_str(), code.error.c_str());
        if (int rc = precompile_landed(code, c.spec.body)) return rc;
        report += tag_with_backend(c.tag, code.backend) + (code.compiled ? " [built] " : " [cached] ") + c.spec.body + "\n";
        built++;
    }
    if (buf && size) snprintf(buf, size, "%s", report.c_str());
    return built;
}

int32_t cpf_run_static(const cpf_problem* problem, int32_t device, uint32_t seed, int64_t num_samples_total,
                       int64_t first, int64_t S, int32_t cp_zero, const float* init, const uint8_t* frozen,
                       cpf_result* out) {
    if (!out || !out->params || !out->regloss || !out->loss || !out->reg || !out->cz || !out->pass)
        return fail(-1, "result buffers must be allocated by the caller");
    if (S < 0) return fail(-10, "num_samples < 0");
    cpf_plan* plan = nullptr;
    if (int rc = cpf_plan_create(problem, device, &plan)) return rc;
    int rc = 0;
    {
        const size_t P = (size_t)plan->P, n = (size_t)S;
        DevBuf d_init, d_frozen, d_params, d_regloss, d_loss, d_reg, d_cz, d_pass;
        hipError_t e = hipSuccess;
        auto al = [&](DevBuf& b, size_t bytes) { if (e == hipSuccess) e = b.alloc(bytes); };
        al(d_init, n * P * 4); al(d_params, n * 2 * P * 4); al(d_regloss, n * 8); al(d_loss, n * 8); al(d_reg, n * 8);
        al(d_cz, n * 4); al(d_pass, n);
        if (frozen) al(d_frozen, n * P);
        if (e == hipSuccess && init) e = hipMemcpy(d_init.p, init, n * P * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess && frozen) e = hipMemcpy(d_frozen.p, frozen, n * P, hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = fail((int)e, "device allocation / upload failed: %s", hipGetErrorString(e));
        if (!rc && !init)
            rc = cpf_initial_angles_device(plan, seed, num_samples_total, first, S, cp_zero, d_init.as<float>(), nullptr);
        if (!rc)
            rc = cpf_optimize_device(plan, S, d_init.as<float>(), frozen ? d_frozen.as<uint8_t>() : nullptr,
                                     d_params.as<float>(), d_regloss.as<float>(), d_loss.as<float>(), d_reg.as<float>(),
                                     d_cz.as<int32_t>(), d_pass.as<uint8_t>(), nullptr, nullptr, nullptr);
        if (!rc) {
            e = hipDeviceSynchronize();
            auto dn = [&](void* dst, const DevBuf& b, size_t bytes) {
                if (e == hipSuccess && bytes) e = hipMemcpy(dst, b.p, bytes, hipMemcpyDeviceToHost);
            };
            dn(out->params, d_params, n * 2 * P * 4); dn(out->regloss, d_regloss, n * 8); dn(out->loss, d_loss, n * 8);
            dn(out->reg, d_reg, n * 8); dn(out->cz, d_cz, n * 4); dn(out->pass, d_pass, n);
            if (e != hipSuccess) rc = fail((int)e, "kernel / download failed: %s", hipGetErrorString(e));
        }
    }
    cpf_plan_destroy(plan);
    return rc;
}

}  // extern "C"
