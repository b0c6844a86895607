// CPU-only driver for the sanitizer build of the library's HOST code (tools/host_sanitize.sh): everything reachable without
// a GPU - argument validation, plan shape, kernel selection, the JIT cache (look-up, forbid switch, a real hipcc child
// process) - runs under AddressSanitizer + UndefinedBehaviorSanitizer.  Exit code 0 = clean.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/cpflow_amd.h"

static int fails = 0;
#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "CHECK failed: %s (line %d): %s\n", #cond, __LINE__, cpf_last_error()); fails++; } } while (0)

static cpf_problem problem(int n, const std::vector<int32_t>& place, const char* rot, const std::vector<double>& target) {
    cpf_problem pb;
    memset(&pb, 0, sizeof(pb));
    pb.n_qubits = n; pb.num_blocks = (int32_t)place.size() / 2; pb.placements = place.data();
    pb.n_rot = (int32_t)strlen(rot); pb.rot_axes = rot;
    pb.entangler = CPF_ENTANGLER_CP; pb.loss_kind = CPF_LOSS_HST; pb.target = target.data();
    pb.penalty.kind = CPF_PENALTY_LINEAR; pb.penalty.xmax = 1.5707963; pb.penalty.ymax = 2; pb.penalty.plato_0 = pb.penalty.plato_1 = pb.penalty.plato_2 = 0.05;
    pb.use_reg = 1; pb.r = 0.00055; pb.lr = 0.1; pb.b1 = 0.9; pb.b2 = 0.999; pb.eps = 1e-8; pb.num_iterations = 10; pb.threshold_cp = 0.2; pb.entry_loss = 1e-3;
    return pb;
}

int main(int argc, char** argv) {
    const char* cache = argc > 1 ? argv[1] : "/tmp/cpf_sanitize_cache";
    setenv("CPF_JIT_CACHE", cache, 1);
    CHECK(cpf_version() == CPF_VERSION);
    CHECK(cpf_plan_num_params(nullptr) < 0);
    char buf[4096];
    // validation errors come back before anything touches a device
    std::vector<double> t3(2 * 8 * 8, 0.0), t4(2 * 16 * 16, 0.0), t6(2 * 64 * 64, 0.0);
    std::vector<int32_t> bad = {0, 0, 1, 2};
    cpf_problem pb = problem(3, bad, "xyz", t3);
    cpf_plan* plan = nullptr;
    CHECK(cpf_plan_create(&pb, 0, &plan) < 0 && strstr(cpf_last_error(), "placement"));
    pb = problem(9, bad, "xyz", t3);
    CHECK(cpf_plan_create(&pb, 0, &plan) < 0 && strstr(cpf_last_error(), "n_qubits"));
    std::vector<int32_t> chain3; for (int b = 0; b < 9; b++) { chain3.push_back(b % 2 ? 2 : 1); chain3.push_back(b % 2 ? 1 : 0); }
    pb = problem(3, chain3, "xq", t3);
    CHECK(cpf_jit_precompile(&pb, 64, buf, sizeof(buf)) < 0 && strstr(cpf_last_error(), "rotation"));
    // kernel selection + JIT cache: forbidden compile, a real compile (hipcc child process), a cache hit
    pb = problem(3, chain3, "yz", t3);
    setenv("CPF_JIT_FORBID_COMPILE", "1", 1);
    CHECK(cpf_jit_precompile(&pb, 64, buf, sizeof(buf)) < 0 && strstr(cpf_last_error(), "CPF_JIT_FORBID_COMPILE"));
    unsetenv("CPF_JIT_FORBID_COMPILE");
    // (a 3-qubit shape: the two-ended ZYZ-form kernel, cpf_zyz3.h, and nothing else.  Since round 6 the precompile step builds
    // every launch regime's cut of a shape - here 9 blocks: cuts 4, 2 and 3 - unless CPF_ZYZ3_M names one; one cut keeps this driver quick)
    CHECK(cpf_set_tuning("CPF_ZYZ3_M", "4") == 0);
    int n = cpf_jit_precompile(&pb, 64, buf, sizeof(buf));
    CHECK(n == 1 && strstr(buf, "[built]") && strstr(buf, "zyz3/jit("));
    n = cpf_jit_precompile(&pb, 64, buf, sizeof(buf));
    CHECK(n == 1 && strstr(buf, "[cached]") && !strstr(buf, "[built]"));
    CHECK(cpf_jit_compiles() == 1);
    // a cached hiprtc object (2x slower code) must not pin itself: built under CPF_JIT_BACKEND=hiprtc (as a profiled run
    // would), it is reused by a process that asks for hiprtc again, but a normal process with hipcc installed builds
    // the hipcc object instead of accepting it; and an unknown tuning switch is rejected
    {
        std::vector<int32_t> chain3b; for (int b = 0; b < 7; b++) { chain3b.push_back(b % 2 ? 2 : 1); chain3b.push_back(b % 2 ? 1 : 0); }
        cpf_problem pr = problem(3, chain3b, "xy", t3);
        setenv("CPF_JIT_BACKEND", "hiprtc", 1);
        n = cpf_jit_precompile(&pr, 64, buf, sizeof(buf));
        CHECK(n >= 1 && strstr(buf, "(hiprtc)") && strstr(buf, "[built]"));
        n = cpf_jit_precompile(&pr, 64, buf, sizeof(buf));
        CHECK(n >= 1 && strstr(buf, "(hiprtc)") && strstr(buf, "[cached]") && !strstr(buf, "[built]"));
        unsetenv("CPF_JIT_BACKEND");
        n = cpf_jit_precompile(&pr, 64, buf, sizeof(buf));
        CHECK(n >= 1 && strstr(buf, "(hipcc)") && strstr(buf, "[built]") && !strstr(buf, "(hiprtc)"));
        setenv("CPF_JIT_FORBID_COMPILE", "1", 1);               // may not compile: whatever the cache holds is taken, hipcc first
        n = cpf_jit_precompile(&pr, 64, buf, sizeof(buf));
        CHECK(n >= 1 && strstr(buf, "(hipcc)") && strstr(buf, "[cached]"));
        unsetenv("CPF_JIT_FORBID_COMPILE");
        CHECK(cpf_set_tuning("CPF_NOT_A_SWITCH", "1") < 0 && cpf_set_tuning("CPF_SPLIT", "0") == 0 && cpf_set_tuning("CPF_SPLIT", nullptr) == 0);
    }
    CHECK(cpf_set_tuning("CPF_ZYZ3_M", nullptr) == 0);
    n = cpf_jit_precompile(&pb, 64, buf, sizeof(buf));           // every cut now: 4 from the cache, 2 and 3 built
    CHECK(n == 3 && strstr(buf, "cut=4 [cached]") && strstr(buf, "cut=2 [built]") && strstr(buf, "cut=3 [built]"));
    const int compiles_so_far = cpf_jit_compiles();
    CHECK(compiles_so_far >= 5);
    // shapes served by ahead-of-time kernels need nothing: config 2 (3 qubits, chain) -> the JIT candidates exist but are never
    // reached at run time; config 5 (6 qubits on a ring) -> the dense-block instance is built ahead of time
    std::vector<int32_t> ring6; for (int b = 0; b < 60; b++) { ring6.push_back(b % 6); ring6.push_back((b % 6 + 1) % 6); }
    pb = problem(6, ring6, "xyz", t6);
    CHECK(cpf_jit_precompile(&pb, 1024, buf, sizeof(buf)) == 0);
    // a 4-qubit star (tutorial shape): one layered kernel, 4 waves per sample below 1024 samples, 1 wave above
    std::vector<int32_t> star4; for (int b = 0; b < 23; b++) { star4.push_back(0); star4.push_back(1 + b % 3); }
    pb = problem(4, star4, "xyz", t4);
    setenv("CPF_JIT_FORBID_COMPILE", "1", 1);
    CHECK(cpf_jit_precompile(&pb, 500, buf, sizeof(buf)) < 0);
    printf("host sanitize driver: %d failure(s)\n", fails);
    return fails ? 1 : 0;
}
