// cpflow_amd: runtime specialisation of the layered kernel (cpf_layered.h) with hiprtc.
//
// A plan's circuit is fixed (qubits, rotation string, layer), so its block loop can be straight-line code.
// The ahead-of-time build carries that specialisation only for the reference's two 3-qubit layers; every
// other (n, waves per sample, rotation string, layer) is instantiated here on first use: the template
// arguments are spliced into a three-line translation unit, compiled for gfx950 (1-6 s; hipcc --genco as a
// child process, hiprtc as fallback), and the code object is cached in memory and on disk ($CPF_JIT_CACHE
// or ~/.cache/cpflow_amd).  If anything fails the caller
// falls back to the ahead-of-time generic kernel.  CPF_JIT=0 disables this path.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

namespace cpf_jit {

inline std::string lib_dir() {
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(&lib_dir), &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        const size_t k = p.find_last_of('/');
        return k == std::string::npos ? std::string(".") : p.substr(0, k);
    }
    return ".";
}

inline std::string read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

inline uint64_t fnv1a(const std::string& s, uint64_t h = 1469598103934665603ull) {
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    return h;
}

inline bool make_dirs(const std::string& d) {
    std::string acc;
    for (size_t i = 0; i <= d.size(); i++)
        if (i == d.size() || d[i] == '/') { if (!acc.empty()) mkdir(acc.c_str(), 0755); if (i < d.size()) acc += '/'; }
        else acc += d[i];
    struct stat st;
    return stat(d.c_str(), &st) == 0 && S_ISDIR(st.st_mode) && access(d.c_str(), W_OK) == 0;
}
inline std::string cache_dir() {
    const char* e = getenv("CPF_JIT_CACHE");
    const std::string first = e ? std::string(e) : std::string(getenv("HOME") ? getenv("HOME") : "/tmp") + "/.cache/cpflow_amd";
    if (make_dirs(first)) return first;
    const std::string fallback = "/tmp/cpflow_amd_cache_" + std::to_string((long)getuid());      // read-only home
    make_dirs(fallback);
    return fallback;
}

struct Kernel { hipModule_t module = nullptr; hipFunction_t fn = nullptr; };

// key -> loaded kernel, per process (and per device: the key carries the device index)
inline std::map<std::string, Kernel>& table() { static std::map<std::string, Kernel> t; return t; }
inline std::mutex& table_mutex() { static std::mutex m; return m; }

// Returns nullptr (and fills err) when the kernel cannot be produced.
// split: the two-ended kernel (cpf_split.h; nq == 3, lw ignored, 128 threads) instead of the layered one.
inline hipFunction_t get_layered(int device, int nq, int lw, int rot_code, int ppt, int mstore, const std::vector<int>& ud,
                                 std::string* err, bool split = false) {
    std::ostringstream name;
    name << (split ? "cpf::k_cp_split<" : "cpf::k_cp_layered<") << nq << ", " << lw << ", " << rot_code << ", " << ppt << ", " << mstore;
    for (int v : ud) name << ", " << v;
    name << ">";
    const std::string expr = name.str();
    const std::string key = "dev" + std::to_string(device) + ":" + expr;
    std::lock_guard<std::mutex> lock(table_mutex());
    auto it = table().find(key);
    if (it != table().end()) return it->second.fn;

    const std::string inc = lib_dir() + "/csrc";
    const std::string hdrs = read_file(inc + "/cpf_device.h") + read_file(inc + "/cpf_blocks.h") + read_file(inc + "/cpf_layered.h") +
                             (split ? read_file(inc + "/cpf_split.h") : std::string());
    if (hdrs.size() < 1000) { if (err) *err = "kernel headers not found under " + inc; return nullptr; }
    // one extern "C" kernel per translation unit: its name needs no lowering, whichever compiler builds it
    std::ostringstream targs;
    if (!split) targs << nq << ", " << lw << ", ";
    targs << rot_code << ", " << ppt << ", " << mstore;
    for (int v : ud) targs << ", " << v;
    const std::string src = std::string(split ? "#include \"cpf_split.h\"\n" : "#include \"cpf_layered.h\"\n") +
                            "extern \"C\" __global__ __launch_bounds__(" + std::to_string(split ? 128 : (64 << lw)) +
                            ", 1) void cpf_jit_kernel(const cpf::KArgs a) {\n    cpf::" +
                            (split ? "split_body<" : "layered_body<") + targs.str() + ">(a);\n}\n";
    // Backend: the offline compiler as a child process by default.  hiprtc builds the same source with the
    // same flags, but on the MI355X boxes its code came out 2x slower (SGPR spills through v_writelane /
    // v_readlane, 4477 vs 3536 instructions for the 3-qubit chain kernel: 20.6 ms vs 11.0 ms on config 2);
    // it stays as the fallback when no hipcc is installed.  CPF_JIT_BACKEND=hiprtc|hipcc overrides.
    const char* be = getenv("CPF_JIT_BACKEND");
    const char* hipcc_env = getenv("HIPCC");
    const std::string hipcc = hipcc_env ? hipcc_env : "/opt/rocm/bin/hipcc";
    struct stat st_;
    const bool have_hipcc = stat(hipcc.c_str(), &st_) == 0;
    const bool use_hipcc = be ? std::string(be) == "hipcc" : have_hipcc;
    char hname[64];
    snprintf(hname, sizeof(hname), "%016llx", (unsigned long long)fnv1a(src + (use_hipcc ? "hipcc" : "hiprtc"), fnv1a(hdrs)));
    const std::string cpath = cache_dir() + "/" + hname + ".hsaco";
    std::string code = read_file(cpath);
    const std::string lowered = "cpf_jit_kernel";
    // ranks of one job may build the same kernel at the same time: private temporaries, atomic rename
    const std::string tmp = cpath + "." + std::to_string((long)getpid()) + ".tmp";
    if (code.empty() && use_hipcc) {
        // offline compiler as a subprocess (same flags as the ahead-of-time build)
        const std::string spath = tmp + ".hip";
        std::ofstream(spath) << src;
        const std::string cmd = hipcc + " --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fno-signed-zeros "
                                "--genco -I" + inc + " -o " + tmp + " " + spath + " > " + tmp + ".log 2>&1";
        if (system(cmd.c_str()) == 0) {
            code = read_file(tmp);
            rename(tmp.c_str(), cpath.c_str());
            unlink((tmp + ".log").c_str());
        } else if (err) {
            *err = "hipcc --genco failed, see " + tmp + ".log; ";
        }
        unlink(spath.c_str());
    }
    if (code.empty()) {
        hiprtcProgram prog;
        if (hiprtcCreateProgram(&prog, src.c_str(), "cpf_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
            if (err) *err = "hiprtcCreateProgram failed";
            return nullptr;
        }
        const std::string i1 = "-I" + inc, i2 = "-I/opt/rocm/include";
        const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fno-signed-zeros",
                              i1.c_str(), i2.c_str()};
        const hiprtcResult r = hiprtcCompileProgram(prog, 7, opts);
        if (r != HIPRTC_SUCCESS) {
            size_t n = 0;
            hiprtcGetProgramLogSize(prog, &n);
            std::string log(n, '\0');
            if (n) hiprtcGetProgramLog(prog, &log[0]);
            if (err) *err = "hiprtc: " + log.substr(0, 400);
            hiprtcDestroyProgram(&prog);
            return nullptr;
        }
        size_t n = 0;
        hiprtcGetCodeSize(prog, &n);
        code.assign(n, '\0');
        hiprtcGetCode(prog, &code[0]);
        hiprtcDestroyProgram(&prog);
        { std::ofstream o(tmp, std::ios::binary); o << code; }
        rename(tmp.c_str(), cpath.c_str());
    }
    Kernel k;
    if (hipModuleLoadData(&k.module, code.data()) != hipSuccess ||
        hipModuleGetFunction(&k.fn, k.module, lowered.c_str()) != hipSuccess) {
        if (err) *err = "hipModuleLoadData / hipModuleGetFunction failed for " + expr;
        table()[key] = Kernel{};          // do not retry every launch
        return nullptr;
    }
    table()[key] = k;
    return k.fn;
}

}  // namespace cpf_jit
