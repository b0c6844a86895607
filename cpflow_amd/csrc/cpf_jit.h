// cpflow_amd: runtime specialisation of the layer-specialised kernels (cpf_layered.h, cpf_split.h, cpf_dense.h).
//
// A plan's circuit is fixed (qubits, rotation string, layer), so its block loop can be straight-line code.  The
// ahead-of-time build carries that specialisation for the BASELINE shapes; every other (n, waves per sample, rotation
// string, layer) is instantiated on first use: the template arguments are spliced into a three-line translation unit
// and compiled for gfx950.  Where the code object comes from, in this order:
//   1. <library dir>/jit_cache/   shipped with the library (python -m cpflow_amd.build --precompile fills it in the
//                                 build container for the reference's tutorial / paper shapes: no compiler at run time)
//   2. $CPF_JIT_CACHE or ~/.cache/cpflow_amd   (per user, mode 0700)
//   3. a compile: hipcc --genco as a child process (posix_spawn, no shell; profiler preloads stripped from its
//      environment), hiprtc in-process as the fallback and whenever a profiler is attached to this process.
// A cached hiprtc object is only accepted by a process that would itself build with hiprtc (or may not compile).
// CPF_JIT_FORBID_COMPILE=1 turns step 3 into an error (a test can prove that a box never compiles), CPF_JIT=0 disables
// the whole path.  Failures are remembered per kernel, so a launch loop does not re-run a failing compiler.
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

extern char** environ;

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

// a directory this user owns and nobody else can write (a planted .hsaco would be loaded with hipModuleLoadData)
inline bool private_dir(const std::string& d) {
    std::string acc;
    for (size_t i = 0; i <= d.size(); i++)
        if (i == d.size() || d[i] == '/') { if (!acc.empty()) mkdir(acc.c_str(), 0700); if (i < d.size()) acc += '/'; }
        else acc += d[i];
    struct stat st;
    return stat(d.c_str(), &st) == 0 && S_ISDIR(st.st_mode) && st.st_uid == getuid() && (st.st_mode & 022) == 0 &&
           access(d.c_str(), W_OK) == 0;
}
inline std::string shipped_dir() { return lib_dir() + "/jit_cache"; }
// note (optional): set when an explicit $CPF_JIT_CACHE was rejected and another directory is used instead
inline std::string cache_dir(std::string* note = nullptr) {
    const char* e = getenv("CPF_JIT_CACHE");
    const std::string first = e ? std::string(e) : std::string(getenv("HOME") ? getenv("HOME") : "/tmp") + "/.cache/cpflow_amd";
    if (private_dir(first)) return first;
    if (e && note) *note = " [CPF_JIT_CACHE=" + first + " rejected: it must be a directory owned by this user, writable, and not group/world-writable]";
    const std::string fallback = "/tmp/cpflow_amd_cache_" + std::to_string((long)getuid());      // read-only home
    if (private_dir(fallback)) return fallback;
    char tmpl[] = "/tmp/cpflow_amd_XXXXXX";                                                       // someone else owns that name
    const char* made = mkdtemp(tmpl);
    return made ? std::string(made) : std::string();
}

// a profiler attached to this process (rocprofv3 preloads its tool library, which initialises the GPU in every child
// that inherits the variables): compile in-process then
inline bool profiler_attached() {
    for (const char* v : {"ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "HSA_TOOLS_LIB", "ROCPROF_OUTPUT_PATH"})
        if (getenv(v)) return true;
    const char* pre = getenv("LD_PRELOAD");
    return pre && (strstr(pre, "rocprof") || strstr(pre, "roctracer"));
}

// hipcc as a child process: argv, no shell; stdout / stderr to `log`; profiler and preload variables removed
inline int run_hipcc(const std::vector<std::string>& argv, const std::string& log) {
    std::vector<char*> av;
    for (const std::string& a : argv) av.push_back(const_cast<char*>(a.c_str()));
    av.push_back(nullptr);
    std::vector<std::string> envs;
    for (char** e = environ; e && *e; e++) {
        const std::string kv(*e);
        if (kv.rfind("LD_PRELOAD=", 0) == 0 || kv.rfind("ROCP", 0) == 0 || kv.rfind("HSA_TOOLS", 0) == 0 || kv.rfind("ROCTRACER", 0) == 0)
            continue;
        envs.push_back(kv);
    }
    std::vector<char*> ev;
    for (const std::string& a : envs) ev.push_back(const_cast<char*>(a.c_str()));
    ev.push_back(nullptr);
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_addopen(&fa, 1, log.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
    posix_spawn_file_actions_adddup2(&fa, 1, 2);
    posix_spawn_file_actions_addopen(&fa, 0, "/dev/null", O_RDONLY, 0);
    pid_t pid = 0;
    const int rc = posix_spawn(&pid, av[0], &fa, nullptr, av.data(), ev.data());
    posix_spawn_file_actions_destroy(&fa);
    if (rc != 0) return -1;
    int status = 0;
    while (waitpid(pid, &status, 0) < 0) { if (errno != EINTR) return -1; }
    return WIFEXITED(status) ? WEXITSTATUS(status) : -1;
}

// what to build: `header` is the file under csrc/ that defines `body` (e.g. "cpf_layered.h",
// "cpf::layered_body<4, 0, 8451, 3, 0, 0, 1, 1, 2, 2, 3>"), `threads` the workgroup size
struct Spec {
    std::string header, body;
    int threads = 64;
    std::string arg = "cpf::KArgs";          // kernel argument type
    int min_waves = 1;                       // second __launch_bounds__ argument (waves per SIMD the register allocator must allow)
    std::vector<std::string> flags;          // extra compiler flags (e.g. -mllvm options)
};

struct Code { std::string bytes, backend, error, note, path; bool compiled = false; };    // note: why $CPF_JIT_CACHE was not used; path: where a compiled object was written    // backend: "hipcc" | "hiprtc"

inline std::atomic<int>& compile_counter() { static std::atomic<int> n{0}; return n; }

// Produce the code object of a spec without touching the GPU (cache look-ups, then a compile unless forbidden).
// cached_only: the look-ups alone; a miss is `error == "not cached"` and nothing is compiled or written.
inline Code build(const Spec& sp, bool cached_only = false) {
    Code out;
    const std::string inc = lib_dir() + "/csrc";
    std::string hdrs;
    for (const char* h : {"cpf_device.h", "cpf_blocks.h", "cpf_layered.h", "cpf_split.h", "cpf_dense.h", "cpf_dense_math.h", "cpf_dense_asm.h", "cpf_zyz.h", "cpf_zyz_math.h", "cpf_zyz_asm.h", "cpf_zyzl.h", "cpf_zyz3.h"}) hdrs += read_file(inc + "/" + h);
    if (hdrs.size() < 1000) { out.error = "kernel headers not found under " + inc; return out; }
    // one extern "C" kernel per translation unit: its name needs no lowering, whichever compiler builds it
    const std::string src = "#include \"" + sp.header + "\"\nextern \"C\" __global__ __launch_bounds__(" + std::to_string(sp.threads) +
                            ", " + std::to_string(sp.min_waves) + ") void cpf_jit_kernel(const " + sp.arg + " a) {\n    " + sp.body + "(a);\n}\n";
    // Backend: the offline compiler as a child process by default.  hiprtc builds the same source with the same flags,
    // but on the MI355X boxes its code came out 2x slower (SGPR spills through v_writelane / v_readlane, 4477 vs 3536
    // instructions for the 3-qubit chain kernel: 20.6 ms vs 11.0 ms on config 2); it is the fallback when no hipcc is
    // installed and the choice under a profiler.  CPF_JIT_BACKEND=hiprtc|hipcc overrides.
    const char* be = getenv("CPF_JIT_BACKEND");
    const char* hipcc_env = getenv("HIPCC");
    const std::string hipcc = hipcc_env ? hipcc_env : "/opt/rocm/bin/hipcc";
    struct stat st_;
    const bool have_hipcc = stat(hipcc.c_str(), &st_) == 0;
    auto name_for = [&](const char* backend) {
        char hname[64];
        std::string salt = src + backend;
        for (const std::string& f : sp.flags) salt += f;
        snprintf(hname, sizeof(hname), "%016llx", (unsigned long long)fnv1a(salt, fnv1a(hdrs)));
        return std::string(hname) + ".hsaco";
    };
    const char* forbid_env = getenv("CPF_JIT_FORBID_COMPILE");
    const bool forbid = forbid_env && forbid_env[0] == '1';
    const bool use_hipcc = be ? std::string(be) == "hipcc" : (have_hipcc && !profiler_attached());
    // An object built by hipcc is the preferred one wherever it is found.  A cached hiprtc object (2x slower code, above)
    // is accepted only where this process would not build with hipcc anyway - no hipcc installed, a profiler attached,
    // CPF_JIT_BACKEND=hiprtc, or compiling forbidden - so that one profiled run cannot pin the slow object for every
    // later run of the shape: those fall through to the hipcc compile.
    std::string dir_note;
    const std::string cdir = cache_dir(&dir_note);
    for (const char* backend : {"hipcc", "hiprtc"}) {
        if (be && std::string(be) != backend) continue;
        if (!strcmp(backend, "hiprtc") && use_hipcc && !forbid) continue;
        for (const std::string& dir : {shipped_dir(), cdir}) {
            if (dir.empty()) continue;
            out.bytes = read_file(dir + "/" + name_for(backend));
            if (!out.bytes.empty()) { out.backend = backend; return out; }
        }
    }
    if (cached_only) { out.error = "not cached"; return out; }
    if (forbid) { out.error = "run-time compile forbidden (CPF_JIT_FORBID_COMPILE=1) and no cached object for " + sp.body + dir_note; return out; }
    if (cdir.empty()) { out.error = "no writable private cache directory" + dir_note; return out; }
    out.note = dir_note;
    // ranks of one job may build the same kernel at the same time: private temporaries, atomic rename
    if (use_hipcc) {
        const std::string cpath = cdir + "/" + name_for("hipcc"), tmp = cpath + "." + std::to_string((long)getpid()) + ".tmp";
        const std::string spath = tmp + ".hip";
        { std::ofstream o(spath); o << src; }
        std::vector<std::string> argv = {hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fno-signed-zeros", "--genco"};
        for (const std::string& f : sp.flags) argv.push_back(f);
        for (const char* t : {"-I", "-o"}) (void)t;
        argv.push_back("-I" + inc); argv.push_back("-o"); argv.push_back(tmp); argv.push_back(spath);
        const int rc = run_hipcc(argv, tmp + ".log");
        unlink(spath.c_str());
        if (rc == 0) {
            out.bytes = read_file(tmp);
            rename(tmp.c_str(), cpath.c_str());
            unlink((tmp + ".log").c_str());
            if (!out.bytes.empty()) { out.backend = "hipcc"; out.compiled = true; out.path = cpath; compile_counter()++; return out; }
        }
        out.error = "hipcc --genco failed (exit " + std::to_string(rc) + "), see " + tmp + ".log; ";
        if (be) return out;
    }
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "cpf_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { out.error += "hiprtcCreateProgram failed"; return out; }
    const std::string i1 = "-I" + inc, i2 = "-I/opt/rocm/include";
    std::vector<const char*> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fno-signed-zeros", i1.c_str(), i2.c_str()};
    for (const std::string& f : sp.flags) opts.push_back(f.c_str());
    if (hiprtcCompileProgram(prog, (int)opts.size(), opts.data()) != HIPRTC_SUCCESS) {
        size_t n = 0;
        hiprtcGetProgramLogSize(prog, &n);
        std::string log(n, '\0');
        if (n) hiprtcGetProgramLog(prog, &log[0]);
        out.error += "hiprtc: " + log.substr(0, 400);
        hiprtcDestroyProgram(&prog);
        return out;
    }
    size_t n = 0;
    hiprtcGetCodeSize(prog, &n);
    out.bytes.assign(n, '\0');
    hiprtcGetCode(prog, &out.bytes[0]);
    hiprtcDestroyProgram(&prog);
    const std::string cpath = cdir + "/" + name_for("hiprtc"), tmp = cpath + "." + std::to_string((long)getpid()) + ".tmp";
    { std::ofstream o(tmp, std::ios::binary); o << out.bytes; }
    rename(tmp.c_str(), cpath.c_str());
    out.backend = "hiprtc"; out.compiled = true; out.path = cpath; out.error.clear(); compile_counter()++;
    return out;
}

struct Kernel { hipModule_t module = nullptr; hipFunction_t fn = nullptr; std::string backend, error; };

// key -> loaded kernel (or remembered failure), per process and device
inline std::map<std::string, Kernel>& table() { static std::map<std::string, Kernel> t; return t; }
inline std::mutex& table_mutex() { static std::mutex m; return m; }

// Returns the loaded kernel; fn == nullptr (with .error set) when it cannot be produced.  cached_only: never compile
// (a kernel that is an optional upgrade over one the library already holds); its misses are remembered apart from the
// failures of a full build, so that a later full build of the same body is still attempted.
inline const Kernel& get(int device, const Spec& sp, bool cached_only = false) {
    std::string key = "dev" + std::to_string(device) + ":" + sp.body;
    for (const std::string& f : sp.flags) key += " " + f;
    std::lock_guard<std::mutex> lock(table_mutex());
    auto it = table().find(key);
    if (it != table().end()) return it->second;
    if (cached_only && (it = table().find("miss:" + key)) != table().end()) return it->second;
    Kernel k;
    const Code code = build(sp, cached_only);
    if (cached_only && code.bytes.empty()) { k.error = code.error; return table()["miss:" + key] = k; }
    k.backend = code.backend;
    if (code.bytes.empty()) k.error = code.error;
    else if (hipModuleLoadData(&k.module, code.bytes.data()) != hipSuccess || hipModuleGetFunction(&k.fn, k.module, "cpf_jit_kernel") != hipSuccess) {
        k.fn = nullptr;
        k.error = "hipModuleLoadData / hipModuleGetFunction failed for " + sp.body;
    }
    return table()[key] = k;          // failures too: do not retry every launch
}

inline std::string body_expr(const char* fn, std::initializer_list<int> head, const std::vector<int>& ud) {
    std::ostringstream s;
    s << "cpf::" << fn << "<";
    bool first = true;
    for (int v : head) { s << (first ? "" : ", ") << v; first = false; }
    for (int v : ud) s << ", " << v;
    s << ">";
    return s.str();
}

}  // namespace cpf_jit
