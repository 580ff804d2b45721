// rtc.cpp -- the structure-specialised chain kernel (device/chain_hot.hpp) for ANY chain: `dls_chain_hot_kernel<NJ, code...>` is
// compiled for the problem's own placement-structure code at problem-creation time through hipRTC, so a robot that is not one of
// the fixture models (kernels_hot.hip pre-builds Cassie's legs and the UR arms) gets the same lane program instead of the general
// build.  The reference's loader takes any URDF (pinocchio::urdf::buildModelFromXML, ik_ros/src/cassie.cpp:34-35); this is what
// keeps the headline kernel from being a property of two URDFs.
//
//  * The four device headers the kernel needs are embedded in the library verbatim (.incbin below) and handed to hipRTC as named
//    headers: what is compiled at run time is exactly the source the pre-built instantiations were compiled from, with the flags of
//    kernels_hot.hip (-fno-signed-zeros -fno-honor-nans -fno-honor-infinities -ffp-contract=on; no fast-math).
//  * libhiprtc is opened with dlopen: when it is absent, or IKGPU_RTC=0, or the compilation fails, the problem runs on the general
//    chain build -- never on a CPU path.
//  * Code objects are cached in memory (per structure code) and on disk ($IKGPU_CACHE_DIR, else $XDG_CACHE_HOME/ikgpu, else
//    ~/.cache/ikgpu; keyed by a hash of headers + source + the flags actually passed + the hipRTC version; each file carries a header
//    -- magic, key, length, checksum -- that is validated on read; the directory must be the caller's own, mode 0700), modules are
//    loaded per device on first launch.
//  * THE COMPILER NEVER RUNS IN THE CALLER'S PROCESS.  LLVM can abort() on a program it cannot lower (round 3: "LLVM ERROR: Cannot
//    scavenge register in FI elimination" took a test process down), and a library whose *_create can kill the robot's control
//    process is a boundary defect (include/ikgpu.h: "nothing here ever throws").  A cache miss therefore spawns the worker
//    `ikgpu_precompile --request <file>` (tools/ikgpu_precompile.cpp, installed next to libikgpu.so; posix_spawn, no fork handlers,
//    no exec of this process), which runs hipRTC and leaves the code object in the cache; a worker that exits non-zero, dies on a
//    signal or outlives IKGPU_RTC_TIMEOUT_S (default 600) means "this problem runs on its general build / interpreter", nothing
//    else.  The worker initialises no device.  IKGPU_RTC_INPROCESS=1 restores in-process compilation (debugging).
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <signal.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "device/chain_hot.hpp"
#include "device/generic_solver.hpp"
#include "device/pik_solver.hpp"
#include "kernels.hpp"

// The device headers, verbatim.  (Paths are relative to ik_amd/csrc, where the Makefile runs the compiler.)
#define IKGPU_EMBED(sym, path)                                                                                   \
    __asm__(".section .rodata\n.global " #sym "\n.type " #sym ", @object\n" #sym ":\n.incbin \"" path "\"\n.byte 0\n" \
            ".global " #sym "_end\n" #sym "_end:\n.previous\n");                                                    \
    extern "C" const char sym[];                                                                                 \
    extern "C" const char sym##_end[];
IKGPU_EMBED(ikgpu_src_lane_math, "device/lane_math.hpp")
IKGPU_EMBED(ikgpu_src_chain_solver, "device/chain_solver.hpp")
IKGPU_EMBED(ikgpu_src_chain_kernel_body, "device/chain_kernel_body.hpp")
IKGPU_EMBED(ikgpu_src_chain_hot, "device/chain_hot.hpp")
IKGPU_EMBED(ikgpu_src_tree_solver, "device/tree_solver.hpp")
IKGPU_EMBED(ikgpu_src_generic_solver, "device/generic_solver.hpp")
IKGPU_EMBED(ikgpu_src_pik_solver, "device/pik_solver.hpp")
IKGPU_EMBED(ikgpu_src_primal_solver, "device/primal_solver.hpp")

extern char **environ;   // (the compile worker inherits the caller's environment: IKGPU_CACHE_DIR, IKGPU_RTC_DEFINES, ...)

namespace ikgpu {
namespace {

struct RtcApi {
    void *lib = nullptr;
    decltype(&hiprtcCreateProgram) create = nullptr;
    decltype(&hiprtcCompileProgram) compile = nullptr;
    decltype(&hiprtcGetCodeSize) code_size = nullptr;
    decltype(&hiprtcGetCode) code = nullptr;
    decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
    decltype(&hiprtcGetProgramLog) log = nullptr;
    decltype(&hiprtcDestroyProgram) destroy = nullptr;
    decltype(&hiprtcVersion) version = nullptr;
    bool ok = false;
};

const RtcApi &rtc_api() {
    static const RtcApi api = [] {
        RtcApi a;
        for (const char *name : {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (a.lib) break;
        }
        if (!a.lib) return a;
        auto sym = [&](const char *n) { return dlsym(a.lib, n); };
        a.create = reinterpret_cast<decltype(a.create)>(sym("hiprtcCreateProgram"));
        a.compile = reinterpret_cast<decltype(a.compile)>(sym("hiprtcCompileProgram"));
        a.code_size = reinterpret_cast<decltype(a.code_size)>(sym("hiprtcGetCodeSize"));
        a.code = reinterpret_cast<decltype(a.code)>(sym("hiprtcGetCode"));
        a.log_size = reinterpret_cast<decltype(a.log_size)>(sym("hiprtcGetProgramLogSize"));
        a.log = reinterpret_cast<decltype(a.log)>(sym("hiprtcGetProgramLog"));
        a.destroy = reinterpret_cast<decltype(a.destroy)>(sym("hiprtcDestroyProgram"));
        a.version = reinterpret_cast<decltype(a.version)>(sym("hiprtcVersion"));
        a.ok = a.create && a.compile && a.code_size && a.code && a.log_size && a.log && a.destroy;
        return a;
    }();
    return api;
}

bool rtc_enabled() {
    const char *env = std::getenv("IKGPU_RTC");
    return !(env && env[0] == '0');
}

uint64_t fnv1a(uint64_t h, const void *data, size_t n) {
    const unsigned char *p = static_cast<const unsigned char *>(data);
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

const char *const kFlags[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-fast-math", "-ffp-contract=on",
                              "-fno-signed-zeros", "-fno-honor-nans", "-fno-honor-infinities"};

// values: non-structural placement entries of the chain.  The refill kernel asks for two waves per SIMD (256 registers) unless the
// parked table alone would take most of them.
std::string hot_source(int nj, const uint64_t code[3], int values) {
    char buf[2048];
    const char *refill_bounds = values <= 40 ? "__launch_bounds__(64, 2)" : "__launch_bounds__(64)";
    std::snprintf(buf, sizeof buf,
                  "#include \"chain_hot.hpp\"\n"
                  "typedef ikdev::ChainStruct<0x%llxull, 0x%llxull, 0x%llxull> S;\n"
                  "extern \"C\" __global__ __launch_bounds__(64) void ikgpu_hot_never(const ikdev::ChainKernelArgs<%d> a, const ikdev::HotTable t) {\n"
                  "    ikdev::hot_kernel_entry<%d, S, true>(a, t);\n}\n"
                  "extern \"C\" __global__ __launch_bounds__(64) void ikgpu_hot_stop(const ikdev::ChainKernelArgs<%d> a, const ikdev::HotTable t) {\n"
                  "    ikdev::hot_kernel_entry<%d, S, false>(a, t);\n}\n"
                  "extern \"C\" __global__ %s void ikgpu_hot_refill(const ikdev::ChainKernelArgs<%d> a, const ikdev::HotTable t, unsigned long long *queue, int chunk) {\n"
                  "    ikdev::hot_refill_entry<%d, S>(a, t, queue, chunk);\n}\n",
                  static_cast<unsigned long long>(code[0]), static_cast<unsigned long long>(code[1]), static_cast<unsigned long long>(code[2]),
                  nj, nj, nj, nj, refill_bounds, nj, nj);
    return buf;
}

// The on-disk cache directory, or "" when there is none this process may trust: code objects read from it are handed to
// hipModuleLoadData, so it has to be a directory (not a symlink) owned by the caller that nobody else can write to.
std::string cache_dir() {
    std::string d;
    if (const char *e = std::getenv("IKGPU_CACHE_DIR")) d = e;
    else if (const char *x = std::getenv("XDG_CACHE_HOME")) d = std::string(x) + "/ikgpu";
    else if (const char *h = std::getenv("HOME")) d = std::string(h) + "/.cache/ikgpu";
    else d = "/tmp/ikgpu-cache-" + std::to_string(static_cast<long>(getuid()));
    // mkdir -p (two levels are enough for the defaults)
    const size_t slash = d.rfind('/');
    if (slash != std::string::npos && slash > 0) (void)mkdir(d.substr(0, slash).c_str(), 0755);
    (void)mkdir(d.c_str(), 0700);
    struct stat st;
    if (lstat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != getuid() || (st.st_mode & (S_IWGRP | S_IWOTH)) != 0) return std::string();
    return d;
}

bool read_file(const std::string &path, std::vector<char> &out) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? static_cast<size_t>(n) : 0);
    const bool ok = n > 0 && std::fread(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    return ok;
}

// A cached code object: {magic, key, payload bytes, FNV-1a of the payload} + the payload.  A file that is truncated, belongs to
// another key or was edited is treated as a miss.
struct CacheHeader {
    char magic[8];
    uint64_t key, bytes, check;
};
constexpr char kCacheMagic[8] = {'I', 'K', 'G', 'P', 'U', 'C', 'O', '2'};

bool read_cached_object(const std::string &path, uint64_t key, std::vector<char> &code) {
    std::vector<char> raw;
    if (!read_file(path, raw) || raw.size() <= sizeof(CacheHeader)) return false;
    CacheHeader h;
    std::memcpy(&h, raw.data(), sizeof h);
    if (std::memcmp(h.magic, kCacheMagic, sizeof kCacheMagic) != 0 || h.key != key || h.bytes != raw.size() - sizeof h) return false;
    if (fnv1a(14695981039346656037ull, raw.data() + sizeof h, h.bytes) != h.check) return false;
    code.assign(raw.begin() + sizeof h, raw.end());
    return true;
}

bool write_file_atomically(const std::string &path, const std::vector<char> &data) {
    const std::string tmp = path + ".tmp." + std::to_string(static_cast<long>(getpid()));
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(data.data(), 1, data.size(), f) == data.size();
    std::fclose(f);
    if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) { (void)std::remove(tmp.c_str()); return false; }
    return true;
}

bool write_cached_object(const std::string &path, uint64_t key, const std::vector<char> &code) {
    CacheHeader h;
    std::memcpy(h.magic, kCacheMagic, sizeof kCacheMagic);
    h.key = key;
    h.bytes = code.size();
    h.check = fnv1a(14695981039346656037ull, code.data(), code.size());
    std::vector<char> raw(sizeof h + code.size());
    std::memcpy(raw.data(), &h, sizeof h);
    std::memcpy(raw.data() + sizeof h, code.data(), code.size());
    return write_file_atomically(path, raw);
}

typedef std::tuple<int, uint64_t, uint64_t, uint64_t> ShapeKey;

struct HotCode {
    bool tried = false, ok = false;
    std::vector<char> code;
    std::string log;
};
struct HotModule {
    hipModule_t mod = nullptr;
    hipFunction_t never = nullptr, stop = nullptr, refill = nullptr;
    int refill_waves_per_cu = 0;
};

// g_mu guards the maps below (held for map access and module loading only, never while a compiler runs); g_compile_mu serialises
// compilations (a second thread asking for the same program waits for the first instead of spawning a second compiler).
std::mutex g_mu, g_compile_mu;
std::map<ShapeKey, HotCode> g_codes;
std::map<std::pair<ShapeKey, int>, HotModule> g_modules;
std::string g_last_log;
std::atomic<bool> g_in_worker{false};   // this process IS the compile worker (ikgpu_rtc_worker_compile): compile here

ShapeKey key_of(const ProblemHost &ph) {
    return ShapeKey(ph.chain.nj, ph.chain_struct.code[0], ph.chain_struct.code[1], ph.chain_struct.code[2]);
}

struct Hdr { const char *name, *begin, *end; };
const Hdr kHeaders[] = {{"lane_math.hpp", ikgpu_src_lane_math, ikgpu_src_lane_math_end},
                        {"chain_solver.hpp", ikgpu_src_chain_solver, ikgpu_src_chain_solver_end},
                        {"chain_kernel_body.hpp", ikgpu_src_chain_kernel_body, ikgpu_src_chain_kernel_body_end},
                        {"chain_hot.hpp", ikgpu_src_chain_hot, ikgpu_src_chain_hot_end},
                        {"tree_solver.hpp", ikgpu_src_tree_solver, ikgpu_src_tree_solver_end},
                        {"generic_solver.hpp", ikgpu_src_generic_solver, ikgpu_src_generic_solver_end},
                        {"pik_solver.hpp", ikgpu_src_pik_solver, ikgpu_src_pik_solver_end},
                        {"primal_solver.hpp", ikgpu_src_primal_solver, ikgpu_src_primal_solver_end}};
constexpr int kNumHeaders = static_cast<int>(sizeof kHeaders / sizeof kHeaders[0]);

// The flag vector a program is compiled with.  if_convert (the generated lane programs): every two-armed choice whose arms the compiler
// may evaluate speculatively becomes a select, whatever the arms cost -- no divergent branch inside the iteration (see IKD_CHOOSE in
// device/lane_math.hpp); and `#pragma unroll` must not give up (16384 unrolled instructions by default): a loop of a static lane
// program left rolled indexes the workspace with a run-time value, which demotes the WHOLE workspace from registers to scratch
// memory (18.8 KB per lane for an M = 29 problem, 20 s of compile time, and "LLVM ERROR: Cannot scavenge register in FI
// elimination" for its sibling without the constraint).  With the threshold lifted: 1.8 KB, 8 s, no error.
// (debugging aid: IKGPU_RTC_PLAIN_FLAGS, bit k set: drop the k-th of the no-signed-zeros / no-NaN / no-infinity flags.  The -mllvm
// flags stay, and the vector that results is what the cache key hashes.)
std::vector<const char *> compile_flags(bool if_convert) {
    std::vector<const char *> flags(kFlags, kFlags + 5);
    long mask = 0;
    if (const char *m = std::getenv("IKGPU_RTC_PLAIN_FLAGS")) mask = std::strtol(m, nullptr, 10);
    for (int k = 0; k < 3; ++k)
        if (!((mask >> k) & 1)) flags.push_back(kFlags[5 + k]);
    if (if_convert) {
        flags.push_back("-mllvm"); flags.push_back("-two-entry-phi-node-folding-threshold=100000");
        flags.push_back("-mllvm"); flags.push_back("-pragma-unroll-threshold=4000000");
    }
    return flags;
}

// Identity of a program: the embedded headers, the generated source, the hipRTC version.  (What the in-memory maps are keyed by.)
uint64_t source_hash(const std::string &src) {
    const RtcApi &api = rtc_api();
    uint64_t h = 14695981039346656037ull;
    for (const Hdr &x : kHeaders) h = fnv1a(h, x.begin, static_cast<size_t>(x.end - x.begin));
    h = fnv1a(h, src.data(), src.size());
    for (const char *f : kFlags) h = fnv1a(h, f, std::strlen(f));
    int vmaj = 0, vmin = 0;
    if (api.version) (void)api.version(&vmaj, &vmin);
    h = fnv1a(h, &vmaj, sizeof vmaj);
    return fnv1a(h, &vmin, sizeof vmin);
}

// Identity of a code object on disk: the program + the flag vector it was actually compiled with.
uint64_t object_key(const std::string &src, const std::vector<const char *> &flags) {
    uint64_t h = source_hash(src);
    for (const char *f : flags) { h = fnv1a(h, f, std::strlen(f)); h = fnv1a(h, "\0", 1); }
    return h;
}

// IKGPU_RTC_DEFINES="-DX -DY=3" (debugging aid) is prepended to the source as #define lines and so takes part in every key.
std::string with_defines(const std::string &src) {
    const char *defs = std::getenv("IKGPU_RTC_DEFINES");
    if (!defs) return src;
    std::string d(defs), pre;
    size_t pos = 0;
    while ((pos = d.find("-D", pos)) != std::string::npos) {
        size_t end = d.find(' ', pos);
        if (end == std::string::npos) end = d.size();
        std::string item = d.substr(pos + 2, end - pos - 2);
        const size_t eq = item.find('=');
        pre += "#define " + (eq == std::string::npos ? item + " 1" : item.substr(0, eq) + " " + item.substr(eq + 1)) + "\n";
        pos = end;
    }
    return pre + src;
}

// hipRTC, in THIS process: only the compile worker (and IKGPU_RTC_INPROCESS=1) gets here.
void compile_here(const std::string &src, const std::vector<const char *> &flags, HotCode &hc) {
    const RtcApi &api = rtc_api();
    hiprtcProgram prog = nullptr;
    const char *hsrc[kNumHeaders], *hname[kNumHeaders];
    for (int i = 0; i < kNumHeaders; ++i) { hsrc[i] = kHeaders[i].begin; hname[i] = kHeaders[i].name; }
    if (api.create(&prog, src.c_str(), "ikgpu_rtc.hip", kNumHeaders, hsrc, hname) != HIPRTC_SUCCESS) { hc.log = "hiprtcCreateProgram failed"; return; }
    std::vector<const char *> f(flags);
    const hiprtcResult rc = api.compile(prog, static_cast<int>(f.size()), f.data());
    size_t ls = 0;
    if (api.log_size(prog, &ls) == HIPRTC_SUCCESS && ls > 1) {
        hc.log.resize(ls);
        (void)api.log(prog, &hc.log[0]);
    }
    size_t cs = 0;
    if (rc == HIPRTC_SUCCESS && api.code_size(prog, &cs) == HIPRTC_SUCCESS && cs > 0) {
        hc.code.resize(cs);
        hc.ok = api.code(prog, hc.code.data()) == HIPRTC_SUCCESS;
    }
    (void)api.destroy(&prog);
}

// ---- the compile worker --------------------------------------------------------------------------------------------------------------
std::string worker_path() {
    if (const char *e = std::getenv("IKGPU_PRECOMPILE_EXE")) return e;
    Dl_info info;
    if (dladdr(reinterpret_cast<const void *>(&rtc_api), &info) == 0 || !info.dli_fname) return std::string();
    std::string dir(info.dli_fname);
    const size_t slash = dir.rfind('/');
    dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
    return dir + "/ikgpu_precompile";
}

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return static_cast<double>(ts.tv_sec) + 1e-9 * static_cast<double>(ts.tv_nsec);
}

std::string file_tail(const std::string &path, size_t max_bytes) {
    std::vector<char> raw;
    if (!read_file(path, raw)) return std::string();
    const size_t from = raw.size() > max_bytes ? raw.size() - max_bytes : 0;
    return std::string(raw.begin() + static_cast<long>(from), raw.end());
}

// Hands (prefix, source, if_convert) to a fresh `ikgpu_precompile --request <file>` and waits for it.  Returns true when the worker
// exited 0 (the code object is then in the cache); otherwise `why` says what happened -- exit status, signal, timeout -- with the tail of
// the worker's stderr.  The child is a NEW program started with posix_spawn: nothing of this process (its GPU context, its threads,
// its locks) exists in it, and this process is never replaced.
bool run_worker(const std::string &dir, const char *prefix, const std::string &src, bool if_convert, std::string &why) {
    const std::string exe = worker_path();
    if (exe.empty() || access(exe.c_str(), X_OK) != 0) { why = "compile worker not found (" + exe + "; IKGPU_PRECOMPILE_EXE overrides)"; return false; }
    static std::atomic<unsigned> counter{0};
    const std::string req = dir + "/request." + std::to_string(static_cast<long>(getpid())) + "." + std::to_string(counter.fetch_add(1)) + ".req";
    const std::string log = req + ".log";
    {
        std::string head = std::string("IKGPU-RTC-REQUEST 1\n") + prefix + "\n" + (if_convert ? "1" : "0") + "\n" + std::to_string(src.size()) + "\n";
        std::vector<char> body(head.begin(), head.end());
        body.insert(body.end(), src.begin(), src.end());
        if (!write_file_atomically(req, body)) { why = "cannot write the compile request " + req; return false; }
    }
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_addopen(&fa, 0, "/dev/null", O_RDONLY, 0);
    posix_spawn_file_actions_addopen(&fa, 1, log.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
    posix_spawn_file_actions_adddup2(&fa, 1, 2);
    std::string a0 = exe, a1 = "--request", a2 = req;
    char *argv[] = {&a0[0], &a1[0], &a2[0], nullptr};
    pid_t pid = -1;
    const int rc = posix_spawn(&pid, exe.c_str(), &fa, nullptr, argv, environ);
    posix_spawn_file_actions_destroy(&fa);
    if (rc != 0) { why = std::string("posix_spawn(") + exe + "): " + std::strerror(rc); (void)std::remove(req.c_str()); return false; }
    double limit = 600.0;
    if (const char *e = std::getenv("IKGPU_RTC_TIMEOUT_S")) { const double v = std::strtod(e, nullptr); if (v > 0.0) limit = v; }
    const double t0 = now_s();
    int status = 0;
    bool reaped = false, timed_out = false;
    for (;;) {
        const pid_t r = waitpid(pid, &status, WNOHANG);
        if (r == pid) { reaped = true; break; }
        if (r < 0 && errno != EINTR) break;   // ECHILD: the host ignores SIGCHLD and the kernel reaped the child -- the cache decides
        if (now_s() - t0 > limit) {
            timed_out = true;
            (void)kill(pid, SIGKILL);          // exactly the process started above
            (void)waitpid(pid, &status, 0);
            break;
        }
        const timespec nap{0, 5 * 1000 * 1000};
        (void)nanosleep(&nap, nullptr);
    }
    bool ok = false;
    char buf[160];
    if (timed_out) std::snprintf(buf, sizeof buf, "compile worker killed after %.0f s (IKGPU_RTC_TIMEOUT_S)", limit);
    else if (!reaped) { ok = true; std::snprintf(buf, sizeof buf, "compile worker's exit status unavailable (SIGCHLD ignored by the host)"); }
    else if (WIFEXITED(status) && WEXITSTATUS(status) == 0) { ok = true; buf[0] = '\0'; }
    else if (WIFEXITED(status)) std::snprintf(buf, sizeof buf, "compile worker exited with status %d", WEXITSTATUS(status));
    else if (WIFSIGNALED(status)) std::snprintf(buf, sizeof buf, "compile worker died on signal %d (%s)", WTERMSIG(status), strsignal(WTERMSIG(status)));
    else std::snprintf(buf, sizeof buf, "compile worker ended abnormally (status 0x%x)", status);
    why = buf;
    const std::string tail = file_tail(log, 4000);
    if (!tail.empty()) why += (why.empty() ? "" : ":\n") + tail;
    (void)std::remove(req.c_str());
    (void)std::remove(log.c_str());
    return ok;
}

// Fetches the code object of `src_in` from the on-disk cache or has it compiled (file name: prefix + object_key).
// IKGPU_RTC_DUMP=<dir>: also writes the generated source there.  Called WITHOUT g_mu.
void compile_cached(const char *prefix, const std::string &src_in, HotCode &hc, bool if_convert = false) {
    hc.tried = true;
    const std::string src = with_defines(src_in);
    const RtcApi &api = rtc_api();
    if (!api.ok) { hc.log = "libhiprtc could not be loaded"; return; }
    const std::vector<const char *> flags = compile_flags(if_convert);
    const uint64_t key = object_key(src, flags);
    char name[96];
    std::snprintf(name, sizeof name, "/%s_%016llx", prefix, static_cast<unsigned long long>(key));
    if (const char *dump = std::getenv("IKGPU_RTC_DUMP")) {
        std::vector<char> text(src.begin(), src.end());
        (void)write_file_atomically(std::string(dump) + name + ".hip", text);
    }
    const std::string dir = cache_dir();
    const std::string path = dir.empty() ? std::string() : dir + name + ".hsaco";
    if (!path.empty() && read_cached_object(path, key, hc.code)) { hc.ok = true; hc.log = "cached: " + path; return; }
    const char *inproc = std::getenv("IKGPU_RTC_INPROCESS");
    if (g_in_worker.load() || (inproc && inproc[0] == '1')) {
        compile_here(src, flags, hc);
        if (hc.ok && !path.empty()) (void)write_cached_object(path, key, hc.code);
    } else if (path.empty()) {
        hc.log = "no cache directory this process may trust (owned by the caller, not group / world writable): the compile worker has nowhere to leave the code object";
    } else {
        std::string why;
        const bool ran = run_worker(dir, prefix, src_in, if_convert, why);
        hc.ok = ran && read_cached_object(path, key, hc.code);
        hc.log = hc.ok ? "compiled by the worker: " + path : (why.empty() ? "compile worker left no code object" : why);
    }
    if (!hc.ok && std::getenv("IKGPU_RTC_VERBOSE")) std::fprintf(stderr, "ikgpu: run-time compilation failed:\n%s\n", hc.log.c_str());
}

// The code object of (map, key): compiled at most once per process, without g_mu held while the compiler runs.
template <class Map, class Key>
HotCode &ensure_code(Map &map, const Key &key, const char *prefix, const std::string &src, bool if_convert) {
    std::lock_guard<std::mutex> one_compiler(g_compile_mu);
    HotCode *hc = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        hc = &map[key];   // (std::map: the node never moves)
        if (hc->tried) { g_last_log = hc->log; return *hc; }
    }
    HotCode fresh;
    compile_cached(prefix, src, fresh, if_convert);
    std::lock_guard<std::mutex> lock(g_mu);
    *hc = std::move(fresh);
    g_last_log = hc->log;
    return *hc;
}

// Compiles (or fetches) the code object of this shape.  Called WITHOUT g_mu.
HotCode &code_for(const ProblemHost &ph) {
    return ensure_code(g_codes, key_of(ph), "chain_hot", hot_source(ph.chain.nj, ph.chain_struct.code, ph.chain_struct.values), false);
}

// The loaded module of this shape on the current device (a copy: the map is guarded).  Called WITHOUT g_mu.
bool module_for(const ProblemHost &ph, HotModule &out, hipError_t *err) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const auto mkey = std::make_pair(key_of(ph), dev);
    {
        std::lock_guard<std::mutex> lock(g_mu);
        const auto it = g_modules.find(mkey);
        if (it != g_modules.end() && it->second.mod) { out = it->second; return true; }
    }
    HotCode &hc = code_for(ph);
    std::lock_guard<std::mutex> lock(g_mu);
    if (!hc.ok) { *err = hipErrorInvalidImage; return false; }
    HotModule &m = g_modules[mkey];
    if (!m.mod) {
        hipError_t e = hipModuleLoadData(&m.mod, hc.code.data());
        if (e == hipSuccess) e = hipModuleGetFunction(&m.never, m.mod, "ikgpu_hot_never");
        if (e == hipSuccess) e = hipModuleGetFunction(&m.stop, m.mod, "ikgpu_hot_stop");
        if (e == hipSuccess) e = hipModuleGetFunction(&m.refill, m.mod, "ikgpu_hot_refill");
        if (e == hipSuccess) {
            int per_cu = 0;
            if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, m.refill, 64, 0) != hipSuccess || per_cu < 1) per_cu = 4;
            m.refill_waves_per_cu = per_cu;
        }
        if (e != hipSuccess) { *err = e; m = HotModule{}; return false; }
    }
    out = m;
    return true;
}

template <int NJ>
hipError_t launch_shape(const ProblemHost &ph, const DeviceTables &dt, const BatchIO &io, const ikgpu_dls_params &prm, hipStream_t stream,
                        const HotModule &m) {
    struct Args {   // the kernel-argument segment: (ChainKernelArgs<NJ> a, HotTable t, unsigned long long *queue), natural alignment
        ikdev::ChainKernelArgs<NJ> a;
        ikdev::HotTable t;
        unsigned long long *queue;
        int chunk;
    } args{};
    static_assert(sizeof(ikdev::ChainKernelArgs<NJ>) % 8 == 0 && sizeof(ikdev::HotTable) % 8 == 0, "argument layout");
    ikdev::ChainKernelArgs<NJ> &a = args.a;
    fill_chain_args(ph, a.ref_pl, a.qidx, a.vidx, &a.nq, &a.nv, &a.prm.priority, &a.prm.idmask, &a.prm.unit_weights);
    a.lower = dt.lower; a.upper = dt.upper; a.q_in_chain = dt.q_in_chain;
    a.prm.max_iterations = prm.max_iterations;
    a.prm.lam2 = prm.damping * prm.damping;
    a.prm.step_length = prm.step_length;
    a.prm.stop_sq_tol = prm.stop_sq_tol;
    a.layout = io.layout; a.B = io.B; a.q0 = io.q0; a.targets = io.targets;
    a.q_out = io.q_out; a.success = io.success; a.iters = io.iters;
    if (ph.chain_hot.size() > static_cast<size_t>(ikdev::kHotTableMax)) return hipErrorInvalidValue;
    std::memcpy(args.t.v, ph.chain_hot.data(), ph.chain_hot.size() * sizeof(double));

    const int64_t waves = (io.B + 63) / 64;
    auto launch = [&](hipFunction_t fn, int64_t grid, size_t nbytes) {
        void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &nbytes, HIP_LAUNCH_PARAM_END};
        return hipModuleLaunchKernel(fn, static_cast<unsigned>(grid), 1, 1, 64, 1, 1, 0, stream, nullptr, config);
    };
    const size_t two = offsetof(Args, queue), three = offsetof(Args, chunk) + sizeof(int);
    if (prm.stop_sq_tol < 0.0) return launch(m.never, waves, two);
    int cus = 256;
    {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
    }
    const int64_t resident = refill_resident(static_cast<int64_t>(m.refill_waves_per_cu) * cus, io.B);
    const int mode = stop_rule_mode(prm, io.B, resident, stream, false);
    if (mode == kStopLockStep) return launch(m.stop, waves, two);
    if (mode == kStopTwoPhase) {
        hipError_t le = hipSuccess;
        const hipError_t te = run_two_phase(dt.queues, io, stream, a, false, [&] { le = launch(m.stop, waves, two); },
                                            [&](unsigned long long *queue) {
                                                args.queue = queue;
                                                args.chunk = refill_chunk(io.B, resident);
                                                const hipError_t r = launch(m.refill, resident, three);
                                                if (le == hipSuccess) le = r;
                                            });
        return te != hipSuccess ? te : le;
    }
    hipError_t e = hipSuccess;
    args.queue = dt.queues.slot_for(stream, &e);
    if (!args.queue) return e;
    void *tmp = nullptr;
    if (!a.iters) {
        if ((e = hipMallocAsync(&tmp, sizeof(int32_t) * static_cast<size_t>(io.B), stream)) != hipSuccess) return e;
        a.iters = static_cast<int32_t *>(tmp);
    }
    args.chunk = refill_chunk(io.B, resident);
    e = launch(m.refill, resident, three);
    if (e == hipSuccess) e = launch_chain_pass_through(ph, dt, io, a.iters, stream);
    if (tmp) {
        const hipError_t f = hipFreeAsync(tmp, stream);
        if (e == hipSuccess) e = f;
    }
    return e;
}

// ---- the generic lane program specialised for ONE problem (device/generic_solver.hpp with IKD_STATIC_TABLES) -----------------------
// The model tree and the task table become `static constexpr` members of a generated type; the workspace is a local array.  With
// every loop unrolled the compiler sees the whole iteration as straight-line code over constants: FK of joints no task reads is
// dead, the dense M x nv Jacobian keeps only its structurally non-zero entries, the workspace lives in registers.

std::string hexd(double v) {
    char b[64];
    if (v != v) return "__builtin_nan(\"\")";
    if (v > 1.7976931348623157e308) return "__builtin_inf()";
    if (v < -1.7976931348623157e308) return "(-__builtin_inf())";
    std::snprintf(b, sizeof b, "%a", v);
    return b;
}

template <class It>
std::string int_array(const char *name, It begin, int n) {
    std::string o = std::string("    static constexpr int ") + name + "[] = {";
    for (int i = 0; i < n; ++i) o += std::to_string(static_cast<long long>(begin[i])) + (i + 1 < n ? ", " : "");
    if (n == 0) o += "0";
    return o + "};\n";
}

template <class It>
std::string dbl_array(const char *name, It begin, int n) {
    std::string o = std::string("    static constexpr double ") + name + "[] = {";
    for (int i = 0; i < n; ++i) o += hexd(begin[i]) + (i + 1 < n ? ", " : "");
    if (n == 0) o += "0.0";
    return o + "};\n";
}

#ifndef IKGPU_STATIC_MAX_ROWS
#define IKGPU_STATIC_MAX_ROWS 24
#endif
// Problems whose PostureTask rows the static program eliminates from the linear system (generic_solver.hpp, TB::elim): those beyond
// what the dense unrolled solve stays compilable at.  Fewer rows keep the dense solve -- the same operations in the same order as the
// interpreter forms, bit for bit.
int static_posture_rows(const ProblemHost &ph) {
    int n = 0;
    for (const ikgpu_task &t : ph.tasks) n += t.type == IKGPU_POSTURE_ROW ? 1 : 0;
    return n;
}
// Rows of the linear system the unrolled program is taken up to.  IKGPU_STATIC_MAX_ROWS can LOWER the built-in cap (tests, A/B);
// raising it needs IKGPU_UNSAFE=1 next to it (experiments: beyond the cap nothing has been through the compiler -- the compile worker
// contains a compiler crash, but minutes of compile time and KBs of scratch per lane are what was measured there).
int static_max_rows() {
    if (const char *env = std::getenv("IKGPU_STATIC_MAX_ROWS")) {
        const long v = std::strtol(env, nullptr, 10);
        const char *unsafe = std::getenv("IKGPU_UNSAFE");
        const long cap = (unsafe && unsafe[0] == '1') ? 64 : IKGPU_STATIC_MAX_ROWS;
        if (v >= 1) return static_cast<int>(std::min(v, cap));
    }
    return IKGPU_STATIC_MAX_ROWS;
}
bool static_elimination(const ProblemHost &ph) {
    const int np = static_posture_rows(ph);
    return np > 0 && (np > 8 || ph.rows > static_max_rows());
}

// ---- the primal, tree-sparse form of the static DLS program (device/primal_solver.hpp) ---------------------------------------------
// Taken when the dense dual program would outgrow the register file (more than 12 solved rows) and the problem has what the form
// covers: frame / alignment / posture rows on revolute, prismatic and free-flyer joints, no centre-of-mass task (rows dense over every
// tangent direction), no constraint.  IKGPU_STATIC_FORM=primal takes it for every such problem, =dual never (tests, A/B).
struct PrimalPlan {
    bool ok = false;
    int max_path = 1, max_pdofs = 1, nnz = 0, lds_words = 0;
    int max_anc = 1;
    std::vector<int> t_npath, t_path, t_nrpath, t_rpath, t_anchor, hidx, park_off, anc_n, anc, shared, grp, t_kc, park_at;
};

constexpr int kPrimalLdsWords = 80;   // 40 KB per one-wave workgroup: four of them fill a CU's 160 KB, one wave per SIMD

PrimalPlan primal_plan(const ProblemHost &ph) {
    PrimalPlan pl;
    const GenericHost &g = ph.generic;
    const int32_t *I = g.ints.data();
    const int nj = g.njoints, nt = ph.ntasks, nv = ph.nv;
    if (g.has_com || !ph.constraints.empty() || nv > 36) return pl;
    auto ndof = [&](int j) { return I[g.o_jtype + j] == ikdev::GJ_FREEFLYER ? 6 : I[g.o_jtype + j] == ikdev::GJ_UNIVERSE ? 0 : 1; };
    auto path_of = [&](int joint) {   // root -> joint, without the universe
        std::vector<int> p;
        for (int j = joint; j > 0; j = I[g.o_parent + j]) p.insert(p.begin(), j);
        return p;
    };
    std::vector<std::vector<int>> paths(static_cast<size_t>(nt)), rpaths(static_cast<size_t>(nt));
    std::vector<char> coupled(static_cast<size_t>(nv) * nv, 0), act(static_cast<size_t>(nv), 0);
    pl.t_anchor.assign(static_cast<size_t>(nt), -1);
    for (int t = 0; t < nt; ++t) {
        const int type = I[g.o_ttype + t];
        if (type == IKGPU_CENTRE_OF_MASS) return pl;
        if (type == IKGPU_POSTURE_ROW) {
            const int c = I[g.o_tfjoint + t];   // (the row's tangent column)
            if (c < 0 || c >= nv) return pl;
            act[static_cast<size_t>(c)] = 1;
            pl.t_anchor[static_cast<size_t>(t)] = c;
            continue;
        }
        paths[static_cast<size_t>(t)] = path_of(I[g.o_tfjoint + t]);
        rpaths[static_cast<size_t>(t)] = path_of(I[g.o_trjoint + t]);
        std::vector<int> dofs;
        for (int j : paths[static_cast<size_t>(t)]) {
            // (the elimination order relies on the model's depth-first numbering: a joint's tangent directions follow its parent's)
            if (I[g.o_parent + j] > 0 && I[g.o_idx_v + j] < I[g.o_idx_v + I[g.o_parent + j]]) return pl;
            for (int u = 0; u < ndof(j); ++u) dofs.push_back(I[g.o_idx_v + j] + u);
        }
        if (dofs.empty()) return pl;   // a frame welded to the universe: nothing to solve for (the dual program handles the row)
        for (int a : dofs) {
            act[static_cast<size_t>(a)] = 1;
            for (int b : dofs) coupled[static_cast<size_t>(a) * nv + b] = 1;
        }
        pl.t_anchor[static_cast<size_t>(t)] = *std::max_element(dofs.begin(), dofs.end());
        pl.max_path = std::max<int>(pl.max_path, static_cast<int>(std::max(paths[static_cast<size_t>(t)].size(), rpaths[static_cast<size_t>(t)].size())));
        pl.max_pdofs = std::max<int>(pl.max_pdofs, static_cast<int>(dofs.size()));
    }
    // the pattern of H and of its factor (no fill: the coupled earlier directions of c are ancestors of c on one task's path, hence
    // coupled among themselves) -- verified here rather than assumed
    for (int c = nv - 1; c >= 0; --c)
        for (int d = 0; d < c; ++d)
            for (int e = 0; e < d; ++e)
                if (coupled[static_cast<size_t>(c) * nv + d] && coupled[static_cast<size_t>(c) * nv + e] && !coupled[static_cast<size_t>(d) * nv + e]) return pl;
    pl.hidx.assign(static_cast<size_t>(nv) * nv, -1);
    for (int c = 0; c < nv; ++c)
        for (int d = 0; d <= c; ++d)
            if ((c == d && act[static_cast<size_t>(c)]) || (c != d && coupled[static_cast<size_t>(c) * nv + d])) pl.hidx[static_cast<size_t>(c) * nv + d] = pl.nnz++;
    if (pl.nnz > 420) return pl;
    pl.anc_n.assign(static_cast<size_t>(nv), 0);
    for (int c = 0; c < nv; ++c)
        for (int d = 0; d < c; ++d) pl.anc_n[static_cast<size_t>(c)] += pl.hidx[static_cast<size_t>(c) * nv + d] >= 0 ? 1 : 0;
    pl.max_anc = std::max(1, *std::max_element(pl.anc_n.begin(), pl.anc_n.end()));
    pl.anc.assign(static_cast<size_t>(nv) * pl.max_anc, 0);
    for (int c = 0; c < nv; ++c) {
        int k = 0;
        for (int d = 0; d < c; ++d)   // ascending
            if (pl.hidx[static_cast<size_t>(c) * nv + d] >= 0) pl.anc[static_cast<size_t>(c) * pl.max_anc + k++] = d;
    }
    // shared / private directions (primal_solver.hpp): a direction moved by tasks of two or more anchors keeps its row in H (the blocks
    // are accumulated into it); one that belongs to a single anchor's tasks forms its row from their blocks when its turn comes
    pl.shared.assign(static_cast<size_t>(nv), 0);
    pl.grp.assign(static_cast<size_t>(nv), -1);
    pl.t_kc.assign(static_cast<size_t>(nt) * nv, -1);
    {
        std::vector<std::vector<int>> anchors(static_cast<size_t>(nv));
        for (int t = 0; t < nt; ++t) {
            if (I[g.o_ttype + t] == IKGPU_POSTURE_ROW) continue;
            int kc = 0;
            for (int j : paths[static_cast<size_t>(t)])
                for (int u = 0; u < ndof(j); ++u) {
                    const int c = I[g.o_idx_v + j] + u;
                    pl.t_kc[static_cast<size_t>(t) * nv + c] = kc++;
                    std::vector<int> &a = anchors[static_cast<size_t>(c)];
                    if (std::find(a.begin(), a.end(), pl.t_anchor[static_cast<size_t>(t)]) == a.end()) a.push_back(pl.t_anchor[static_cast<size_t>(t)]);
                }
        }
        for (int c = 0; c < nv; ++c) {
            pl.shared[static_cast<size_t>(c)] = anchors[static_cast<size_t>(c)].size() > 1 ? 1 : 0;
            pl.grp[static_cast<size_t>(c)] = anchors[static_cast<size_t>(c)].size() == 1 ? anchors[static_cast<size_t>(c)][0] : (anchors[static_cast<size_t>(c)].empty() ? c : -1);
        }
        // (an ancestor of a shared direction is shared: both anchors' paths run through it)
        for (int c = 0; c < nv; ++c)
            for (int d = 0; d < c; ++d)
                if (pl.shared[static_cast<size_t>(c)] && coupled[static_cast<size_t>(c) * nv + d] && !pl.shared[static_cast<size_t>(d)]) return pl;
    }
    // finished columns parked in LDS until the back substitution: in elimination order (deepest first), the root joint's excepted
    // (they are read back first), while the slab has room
    int budget = kPrimalLdsWords;
    if (const char *env = std::getenv("IKGPU_PRIMAL_LDS_WORDS")) budget = std::max(0, std::min(kPrimalLdsWords, std::atoi(env)));
    pl.lds_words = ph.nq;   // the slab's first words hold q (primal_solver.hpp WsPrimal)
    if (pl.lds_words > budget) return pl;
    pl.park_off.assign(static_cast<size_t>(nv), -1);
    std::vector<int> dof_joint(static_cast<size_t>(nv), 0);
    for (int j = 1; j < nj; ++j)
        for (int u = 0; u < ndof(j); ++u) dof_joint[static_cast<size_t>(I[g.o_idx_v + j] + u)] = j;
    for (int c = nv - 1; c >= 0; --c) {
        if (!act[static_cast<size_t>(c)] || I[g.o_parent + dof_joint[static_cast<size_t>(c)]] == 0) continue;
        int words = 2;
        for (int d = 0; d < c; ++d) words += pl.hidx[static_cast<size_t>(c) * nv + d] >= 0 ? 1 : 0;
        if (pl.lds_words + words > budget) continue;
        pl.park_off[static_cast<size_t>(c)] = pl.lds_words;
        pl.lds_words += words;
    }
    // a shared column is parked when it is finished; a private one when the LAST (lowest) private direction of its anchor is, since
    // the rows of that anchor's private directions read the deeper columns
    pl.park_at.assign(static_cast<size_t>(nv), -1);
    for (int c = 0; c < nv; ++c) {
        if (pl.park_off[static_cast<size_t>(c)] < 0) continue;
        int at = c;
        if (!pl.shared[static_cast<size_t>(c)])
            for (int d = 0; d < c; ++d)
                if (act[static_cast<size_t>(d)] && !pl.shared[static_cast<size_t>(d)] && pl.grp[static_cast<size_t>(d)] == pl.grp[static_cast<size_t>(c)]) { at = d; break; }
        pl.park_at[static_cast<size_t>(c)] = at;
    }
    pl.t_npath.assign(static_cast<size_t>(nt), 0);
    pl.t_nrpath.assign(static_cast<size_t>(nt), 0);
    pl.t_path.assign(static_cast<size_t>(nt) * pl.max_path, 0);
    pl.t_rpath.assign(static_cast<size_t>(nt) * pl.max_path, 0);
    for (int t = 0; t < nt; ++t) {
        pl.t_npath[static_cast<size_t>(t)] = static_cast<int>(paths[static_cast<size_t>(t)].size());
        pl.t_nrpath[static_cast<size_t>(t)] = static_cast<int>(rpaths[static_cast<size_t>(t)].size());
        for (size_t k = 0; k < paths[static_cast<size_t>(t)].size(); ++k) pl.t_path[static_cast<size_t>(t) * pl.max_path + k] = paths[static_cast<size_t>(t)][k];
        for (size_t k = 0; k < rpaths[static_cast<size_t>(t)].size(); ++k) pl.t_rpath[static_cast<size_t>(t) * pl.max_path + k] = rpaths[static_cast<size_t>(t)][k];
    }
    pl.ok = true;
    return pl;
}

bool static_form_is_primal(const ProblemHost &ph) {
    const char *form = std::getenv("IKGPU_STATIC_FORM");
    if (form && std::strcmp(form, "dual") == 0) return false;
    const bool forced = form && std::strcmp(form, "primal") == 0;
    const int np = static_posture_rows(ph);
    const int solve_rows = (np > 0 && (np > 8 || ph.rows > IKGPU_STATIC_MAX_ROWS)) ? ph.rows - np : ph.rows;
    if (!forced && solve_rows <= 12) return false;
    return primal_plan(ph).ok;
}

enum StaticKind { kStaticDls = 0, kStaticDlsRefill = 1, kStaticPik = 2, kStaticPikDa = 3 };

std::string generic_static_source(const ProblemHost &ph, int kind = kStaticDls) {
    const bool refill = kind == kStaticDlsRefill;
    const GenericHost &g = ph.generic;
    const int nj = g.njoints, nt = ph.ntasks;
    const int32_t *I = g.ints.data();
    const double *D = g.dbls.data();
    const bool primal = (kind == kStaticDls || kind == kStaticDlsRefill) && static_form_is_primal(ph);
    std::string o = "#define IKD_STATIC_TABLES 1\n#include \"chain_kernel_body.hpp\"\n#include \"generic_solver.hpp\"\n";
    if (kind == kStaticPik || kind == kStaticPikDa) o += "#include \"pik_solver.hpp\"\n";
    if (primal) o += "#include \"primal_solver.hpp\"\n";
    o += "namespace {\nstruct T {\n";
    auto scalar = [&](const char *n, long long v) { o += std::string("    static constexpr int ") + n + " = " + std::to_string(v) + ";\n"; };
    scalar("njoints", nj); scalar("nq", ph.nq); scalar("nv", ph.nv); scalar("ntasks", nt); scalar("M", ph.rows);
    o += int_array("jtype", I + g.o_jtype, nj) + int_array("parent", I + g.o_parent, nj) + int_array("idx_q", I + g.o_idx_q, nj) + int_array("idx_v", I + g.o_idx_v, nj);
    o += dbl_array("placement", D + g.o_placement, 12 * nj) + dbl_array("axis", D + g.o_axis, 3 * nj);
    o += dbl_array("lower", D + g.o_lower, ph.nq) + dbl_array("upper", D + g.o_upper, ph.nq);
    o += int_array("t_type", I + g.o_ttype, nt) + int_array("t_fjoint", I + g.o_tfjoint, nt) + int_array("t_rjoint", I + g.o_trjoint, nt);
    o += int_array("t_row", I + g.o_trow, nt) + int_array("t_dim", I + g.o_tdim, nt) + int_array("t_prio", I + g.o_tprio, nt);
    o += dbl_array("t_fpl", D + g.o_tfpl, 12 * nt) + dbl_array("t_rpl", D + g.o_trpl, 12 * nt) + dbl_array("t_w", D + g.o_tw, 6 * nt);
    scalar("off_q", g.off_q); scalar("off_oMi", g.off_oMi); scalar("off_Jw", g.off_Jw); scalar("off_e", g.off_e); scalar("off_J", g.off_J);
    scalar("off_G", g.off_G); scalar("off_y", g.off_y); scalar("off_dq", g.off_dq); scalar("ws_words", g.ws_words);
    scalar("nlevels", g.nlevels);
    o += int_array("lvl_row0", I + g.o_lvlrow0, g.nlevels + 1);
    {   // the rows of the stacked system by kind: PostureTask rows (one non-zero each) are eliminated from the solve when there are many
        std::vector<int> f_row, p_row, p_col;
        for (int t = 0; t < nt; ++t) {
            const int row = I[g.o_trow + t], dim = I[g.o_tdim + t];
            if (I[g.o_ttype + t] == IKGPU_POSTURE_ROW) { p_row.push_back(row); p_col.push_back(I[g.o_tfjoint + t]); }
            else for (int r = 0; r < dim; ++r) f_row.push_back(row + r);
        }
        std::sort(f_row.begin(), f_row.end());
        scalar("elim", static_elimination(ph) ? 1 : 0); scalar("Mf", static_cast<long long>(f_row.size())); scalar("Mp", static_cast<long long>(p_row.size()));
        o += int_array("f_row", f_row.begin(), static_cast<int>(f_row.size())) + int_array("p_row", p_row.begin(), static_cast<int>(p_row.size())) +
             int_array("p_col", p_col.begin(), static_cast<int>(p_col.size()));
    }
    const int nc = static_cast<int>(ph.constraints.size());
    scalar("ncons", nc); scalar("Mc", ph.crows); scalar("off_Jc", g.off_Jc);
    o += int_array("c_type", I + g.o_ctype, nc) + int_array("c_fjoint", I + g.o_cfjoint, nc) + int_array("c_rjoint", I + g.o_crjoint, nc);
    o += int_array("c_row", I + g.o_crow, nc) + int_array("c_dim", I + g.o_cdim, nc);
    o += dbl_array("c_fpl", D + g.o_cfpl, 12 * nc) + dbl_array("c_rpl", D + g.o_crpl, 12 * nc);
    scalar("has_com", g.has_com); scalar("off_sf", g.off_sf);
    if (g.has_com) o += dbl_array("j_mass", D + g.o_jmass, nj) + dbl_array("j_lever", D + g.o_jlever, 3 * nj) + dbl_array("j_submass", D + g.o_jsubmass, nj);
    else o += "    static constexpr double j_mass[] = {0.0}, j_lever[] = {0.0}, j_submass[] = {0.0};\n";
    o += "    static constexpr double inv_total_mass = " + hexd(g.inv_total_mass) + ";\n";
    {   // ik::pik (pik_solver.hpp static_pik): the widest level, the last non-empty one, the rows of the levels before it
        int max_rows = 1, last = 0;
        for (int l = 0; l < g.nlevels; ++l) {
            const int ml = I[g.o_lvlrow0 + l + 1] - I[g.o_lvlrow0 + l];
            max_rows = std::max(max_rows, ml);
            if (ml > 0) last = l;
        }
        scalar("pik_max_rows", max_rows); scalar("pik_last_level", last); scalar("pik_basis_rows", I[g.o_lvlrow0 + last]);
    }
    if (!primal) scalar("primal", 0);
    if (primal) {
        const PrimalPlan pl = primal_plan(ph);
        scalar("primal", 1); scalar("max_path", pl.max_path); scalar("max_pdofs", pl.max_pdofs); scalar("nnz", pl.nnz); scalar("lds_words", std::max(1, pl.lds_words));
        o += int_array("t_npath", pl.t_npath.begin(), nt) + int_array("t_path", pl.t_path.begin(), nt * pl.max_path);
        o += int_array("t_nrpath", pl.t_nrpath.begin(), nt) + int_array("t_rpath", pl.t_rpath.begin(), nt * pl.max_path);
        o += int_array("t_anchor", pl.t_anchor.begin(), nt) + int_array("hidx", pl.hidx.begin(), ph.nv * ph.nv) + int_array("park_off", pl.park_off.begin(), ph.nv);
        {
            std::vector<int> anch_n(static_cast<size_t>(ph.nv), 0), anch_t(static_cast<size_t>(ph.nv) * nt, 0);
            for (int t = 0; t < nt; ++t) {
                const int c = pl.t_anchor[static_cast<size_t>(t)];
                if (c >= 0) anch_t[static_cast<size_t>(c) * nt + anch_n[static_cast<size_t>(c)]++] = t;
            }
            o += int_array("anch_n", anch_n.begin(), ph.nv) + int_array("anch_t", anch_t.begin(), ph.nv * nt);
        }
        o += int_array("shared", pl.shared.begin(), ph.nv) + int_array("grp", pl.grp.begin(), ph.nv) + int_array("t_kc", pl.t_kc.begin(), nt * ph.nv) +
             int_array("park_at", pl.park_at.begin(), ph.nv);
        scalar("max_anc", pl.max_anc);
        o += int_array("anc_n", pl.anc_n.begin(), ph.nv) + int_array("anc", pl.anc.begin(), ph.nv * pl.max_anc);
    }
    o += "};\n}  // namespace\n";
    if (primal) {   // (lock-step only: a stop-rule batch larger than the machine runs this kernel too)
        o += "extern \"C\" __global__ __launch_bounds__(64) void ikgpu_lane_dls(const ikdev::GenericKernelArgs a) {\n"
             "    __shared__ double slab[T::lds_words][64];\n"
             "    double w[T::ws_words];\n"
             "    ikdev::dls_primal_body_ws(a, T{}, static_cast<int64_t>(blockIdx.x) * 64 + threadIdx.x, ikdev::WsReg{w}, ikdev::LdsColumn{&slab[0][threadIdx.x], 64},\n"
             "                              [](bool act) { return __any(act) != 0; }, static_cast<int64_t>(blockIdx.x) * 64);\n"
             "}\n";
        return o;
    }
    if (kind == kStaticPik || kind == kStaticPikDa)
        o += std::string("extern \"C\" __global__ __launch_bounds__(64) void ikgpu_lane_pik(const ikdev::PikKernelArgs a) {\n"
                         "    double w[T::ws_words];\n"
                         "    ikdev::pik_static_body<") + (kind == kStaticPikDa ? "true" : "false") +
             ">(a, T{}, static_cast<int64_t>(blockIdx.x) * 64 + threadIdx.x, ikdev::WsReg{w}, [](bool act) { return __any(act) != 0; });\n"
             "}\n";
    else if (refill)   // the stop-rule mode on batches larger than the machine (generic_solver.hpp GenericRefill): its own module, compiled when first needed
        o += "extern \"C\" __global__ __launch_bounds__(64) void ikgpu_lane_dls_refill(const ikdev::GenericKernelArgs a, unsigned long long *queue, int chunk) {\n"
             "    double w[T::ws_words];\n"
             "    ikdev::dls_generic_refill_body(a, T{}, static_cast<int64_t>(blockIdx.x), static_cast<int64_t>(gridDim.x), ikdev::WsReg{w}, queue, chunk);\n"
             "}\n";
    else
        o += "extern \"C\" __global__ __launch_bounds__(64) void ikgpu_lane_dls(const ikdev::GenericKernelArgs a) {\n"
             "    double w[T::ws_words];\n"
             "    ikdev::dls_generic_body_ws(a, T{}, static_cast<int64_t>(blockIdx.x) * 64 + threadIdx.x, ikdev::WsReg{w}, ikdev::KeepGoing{a.leave_active, a.leave_after, 0},\n"
             "                               T::M > 12 ? static_cast<int64_t>(blockIdx.x) * 64 : int64_t{-1});\n"
             "}\n";
    return o;
}

struct GenModule {
    hipModule_t mod = nullptr;
    hipFunction_t dls = nullptr;
    // the refill program: compiled and loaded at the first launch that wants it (or by ikgpu_problem_precompile)
    bool refill_tried = false;
    hipModule_t refill_mod = nullptr;
    hipFunction_t refill = nullptr;
    int refill_waves_per_cu = 0;
};
std::map<uint64_t, HotCode> g_gen_codes;                      // by hash of the generated source
std::map<std::pair<uint64_t, int>, GenModule> g_gen_modules;   // (hash, device)

}  // namespace

int rtc_static_solve_rows(const ProblemHost &ph) { return static_elimination(ph) ? ph.rows - static_posture_rows(ph) : ph.rows; }

bool rtc_chain_hot_available(const ProblemHost &ph, bool compile) {
    if (!rtc_enabled() || ph.kind != KernelKind::Chain || !ph.chain_struct.fits || ph.chain.nj < 1 || ph.chain.nj > 7) return false;
    if (!rtc_api().ok) return false;
    if (!compile) return true;
    return code_for(ph).ok;
}

std::string rtc_last_log() {
    std::lock_guard<std::mutex> lock(g_mu);
    return g_last_log;
}

hipError_t rtc_launch_chain_hot(const ProblemHost &ph, const DeviceTables &dt, const BatchIO &io, const ikgpu_dls_params &prm, hipStream_t stream) {
    HotModule m;
    {
        hipError_t e = hipSuccess;
        if (!module_for(ph, m, &e)) return e;
    }
    switch (ph.chain.nj) {
#define X(N) case N: return launch_shape<N>(ph, dt, io, prm, stream, m);
        X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ikgpu

namespace ikgpu {

// What the static program takes: sizes the unrolled program stays compilable at (constraints project by Gram-Schmidt there).  IKGPU_GENERIC_KERNEL (lane / lds / coop: tests, A/B) keeps the other forms.
bool rtc_generic_static_available(const ProblemHost &gen, bool compile, uint64_t *key_out) {
    if (!rtc_enabled() || gen.kind != KernelKind::Generic || gen.crows > 12) return false;
    if (std::getenv("IKGPU_GENERIC_KERNEL")) return false;
    if (const char *env = std::getenv("IKGPU_GENERIC_STATIC")) { if (env[0] == '0') return false; }
    // (measured compile times of the unrolled program: M = 10: 2 s, 15: 9 s, 21: ~20 s, 31: 48 s with 19 KB of scratch per lane;
    // 28 rows of which 16 are posture rows: 440 s -- such problems keep the cooperative kernel)
    // (with the PostureTask rows eliminated from the solve the caps apply to the rows that are left: the reference demo with every line
    // switched on -- M = 29, 16 of them posture rows -- is a 13 x 13 system)
    const int solve_rows = rtc_static_solve_rows(gen);
    // (the primal tree-sparse form, primal_solver.hpp, keeps no dense matrix: it is not bound by the dual program's row cap)
    const bool primal = static_form_is_primal(gen);
    if (gen.rows < 1 || solve_rows < 1 || (!primal && solve_rows > static_max_rows()) || gen.rows > 64 || gen.nv > 36 || gen.generic.ws_words > 2400) return false;
    if (!rtc_api().ok) return false;
    if (!compile) return true;
    const std::string src = generic_static_source(gen);
    const uint64_t key = source_hash(with_defines(src));
    if (key_out) *key_out = key;
    return ensure_code(g_gen_codes, key, "generic_static", src, /*if_convert=*/true).ok;
}

// The refill program of a static lane program (programs above 12 rows spill, and a spilling program is kept off the refill path
// below: do not even compile it).
static bool refill_program_worth_compiling(const ProblemHost &gen) { return rtc_static_solve_rows(gen) <= 12 && !static_form_is_primal(gen); }

static HotCode &refill_code(const ProblemHost &gen) {   // (called WITHOUT g_mu: the compiler may run)
    const std::string src = generic_static_source(gen, kStaticDlsRefill);
    return ensure_code(g_gen_codes, source_hash(with_defines(src)), "generic_static_refill", src, /*if_convert=*/true);
}

// (g_mu held; `hc` tried) loads the refill program into gm, or leaves gm.refill == nullptr
static void load_generic_refill(const HotCode &hc, GenModule &gm) {
    if (gm.refill_tried) return;
    gm.refill_tried = true;
    if (!hc.ok) return;
    if (hipModuleLoadData(&gm.refill_mod, hc.code.data()) != hipSuccess) { gm.refill_mod = nullptr; return; }
    if (hipModuleGetFunction(&gm.refill, gm.refill_mod, "ikgpu_lane_dls_refill") != hipSuccess) { gm.refill = nullptr; return; }
    // A program that spills is kept off the refill path: its loop has divergent regions (a done lane stores and reloads), and a
    // spill store the register allocator places inside one is the hazard tools/spill_exec_check.py describes.  The lock-step
    // program's loop has none (tools/static_program_check.sh).
    int local_bytes = 0;
    if (hipFuncGetAttribute(&local_bytes, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, gm.refill) != hipSuccess || local_bytes > 0) { gm.refill = nullptr; return; }
    int per_cu = 0;
    if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gm.refill, 64, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    gm.refill_waves_per_cu = per_cu;
}

static hipError_t generic_module(uint64_t key, GenModule **out) {   // (g_mu held)
    int dev = 0;
    (void)hipGetDevice(&dev);
    GenModule &gm = g_gen_modules[std::make_pair(key, dev)];
    if (!gm.mod) {
        const HotCode &hc = g_gen_codes[key];
        if (!hc.ok) return hipErrorInvalidImage;
        hipError_t e = hipModuleLoadData(&gm.mod, hc.code.data());
        if (e == hipSuccess) e = hipModuleGetFunction(&gm.dls, gm.mod, "ikgpu_lane_dls");
        if (e != hipSuccess) { gm = GenModule{}; return e; }
    }
    *out = &gm;
    return hipSuccess;
}

bool rtc_generic_static_precompile_refill(const ProblemHost &gen) {   // (compiles / fetches the code object; modules load per device at launch)
    if (!refill_program_worth_compiling(gen)) return false;
    return refill_code(gen).ok;
}

hipError_t rtc_launch_generic_static(const ProblemHost &gen, uint64_t key, const BatchIO &io, const ikgpu_dls_params &prm, hipStream_t stream,
                                     QueuePool *queues) {
    struct Args {
        ikdev::GenericKernelArgs a;
        unsigned long long *queue;
        int chunk;
    } args{};
    static_assert(sizeof(ikdev::GenericKernelArgs) % 8 == 0, "argument layout");
    ikdev::GenericKernelArgs &a = args.a;
    a.prm.max_iterations = prm.max_iterations;
    a.prm.lam2 = prm.damping * prm.damping;
    a.prm.step_length = prm.step_length;
    a.prm.stop_sq_tol = prm.stop_sq_tol;
    fill_visitor(a.prm, prm);
    a.layout = io.layout; a.B = io.B; a.q0 = io.q0; a.targets = io.targets;
    a.q_out = io.q_out; a.success = io.success; a.iters = io.iters;
    // stop-rule mode on a batch larger than the machine: lane refill (as the chain and tree kernels; `refill_wanted` is false at or
    // below one wave per SIMD, so the small-batch path never meets the compiler here)
    const bool maybe_refill = queues && refill_wanted(prm, io.B, 1024);
    GenModule m;
    const HotCode *refill_obj = nullptr;
    for (int pass = 0;; ++pass) {
        {
            std::lock_guard<std::mutex> lock(g_mu);
            GenModule *gm = nullptr;
            const hipError_t e = generic_module(key, &gm);
            if (e != hipSuccess) return e;
            if (pass == 1) {
                if (refill_obj) load_generic_refill(*refill_obj, *gm);
                else gm->refill_tried = true;
            }
            m = *gm;
        }
        if (pass == 1 || !maybe_refill || m.refill_tried) break;
        // the first large stop-rule batch of this program: its refill twin is compiled (or fetched) now, with no lock held
        if (refill_program_worth_compiling(gen)) refill_obj = &refill_code(gen);
    }
    auto launch = [&](hipFunction_t fn, int64_t grid, size_t nbytes) {
        void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &nbytes, HIP_LAUNCH_PARAM_END};
        return hipModuleLaunchKernel(fn, static_cast<unsigned>(grid), 1, 1, 64, 1, 1, 0, stream, nullptr, config);
    };
    if (maybe_refill && m.refill) {
        int cus = 256;
        {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
        }
        const int64_t resident = refill_resident(static_cast<int64_t>(m.refill_waves_per_cu) * cus, io.B);
        // (the switch point of the tree kernels: a static lane program is as heavy, one wave per SIMD in many rounds -- kernels.hpp)
        const int mode = stop_rule_mode(prm, io.B, resident, stream, true);
        if (mode == kStopRefill) {
            hipError_t e = hipSuccess;
            args.queue = queues->slot_for(stream, &e);
            if (!args.queue) return e;
            args.chunk = refill_chunk(io.B, resident);
            return launch(m.refill, resident, offsetof(Args, chunk) + sizeof(int));
        }
        if (mode == kStopTwoPhase) {   // (kernels.hpp run_two_phase: the lock-step program until a wave's stragglers are few, the refill twin on those)
            hipError_t le = hipSuccess;
            const hipError_t te = run_two_phase(*queues, io, stream, a, true, [&] { le = launch(m.dls, (io.B + 63) / 64, sizeof(ikdev::GenericKernelArgs)); },
                                                [&](unsigned long long *queue) {
                                                    args.queue = queue;
                                                    args.chunk = refill_chunk(io.B, resident);
                                                    const hipError_t r = launch(m.refill, resident, offsetof(Args, chunk) + sizeof(int));
                                                    if (le == hipSuccess) le = r;
                                                });
            return le != hipSuccess ? le : te;
        }
    }
    return launch(m.dls, (io.B + 63) / 64, sizeof(ikdev::GenericKernelArgs));
}

}  // namespace ikgpu

namespace ikgpu {

// ---- ik::pik as a compiled lane program (device/pik_solver.hpp static_pik) -------------------------------------------------------------
// Eligibility: what the DLS program takes (rows, nv, workspace), every level within 12 rows (one level without a secondary step IS
// the DLS iteration and never gets here: capi.cpp routes it to the problem's DLS kernel) (its dual system is unrolled in registers).  lambda > 0 on every
// level is a property of the CALL (capi.cpp checks it): lambda = 0 stays on the one-sided-Jacobi interpreter.
namespace {
struct PikModule {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
};
std::map<std::pair<uint64_t, int>, PikModule> g_pik_modules;   // (hash, device); guarded by g_mu
}  // namespace

bool rtc_pik_static_available(const ProblemHost &gen, bool with_da, bool compile, uint64_t *key_out) {
    if (!rtc_enabled() || gen.kind != KernelKind::Generic || !rtc_api().ok) return false;
    if (std::getenv("IKGPU_GENERIC_KERNEL")) return false;
    if (const char *env = std::getenv("IKGPU_PIK_STATIC")) { if (env[0] == '0') return false; }
    const GenericHost &g = gen.generic;
    int levels = 0, widest = 0;
    for (int l = 0; l < g.nlevels; ++l) {
        const int ml = g.ints[static_cast<size_t>(g.o_lvlrow0 + l + 1)] - g.ints[static_cast<size_t>(g.o_lvlrow0 + l)];
        levels += ml > 0 ? 1 : 0;
        widest = std::max(widest, ml);
    }
    if (levels < 1 || widest > 12 || gen.rows > static_max_rows() || gen.nv > 36 || g.ws_words > 2400 || gen.crows > 0) return false;
    if (!compile) return true;
    const std::string src = generic_static_source(gen, with_da ? kStaticPikDa : kStaticPik);
    const uint64_t key = source_hash(with_defines(src));
    if (key_out) *key_out = key;
    return ensure_code(g_gen_codes, key, with_da ? "pik_static_da" : "pik_static", src, /*if_convert=*/true).ok;
}

hipError_t rtc_launch_pik_static(const ProblemHost &gen, uint64_t key, const BatchIO &io, const ikgpu_pik_params &prm, hipStream_t stream) {
    ikdev::PikKernelArgs a{};
    a.prm.max_iterations = prm.max_iterations;
    a.prm.step_length = prm.step_length;
    a.prm.stop_sq_tol = prm.stop_sq_tol;
    for (int l = 0; l < ikdev::kMaxPikLevels; ++l) a.prm.lam2[l] = l < prm.num_levels ? prm.lambda[l] * prm.lambda[l] : 1.0;
    a.prm.has_da = 0;
    if (prm.da)
        for (int k = 0; k < gen.nv && k < ikdev::kMaxPikDa; ++k) {
            a.prm.da[k] = prm.da[k];
            if (prm.da[k] != 0.0) a.prm.has_da = 1;
        }
    a.layout = io.layout; a.B = io.B; a.q0 = io.q0; a.targets = io.targets;
    a.q_out = io.q_out; a.success = io.success; a.iters = io.iters;
    PikModule m;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        int dev = 0;
        (void)hipGetDevice(&dev);
        PikModule &pm = g_pik_modules[std::make_pair(key, dev)];
        if (!pm.mod) {
            const HotCode &hc = g_gen_codes[key];
            if (!hc.ok) return hipErrorInvalidImage;
            hipError_t e = hipModuleLoadData(&pm.mod, hc.code.data());
            if (e == hipSuccess) e = hipModuleGetFunction(&pm.fn, pm.mod, "ikgpu_lane_pik");
            if (e != hipSuccess) { pm = PikModule{}; return e; }
        }
        m = pm;
    }
    size_t nbytes = sizeof a;
    void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &nbytes, HIP_LAUNCH_PARAM_END};
    return hipModuleLaunchKernel(m.fn, static_cast<unsigned>((io.B + 63) / 64), 1, 1, 64, 1, 1, 0, stream, nullptr, config);
}

}  // namespace ikgpu

// ---- the compile worker's entry (include/ikgpu.h ikgpu_rtc_worker_compile; tools/ikgpu_precompile.cpp --request) ---------------------
// Runs in the process `run_worker` spawned: reads the request, compiles in THIS process, leaves the code object in the cache.  Touches
// no device.  IKGPU_RTC_WORKER_FAULT = abort | segv | hang | exit (fault injection for tests/test_rtc_containment.py: what the
// caller's process must survive).
extern "C" int ikgpu_rtc_worker_compile(const char *request_path) {
    using namespace ikgpu;
    if (!request_path) return 2;
    g_in_worker.store(true);
    if (const char *f = std::getenv("IKGPU_RTC_WORKER_FAULT")) {
        std::fprintf(stderr, "ikgpu_precompile: injected fault '%s'\n", f);
        std::fflush(stderr);
        if (std::strcmp(f, "abort") == 0) std::abort();
        if (std::strcmp(f, "segv") == 0) (void)raise(SIGSEGV);
        if (std::strcmp(f, "hang") == 0) for (;;) (void)pause();
        if (std::strcmp(f, "exit") == 0) return 3;
    }
    std::vector<char> raw;
    if (!read_file(request_path, raw)) { std::fprintf(stderr, "ikgpu_precompile: cannot read %s\n", request_path); return 2; }
    const std::string text(raw.begin(), raw.end());
    size_t pos = 0;
    auto line = [&]() {
        const size_t nl = text.find('\n', pos);
        if (nl == std::string::npos) { pos = text.size(); return std::string(); }
        std::string l = text.substr(pos, nl - pos);
        pos = nl + 1;
        return l;
    };
    if (line() != "IKGPU-RTC-REQUEST 1") { std::fprintf(stderr, "ikgpu_precompile: %s is not a compile request\n", request_path); return 2; }
    const std::string prefix = line();
    const bool if_convert = line() == "1";
    const size_t n = static_cast<size_t>(std::strtoull(line().c_str(), nullptr, 10));
    if (prefix.empty() || prefix.find('/') != std::string::npos || n == 0 || pos + n != text.size()) {
        std::fprintf(stderr, "ikgpu_precompile: malformed compile request %s\n", request_path);
        return 2;
    }
    HotCode hc;
    compile_cached(prefix.c_str(), text.substr(pos, n), hc, if_convert);
    if (!hc.ok) { std::fprintf(stderr, "%s\n", hc.log.c_str()); return 1; }
    return 0;
}
