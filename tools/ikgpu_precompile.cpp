// ikgpu_precompile -- the run-time compiler of libikgpu, in a process of its own.  Installed next to libikgpu.so (ik_amd/csrc/Makefile).
//
// Two uses:
//   1. deployment:  ikgpu_precompile --urdf robot.urdf [--free-flyer] --task FRAME[:REFERENCE[:TYPE[:PRIORITY[:w1,w2,...]]]] ...
//                                    [--constraint FRAME[:REFERENCE[:TYPE]]] ...
//      compiles whatever ikgpu_problem_create would compile for this problem (the structure-specialised chain kernel of a chain
//      without a pre-built instantiation, the static lane program of a generic problem, their refill twins) into the on-disk cache
//      ($IKGPU_CACHE_DIR, else $XDG_CACHE_HOME/ikgpu, else ~/.cache/ikgpu) and prints the name of the kernel the problem will run
//      on.  Run it once per robot / task list / library build; afterwards ikgpu_problem_create in the control process is a cache
//      hit: no compiler and no child process run there (include/ikgpu.h).  TYPE: position | orientation | full (default) | align-x |
//      align-y | align-z; REFERENCE defaults to "universe".  Stands in for nothing in the reference (its kernels are compiled with
//      the library); the problem description mirrors InverseKinematicsProblem::add_frame_task / add_align_axis_task /
//      add_frame_constraint (reference ik/ik/problem.hpp:55-105).
//   2. the library's own compile worker:  ikgpu_precompile --request FILE   (spawned by rtc.cpp on a cache miss; FILE is written by
//      the library).  A crash of the compiler ends THIS process; the caller falls back to its general kernel.
// Touches no device in either mode.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "ikgpu.h"

namespace {

std::vector<std::string> split(const std::string &s, char sep) {
    std::vector<std::string> out;
    std::string item;
    std::istringstream is(s);
    while (std::getline(is, item, sep)) out.push_back(item);
    return out;
}

int type_of(const std::string &t) {
    if (t.empty() || t == "full") return IKGPU_FULL;
    if (t == "position") return IKGPU_POSITION;
    if (t == "orientation") return IKGPU_ORIENTATION;
    if (t == "align-x") return IKGPU_ALIGN_AXIS_X;
    if (t == "align-y") return IKGPU_ALIGN_AXIS_Y;
    if (t == "align-z") return IKGPU_ALIGN_AXIS_Z;
    return -1;
}

bool parse_row(const ikgpu_model *m, const std::string &spec, int32_t nframes, ikgpu_task &k) {
    const std::vector<std::string> f = split(spec, ':');
    if (f.empty() || f[0].empty()) return false;
    k.frame = ikgpu_model_frame_id(m, f[0].c_str());
    k.reference = ikgpu_model_frame_id(m, f.size() > 1 && !f[1].empty() ? f[1].c_str() : "universe");
    k.type = type_of(f.size() > 2 ? f[2] : "");
    k.priority = f.size() > 3 && !f[3].empty() ? std::atoi(f[3].c_str()) : 0;
    for (double &w : k.weight) w = 1.0;
    if (f.size() > 4) {
        const std::vector<std::string> w = split(f[4], ',');
        for (size_t i = 0; i < w.size() && i < 6; ++i) k.weight[i] = std::atof(w[i].c_str());
    }
    if (k.frame < 0 || k.frame >= nframes) { std::fprintf(stderr, "ikgpu_precompile: no frame '%s' in the model\n", f[0].c_str()); return false; }
    if (k.reference < 0 || k.reference >= nframes) { std::fprintf(stderr, "ikgpu_precompile: no frame '%s' in the model\n", f[1].c_str()); return false; }
    if (k.type < 0) { std::fprintf(stderr, "ikgpu_precompile: unknown task type '%s'\n", f[2].c_str()); return false; }
    return true;
}

int usage() {
    std::fprintf(stderr, "usage: ikgpu_precompile --urdf FILE [--free-flyer] --task FRAME[:REFERENCE[:TYPE[:PRIORITY[:w1,w2,...]]]] ... "
                         "[--constraint FRAME[:REFERENCE[:TYPE]]] ...\n"
                         "       ikgpu_precompile --request FILE        (internal: the library's compile worker)\n");
    return 2;
}

}  // namespace

int main(int argc, char **argv) {
    std::string urdf;
    bool free_flyer = false;
    std::vector<std::string> task_specs, cons_specs;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--request" && i + 1 < argc) return ikgpu_rtc_worker_compile(argv[i + 1]);
        else if (a == "--urdf" && i + 1 < argc) urdf = argv[++i];
        else if (a == "--free-flyer") free_flyer = true;
        else if (a == "--task" && i + 1 < argc) task_specs.push_back(argv[++i]);
        else if (a == "--constraint" && i + 1 < argc) cons_specs.push_back(argv[++i]);
        else return usage();
    }
    if (urdf.empty() || task_specs.empty()) return usage();
    std::ifstream in(urdf, std::ios::binary);
    if (!in) { std::fprintf(stderr, "ikgpu_precompile: cannot read %s\n", urdf.c_str()); return 2; }
    std::stringstream text;
    text << in.rdbuf();
    const std::string xml = text.str();
    ikgpu_model *m = nullptr;
    if (ikgpu_model_from_urdf(xml.data(), xml.size(), free_flyer ? IKGPU_ROOT_FREEFLYER : IKGPU_ROOT_FIXED, &m) != IKGPU_OK) {
        std::fprintf(stderr, "ikgpu_precompile: %s\n", ikgpu_last_error());
        return 1;
    }
    ikgpu_flat_model flat;
    if (ikgpu_model_get_flat(m, &flat) != IKGPU_OK) { std::fprintf(stderr, "ikgpu_precompile: %s\n", ikgpu_last_error()); return 1; }
    std::vector<ikgpu_task> tasks(task_specs.size()), cons(cons_specs.size());
    for (size_t i = 0; i < tasks.size(); ++i)
        if (!parse_row(m, task_specs[i], flat.nframes, tasks[i])) return 2;
    for (size_t i = 0; i < cons.size(); ++i)
        if (!parse_row(m, cons_specs[i], flat.nframes, cons[i])) return 2;
    char name[200] = "";
    const int rc = ikgpu_problem_precompile(m, tasks.data(), static_cast<int32_t>(tasks.size()), cons.empty() ? nullptr : cons.data(),
                                            static_cast<int32_t>(cons.size()), name, sizeof name);
    ikgpu_model_destroy(m);
    if (rc == IKGPU_OK) { std::printf("%s\n", name); return 0; }
    std::fprintf(stderr, "ikgpu_precompile: %s\n", ikgpu_last_error());
    if (name[0]) std::printf("%s\n", name);
    return rc == IKGPU_ERR_UNSUPPORTED ? 3 : 1;
}
