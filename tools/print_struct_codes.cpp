// Prints the placement-structure codes (ikgpu::chain_structure) of the chains of the fixture models: the list of
// instantiations in ik_amd/csrc/kernels_hot.hip is generated from this output.
//   g++ -O1 -std=c++17 -Iinclude -Iik_amd/csrc tools/print_struct_codes.cpp ik_amd/csrc/model.cpp ik_amd/csrc/problem.cpp -o /tmp/psc && /tmp/psc
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>

#include "model.hpp"
#include "problem.hpp"

int main(int argc, char **argv) {
    const char *cases[][2] = {{"cassie_fixed", "LeftFootFront"}, {"cassie_fixed", "RightFootFront"}, {"ur5", "tool0"}, {"ur10", "tool0"},
                              {"ur5", "ee_link"}, {"cassie_fixed", "LeftFootBack"}};
    const std::string dir = argc > 1 ? argv[1] : "fixtures/models";
    for (auto &cs : cases) {
        std::ifstream f(dir + "/" + cs[0] + ".kin.urdf");
        std::stringstream ss; ss << f.rdbuf();
        const std::string xml = ss.str();
        try {
            ikgpu::Model m = ikgpu::Model::from_urdf(xml.data(), xml.size(), false);
            ikgpu_task t{};
            t.frame = m.frame_id(cs[1]); t.reference = 0; t.type = IKGPU_FULL; t.priority = 0;
            for (double &w : t.weight) w = 1.0;
            ikgpu::ProblemHost ph = ikgpu::analyse_problem(m, &t, 1);
            ikgpu::ChainStructure s = ikgpu::chain_structure(ph.chain);
            std::printf("    X(%d, 0x%016llxull, 0x%016llxull, 0x%016llxull)  /* %s %s: %d table values */ \\\n", ph.chain.nj,
                        (unsigned long long)s.code[0], (unsigned long long)s.code[1], (unsigned long long)s.code[2], cs[0], cs[1], s.values);
        } catch (const std::exception &e) { std::printf("// %s %s: %s\n", cs[0], cs[1], e.what()); }
    }
    return 0;
}
