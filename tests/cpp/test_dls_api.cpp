// test_dls_api.cpp -- exercises the C++ mirror of the reference API (ik_amd/csrc/host/ik/) the way the
// reference's own (commented-out) tests do (reference ik/test/dls.cpp:10-76, ik/test/ik.cpp:9-20):
// load a model, create FrameTasks, add them to an InverseKinematicsProblem, set targets, call ik::dls.
// Driven by tests/test_gpu_parity.py, which compares the printed solution with the CPU oracle.
//
//   test_dls_api <urdf> <free_flyer 0|1> <max_it> <damping> <step> <tol> <ntasks>
//                { <frame> <type 0|1|2> <priority> <12 target numbers> } x ntasks   <nq numbers of q0>
//                [ posture <nj> <priority> <weight> <nj target numbers> ]  [ pik <lambda per level ...> ]
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "ik/dls.hpp"
#include "ik/pik.hpp"
#include "ik/posture.hpp"
#include "ik/problem.hpp"

int main(int argc, char **argv) {
    try {
        int a = 1;
        auto next = [&]() -> std::string {
            if (a >= argc) throw std::runtime_error("not enough arguments");
            return argv[a++];
        };
        const std::string urdf_filename = next();
        const bool free_flyer = std::atoi(next().c_str()) != 0;
        ik::dls_parameters p;
        p.max_iterations = std::atoi(next().c_str());
        p.damping = std::atof(next().c_str());
        p.step_length = std::atof(next().c_str());
        const double tol = std::atof(next().c_str());
        const int ntasks = std::atoi(next().c_str());

        // Load a model
        ik::model_t model;
        if (free_flyer) ik::urdf::buildModel(urdf_filename, ik::JointModelFreeFlyer(), model);
        else ik::urdf::buildModel(urdf_filename, model);

        std::size_t max_priority = 0;
        struct Spec { std::string frame; int type; std::size_t prio; double target[12]; };
        std::vector<Spec> specs(ntasks);
        for (auto &s : specs) {
            s.frame = next();
            s.type = std::atoi(next().c_str());
            s.prio = std::atoi(next().c_str());
            for (double &x : s.target) x = std::atof(next().c_str());
            if (s.prio > max_priority) max_priority = s.prio;
        }
        ik::vector_t q0 = ik::vector_t::Zero(model.nq);
        for (int i = 0; i < model.nq; ++i) q0[i] = std::atof(next().c_str());
        std::size_t posture_nj = 0, posture_prio = 0;
        double posture_weight = 1.0;
        std::vector<double> posture_target;
        bool use_pik = false;
        std::vector<double> pik_lambda;
        bool use_com = false;
        std::string com_reference;
        std::size_t com_prio = 0;
        double com_target[3] = {0, 0, 0};
        struct ConstraintSpec { std::string frame, reference; int type; };
        std::vector<ConstraintSpec> constraint_specs;
        while (a < argc) {
            const std::string opt = next();
            if (opt == "posture") {
                posture_nj = std::atoi(next().c_str());
                posture_prio = std::atoi(next().c_str());
                posture_weight = std::atof(next().c_str());
                for (std::size_t i = 0; i < posture_nj; ++i) posture_target.push_back(std::atof(next().c_str()));
                if (posture_prio > max_priority) max_priority = posture_prio;
            } else if (opt == "com") {  // a CentreOfMassTask: <reference frame> <priority> <target x y z>
                com_reference = next();
                com_prio = std::atoi(next().c_str());
                for (double &x : com_target) x = std::atof(next().c_str());
                use_com = true;
                if (com_prio > max_priority) max_priority = com_prio;
            } else if (opt == "constraint") {  // a FrameConstraint: <frame> <type 0|1|2> <reference frame>
                ConstraintSpec c;
                c.frame = next();
                c.type = std::atoi(next().c_str());
                c.reference = next();
                constraint_specs.push_back(c);
            } else if (opt == "pik") {  // solve with ik::pik; the rest of the line is lambda per level
                use_pik = true;
                while (a < argc) pik_lambda.push_back(std::atof(next().c_str()));
            } else {
                throw std::runtime_error("unknown option " + opt);
            }
        }

        ik::InverseKinematicsProblem problem(model, max_priority);
        int k = 0;
        for (auto &s : specs) {
            const ik::KinematicType type = s.type == 0 ? ik::KinematicType::Position
                                           : s.type == 1 ? ik::KinematicType::Orientation : ik::KinematicType::Full;
            auto task = ik::FrameTask::create(model, s.frame, type, "universe");
            problem.add_frame_task("task" + std::to_string(k++), task, s.prio);
        }
        k = 0;
        for (auto &s : specs) {  // targets are edited after registration, as the demo does (cassie.cpp:95-99)
            auto task = problem.get_frame_task("task" + std::to_string(k++));
            task->target.rotation() << s.target[0], s.target[1], s.target[2], s.target[3], s.target[4], s.target[5],
                s.target[6], s.target[7], s.target[8];
            task->target.translation() << s.target[9], s.target[10], s.target[11];
        }
        if (posture_nj) {  // a posture regulariser (reference ik/ik/posture.hpp:17-85; problem.hpp:134-145)
            auto posture = ik::PostureTask::create(model, posture_nj);
            problem.add_posture_task("posture", posture, posture_prio);
            for (std::size_t i = 0; i < posture_nj; ++i) problem.get_posture_task("posture")->target[i] = posture_target[i];
            posture->weighting().setConstant(posture_weight);
        }

        if (use_com) {  // reference ik/ik/centre_of_mass.hpp:28-31; problem.hpp:121-132; cassie.cpp:58,79,101
            problem.add_centre_of_mass_task(ik::CentreOfMassTask::create(model, com_reference), com_prio);
            problem.get_centre_of_mass_task()->target << com_target[0], com_target[1], com_target[2];
        }
        k = 0;
        for (auto &c : constraint_specs) {  // reference ik/ik/problem.hpp:68-77
            const ik::KinematicType type = c.type == 0 ? ik::KinematicType::Position
                                           : c.type == 1 ? ik::KinematicType::Orientation : ik::KinematicType::Full;
            problem.add_frame_constraint("constraint" + std::to_string(k++), ik::FrameConstraint::create(model, c.frame, type, c.reference));
        }
        if (problem.c_size() != problem.get_all_constraints().size() * 3 && constraint_specs.size() == 1 && constraint_specs[0].type != 2)
            throw std::runtime_error("c_size() does not add up");

        struct tol_visitor : ik::inverse_kinematics_visitor {
            double t;
            explicit tol_visitor(double t_) : t(t_) {}
            double stop_tolerance() const override { return t; }
        } visitor(tol);

        // Solve, and once more through the same data object (warm start from the result, as cassie.cpp:112 does)
        ik::vector_t q, q2;
        bool success = false;
        std::size_t iterations = 0;
        std::string kernel;
        if (use_pik) {  // reference ik/ik/pik.hpp:56-59
            ik::pik_data data(problem);
            ik::pik_parameters pp;
            pp.max_iterations = static_cast<int>(p.max_iterations);
            pp.step_length = p.step_length;
            if (pik_lambda.size() != data.lambda.size()) throw std::runtime_error("one lambda per priority level, please");
            data.lambda = pik_lambda;
            q = ik::pik(problem, q0, data, visitor, pp);
            q2 = ik::pik(problem, q, data, visitor, pp);
            success = data.success; iterations = data.iterations; kernel = data.kernel();
        } else {
            ik::dls_data data(problem);
            q = ik::dls(problem, q0, data, visitor, p);
            q2 = ik::dls(problem, q, data, visitor, p);
            success = data.success; iterations = data.iterations; kernel = data.kernel();
        }

        std::printf("{\"kernel\": \"%s\", \"success\": %d, \"iterations\": %zu, \"q\": [", kernel.c_str(), success ? 1 : 0, iterations);
        for (ik::index_t i = 0; i < q2.size(); ++i) std::printf("%s%.17g", i ? ", " : "", q2[i]);
        std::printf("], \"q_first\": [");
        for (ik::index_t i = 0; i < q.size(); ++i) std::printf("%s%.17g", i ? ", " : "", q[i]);
        std::printf("]}\n");
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
