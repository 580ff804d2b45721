// test_demo_loop.cpp -- the reference's only real caller of ik::dls(), the Cassie demo
// (reference ik_ros/src/cassie.cpp:19-130), written against the C++ mirror with ROS taken out:
// same model (free-flyer Cassie), same tasks (left foot position w.r.t. the pelvis, pelvis pose in
// the world, foot Y axis aligned with X), same parameters (damping 0.1, 200 iterations, step 0.1),
// warm-started tick after tick.  Prints q after each tick; tests/test_gpu_generic.py compares with
// the CPU oracle.       usage: test_demo_loop <cassie.urdf> <ticks>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>

#include "ik/dls.hpp"
#include "ik/problem.hpp"

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    try {
        ik::model_t model;
        ik::urdf::buildModel(argv[1], ik::JointModelFreeFlyer(), model);

        auto ik_ = std::make_unique<ik::InverseKinematicsProblem>(model, 1);
        // Create leg task (with respect to pelvis frame)
        auto fl = ik::FrameTask::create(model, "LeftFootFront", ik::KinematicType::Position, "pelvis");
        fl->weighting().setConstant(1e0);
        // Pelvis pose tracking task in world frame
        auto pelvis = ik::FrameTask::create(model, "pelvis", ik::KinematicType::Full);
        auto foot_alignment = ik::AlignAxisTask::create(model, "LeftFootFront", ik::AlignAxisType::AxisY);
        foot_alignment->target = ik::vector3_t::UnitX();

        // Create configuration vector, quaternion w component to 1.0
        ik::vector_t q_ = ik::vector_t::Zero(model.nq);
        q_[6] = 1.0;

        ik_->add_frame_task("fl", fl);
        ik_->add_frame_task("pelvis", pelvis);
        ik_->add_align_axis_task("align", foot_alignment);
        auto dls_data_ = std::make_unique<ik::dls_data>(*ik_);

        std::printf("{\"kernel\": \"%s\", \"ticks\": [", dls_data_->kernel());
        const int ticks = std::atoi(argv[2]);
        for (int k = 0; k < ticks; ++k) {
            const double t = static_cast<double>(k);
            ik_->get_frame_task("fl")->target.translation() << 0.0, 0.1, -0.6 + 0.2 * std::sin(0.5 * t);
            ik_->get_frame_task("pelvis")->target.translation().setZero();
            ik_->get_frame_task("pelvis")->target.rotation().setIdentity();

            ik::dls_parameters p;
            p.damping = 1e-1;
            p.max_iterations = 200;
            p.step_length = 1e-1;
            q_ = ik::dls(*ik_, q_, *dls_data_, ik::inverse_kinematics_visitor(), p);

            std::printf("%s{\"success\": %d, \"iterations\": %zu, \"q\": [", k ? ", " : "", dls_data_->success ? 1 : 0, dls_data_->iterations);
            for (ik::index_t i = 0; i < q_.size(); ++i) std::printf("%s%.17g", i ? ", " : "", q_[i]);
            std::printf("]}");
        }
        std::printf("]}\n");
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
