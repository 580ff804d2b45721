// model.hpp -- host-side flat kinematic model with Pinocchio's conventions.
// Stands in for pinocchio::Model as the IK path reads it (reference ik/ik/common.hpp:16,
// ik/ik/common.hpp:47-56) and for pinocchio::urdf::buildModelFromXML (reference
// ik_ros/src/cassie.cpp:34-35).  Conventions restated from SURVEY.md Appendix A.1.
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "ikgpu.h"

namespace ikgpu {

using SE3 = std::array<double, 12>;  // rotation row-major (9) + translation (3)

inline SE3 se3_identity() { return SE3{1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}; }
SE3 se3_mul(const SE3 &a, const SE3 &b);

struct Model {
    int32_t nq = 0, nv = 0;
    std::vector<int32_t> joint_type, joint_parent, joint_idx_q, joint_idx_v;
    std::vector<SE3> joint_placement;
    std::vector<std::array<double, 3>> joint_axis;
    std::vector<double> lower, upper;
    std::vector<std::string> joint_names;
    std::vector<int32_t> frame_parent;
    std::vector<SE3> frame_placement;
    std::vector<std::string> frame_names;
    // mass and lever (centre of mass in the joint frame) of the bodies attached to each joint: what
    // pinocchio::centerOfMass reads of model.inertias[j]
    std::vector<double> joint_mass;
    std::vector<std::array<double, 3>> joint_com;
    // appendBodyToJoint restricted to (mass, lever): inertias[joint] += placement.act(Y)
    void append_body(int32_t joint, const SE3 &placement, double mass, const std::array<double, 3> &com);
    double total_mass() const;  // of joints 1.. (bodies welded to the universe do not count, as in Pinocchio)
    // C views handed out by ikgpu_model_get_flat
    std::vector<const char *> joint_name_ptrs, frame_name_ptrs;

    int32_t njoints() const { return static_cast<int32_t>(joint_type.size()); }
    int32_t nframes() const { return static_cast<int32_t>(frame_parent.size()); }
    int32_t frame_id(const std::string &name) const;  // nframes() when absent, as Model::getFrameId
    int32_t joint_id(const std::string &name) const;  // njoints() when absent
    void finalize();                                  // builds the C views, validates sizes

    // Throws std::runtime_error with a message on malformed / unsupported input.
    static Model from_urdf(const char *xml, size_t len, bool free_flyer);
    static Model from_flat(const ikgpu_flat_model &f);
};

}  // namespace ikgpu

// The opaque C handle is the C++ object.
struct ikgpu_model {
    ikgpu::Model m;
};
