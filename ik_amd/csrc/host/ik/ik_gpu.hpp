// ik_gpu.hpp -- C++ host-side mirror of the reference's API for the DLS path, header-only over the
// C ABI (include/ikgpu.h, libikgpu.so).  A caller of dazzmo/ik keeps its code:
//
//     ik::model_t model;
//     ik::urdf::buildModelFromXML(xml, ik::JointModelFreeFlyer(), model);      // ik_ros/src/cassie.cpp:34-35
//     ik::InverseKinematicsProblem problem(model, 1);                           // ik/ik/problem.hpp:17
//     auto foot = ik::FrameTask::create(model, "LeftFootFront", ik::KinematicType::Full);   // ik/ik/frame.hpp:123
//     problem.add_frame_task("fl", foot);                                       // ik/ik/problem.hpp:55
//     ik::dls_data data(problem);                                               // ik/ik/dls.hpp:36
//     foot->target.translation() << 0.0, 0.1, -0.6;                             // ik_ros/src/cassie.cpp:95
//     ik::dls_parameters p;  p.damping = 1e-2;                                  // ik/ik/dls.hpp:24
//     q = ik::dls(problem, q, data, ik::inverse_kinematics_visitor(), p);       // ik/ik/dls.hpp:111
//
// and gains ik::dls_batch(...) for B problems in lockstep on one MI355X.  Pinocchio and Eigen are not
// needed: the few value types the path touches (vector_t, se3_t, model_t) are provided here with the
// member names the reference code uses.  No arithmetic of the solve happens on the host.
//
// Differences that cannot be hidden, by design of a device path:
//   * a C++ visitor cannot run inside a kernel: inverse_kinematics_visitor carries the one-parameter
//     family of ik/ik/visitor.hpp:19 (`||e[0]||^2 < tolerance`); override stop_tolerance(), not should_stop();
//   * every task kind of the reference is on the device -- FrameTask, AlignAxisTask, PostureTask, CentreOfMassTask -- and
//     so are FrameConstraint (ik::dls) and the prioritised solver ik::pik; ik::dls_data::kernel() names the kernel chosen;
//   * failures of the device call throw std::runtime_error (the reference has no failure channel).
#pragma once

#include <cstddef>
#include <cstdint>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "ikgpu.h"

namespace ik {

typedef int int_t;
typedef std::size_t index_t;
typedef double number_t;
typedef std::string string_t;

// ---- the slice of Eigen::VectorXd the path uses (ik/ik/common.hpp:31) -----------------------------
class vector_t {
   public:
    vector_t() = default;
    explicit vector_t(index_t n, number_t v = 0.0) : d_(n, v) {}
    vector_t(std::initializer_list<number_t> l) : d_(l) {}
    static vector_t Zero(index_t n) { return vector_t(n, 0.0); }
    static vector_t Ones(index_t n) { return vector_t(n, 1.0); }
    static vector_t Constant(index_t n, number_t v) { return vector_t(n, v); }
    index_t size() const { return d_.size(); }
    index_t rows() const { return d_.size(); }
    number_t &operator[](index_t i) { return d_[i]; }
    const number_t &operator[](index_t i) const { return d_[i]; }
    number_t &operator()(index_t i) { return d_[i]; }
    const number_t &operator()(index_t i) const { return d_[i]; }
    number_t *data() { return d_.data(); }
    const number_t *data() const { return d_.data(); }
    void setZero() { for (auto &x : d_) x = 0.0; }
    void setOnes() { for (auto &x : d_) x = 1.0; }
    void setConstant(number_t v) { for (auto &x : d_) x = v; }
    number_t squaredNorm() const { number_t s = 0; for (auto x : d_) s += x * x; return s; }
    bool operator==(const vector_t &o) const { return d_ == o.d_; }

   private:
    std::vector<number_t> d_;
};

// Eigen-style comma initialiser for the fixed-size accessors below: v << a, b, c;
class comma_init {
   public:
    comma_init(number_t *p, int n, number_t first) : p_(p), n_(n), k_(0) { put(first); }
    comma_init &operator,(number_t v) { put(v); return *this; }

   private:
    void put(number_t v) { if (k_ >= n_) throw std::out_of_range("too many coefficients in comma initialiser"); p_[k_++] = v; }
    number_t *p_;
    int n_, k_;
};

class vector3_ref {
   public:
    explicit vector3_ref(number_t *p) : p_(p) {}
    number_t &operator[](int i) { return p_[i]; }
    number_t &operator()(int i) { return p_[i]; }
    comma_init operator<<(number_t v) { return comma_init(p_, 3, v); }
    void setZero() { p_[0] = p_[1] = p_[2] = 0.0; }

   private:
    number_t *p_;
};

class matrix3_ref {  // row-major 3 x 3
   public:
    explicit matrix3_ref(number_t *p) : p_(p) {}
    number_t &operator()(int i, int j) { return p_[3 * i + j]; }
    comma_init operator<<(number_t v) { return comma_init(p_, 9, v); }  // row by row, as Eigen
    void setIdentity() { for (int i = 0; i < 9; ++i) p_[i] = (i % 4 == 0) ? 1.0 : 0.0; }

   private:
    number_t *p_;
};

// ---- pinocchio::SE3 as the path uses it (ik/ik/common.hpp:19; ik/ik/frame.hpp:189) ---------------
class se3_t {
   public:
    se3_t() { setIdentity(); }
    static se3_t Identity() { return se3_t(); }
    void setIdentity() { for (int i = 0; i < 12; ++i) m_[i] = (i < 9 && i % 4 == 0) ? 1.0 : 0.0; }
    matrix3_ref rotation() { return matrix3_ref(m_); }
    vector3_ref translation() { return vector3_ref(m_ + 9); }
    const number_t *data() const { return m_; }  // rotation row-major (9) + translation (3): the ABI's SE(3)
    number_t *data() { return m_; }

   private:
    number_t m_[12];
};

// ---- pinocchio::Model as the path reads it (ik/ik/common.hpp:16) --------------------------------
struct JointModelFreeFlyer {};

class model_t {
   public:
    model_t() = default;
    int nq = 0, nv = 0, njoints = 0, nframes = 0;
    std::vector<string_t> names;        // joint names, "universe" first
    std::vector<string_t> frame_names;
    vector_t lowerPositionLimit, upperPositionLimit;

    index_t getFrameId(const string_t &name) const { return static_cast<index_t>(ikgpu_model_frame_id(h_.get(), name.c_str())); }
    index_t getJointId(const string_t &name) const { return static_cast<index_t>(ikgpu_model_joint_id(h_.get(), name.c_str())); }
    bool existFrame(const string_t &name) const { return getFrameId(name) < static_cast<index_t>(nframes); }
    const ikgpu_model *handle() const { return h_.get(); }

    void adopt(ikgpu_model *h) {
        h_ = std::shared_ptr<ikgpu_model>(h, ikgpu_model_destroy);
        ikgpu_flat_model f;
        if (ikgpu_model_get_flat(h, &f) != IKGPU_OK) throw std::runtime_error(ikgpu_last_error());
        nq = f.nq; nv = f.nv; njoints = f.njoints; nframes = f.nframes;
        names.assign(f.joint_names, f.joint_names + f.njoints);
        frame_names.assign(f.frame_names, f.frame_names + f.nframes);
        lowerPositionLimit = vector_t(f.nq);
        upperPositionLimit = vector_t(f.nq);
        for (int i = 0; i < f.nq; ++i) { lowerPositionLimit[i] = f.lower[i]; upperPositionLimit[i] = f.upper[i]; }
    }

   private:
    std::shared_ptr<ikgpu_model> h_;
};

namespace urdf {  // pinocchio::urdf::buildModel* as called at ik_ros/src/cassie.cpp:34-35, ik/test/dls.cpp:13
inline model_t &buildModelFromXML(const string_t &xml, model_t &model) {
    ikgpu_model *h = nullptr;
    if (ikgpu_model_from_urdf(xml.data(), xml.size(), IKGPU_ROOT_FIXED, &h) != IKGPU_OK) throw std::invalid_argument(ikgpu_last_error());
    model.adopt(h);
    return model;
}
inline model_t &buildModelFromXML(const string_t &xml, const JointModelFreeFlyer &, model_t &model) {
    ikgpu_model *h = nullptr;
    if (ikgpu_model_from_urdf(xml.data(), xml.size(), IKGPU_ROOT_FREEFLYER, &h) != IKGPU_OK) throw std::invalid_argument(ikgpu_last_error());
    model.adopt(h);
    return model;
}
inline string_t read_file(const string_t &filename) {
    std::ifstream in(filename, std::ios::binary);
    if (!in) throw std::invalid_argument("cannot open URDF file " + filename);
    std::ostringstream ss;
    ss << in.rdbuf();
    return ss.str();
}
inline model_t &buildModel(const string_t &filename, model_t &model) { return buildModelFromXML(read_file(filename), model); }
inline model_t &buildModel(const string_t &filename, const JointModelFreeFlyer &ff, model_t &model) {
    return buildModelFromXML(read_file(filename), ff, model);
}
}  // namespace urdf

// ---- ik::Task (ik/ik/task.hpp:19-57) ------------------------------------------------------------
class Task {
   public:
    Task() : dimension_(0) {}
    virtual ~Task() = default;
    index_t dimension() const { return dimension_; }
    vector_t &weighting() { return weighting_; }
    const vector_t &weighting() const { return weighting_; }

   protected:
    void set_dimension(const index_t &dimension) {
        dimension_ = dimension;
        weighting_ = vector_t::Ones(dimension);
    }

   private:
    index_t dimension_;
    vector_t weighting_;
};

enum class KinematicType { Position, Orientation, Full };  // ik/ik/frame.hpp:20

// A task as the device sees it: the ikgpu_task rows it contributes and one 12-double target slot per row.
class DeviceTask : public Task {
   public:
    virtual void abi_rows(int32_t priority, std::vector<ikgpu_task> &out) const = 0;
    virtual void abi_targets(std::vector<number_t> &out) const = 0;
};

// FrameTask and AlignAxisTask: one ABI row whose frame / reference are model frames.
class SingleRowTask : public DeviceTask {
   public:
    virtual int32_t abi_type() const = 0;                 // ikgpu_kinematic_type
    virtual void abi_target(number_t *out12) const = 0;   // the 12-double target slot
    index_t frame_id() const { return frame_id_; }
    index_t reference_id() const { return reference_id_; }
    void abi_rows(int32_t priority, std::vector<ikgpu_task> &out) const override {
        ikgpu_task k;
        k.frame = static_cast<int32_t>(frame_id_);
        k.reference = static_cast<int32_t>(reference_id_);
        k.type = abi_type();
        k.priority = priority;
        for (int i = 0; i < 6; ++i) k.weight[i] = i < static_cast<int>(dimension()) ? weighting()[i] : 1.0;
        out.push_back(k);
    }
    void abi_targets(std::vector<number_t> &out) const override {
        number_t slot[12];
        abi_target(slot);
        out.insert(out.end(), slot, slot + 12);
    }

   protected:
    void resolve(const model_t &model, const string_t &frame, const string_t &reference_frame) {
        frame_id_ = model.getFrameId(frame);
        reference_id_ = model.getFrameId(reference_frame);
        if (frame_id_ >= static_cast<index_t>(model.nframes)) throw std::invalid_argument("Frame not found in model: " + frame);
        if (reference_id_ >= static_cast<index_t>(model.nframes))
            throw std::invalid_argument("Reference frame not found in model: " + reference_frame);
    }
    index_t frame_id_ = 0, reference_id_ = 0;
};

// ---- ik::FrameTask (ik/ik/frame.hpp:78-200) -----------------------------------------------------
class FrameTask : public SingleRowTask {
   public:
    FrameTask(const model_t &model, const string_t &frame, const KinematicType &type = KinematicType::Full,
              const string_t &reference_frame = "universe")
        : target(se3_t::Identity()), type_(type), frame_(frame), reference_frame_(reference_frame) {
        resolve(model, frame, reference_frame);
        set_dimension(type == KinematicType::Full ? 6 : 3);
    }
    static std::shared_ptr<FrameTask> create(const model_t &model, const string_t &frame,
                                             const KinematicType &type = KinematicType::Full,
                                             const string_t &reference_frame = "universe") {
        return std::make_shared<FrameTask>(model, frame, type, reference_frame);
    }
    se3_t target;  // w.r.t. the reference frame; the caller edits it between solves (cassie.cpp:95-99)

    KinematicType type() const { return type_; }
    const string_t &frame() const { return frame_; }
    const string_t &reference_frame() const { return reference_frame_; }
    int32_t abi_type() const override {
        return type_ == KinematicType::Position ? IKGPU_POSITION : type_ == KinematicType::Orientation ? IKGPU_ORIENTATION : IKGPU_FULL;
    }
    void abi_target(number_t *o) const override {
        for (int i = 0; i < 12; ++i) o[i] = target.data()[i];
    }

   protected:
    KinematicType type_;
    string_t frame_, reference_frame_;
};

// ---- ik::AlignAxisTask (ik/ik/frame.hpp:202-319) ------------------------------------------------
enum class AlignAxisType { AxisX = 0, AxisY = 1, AxisZ = 2 };

class vector3_t {  // the slice of Eigen::Vector3d the task uses: target << x, y, z;  target[i]
   public:
    vector3_t() : d_{1.0, 0.0, 0.0} {}
    number_t &operator[](int i) { return d_[i]; }
    const number_t &operator[](int i) const { return d_[i]; }
    comma_init operator<<(number_t v) { return comma_init(d_, 3, v); }
    static vector3_t UnitX() { vector3_t v; v.d_[0] = 1; v.d_[1] = 0; v.d_[2] = 0; return v; }
    static vector3_t UnitY() { vector3_t v; v.d_[0] = 0; v.d_[1] = 1; v.d_[2] = 0; return v; }
    static vector3_t UnitZ() { vector3_t v; v.d_[0] = 0; v.d_[1] = 0; v.d_[2] = 1; return v; }

   private:
    number_t d_[3];
};

class AlignAxisTask : public SingleRowTask {
   public:
    AlignAxisTask(const model_t &model, const string_t &frame, const AlignAxisType &axis,
                  const string_t &reference_frame = "universe")
        : axis_(axis), frame_(frame), reference_frame_(reference_frame) {
        resolve(model, frame, reference_frame);
        set_dimension(index_t(1));
    }
    static std::shared_ptr<AlignAxisTask> create(const model_t &model, const string_t &frame, const AlignAxisType &axis,
                                                 const string_t &reference_frame = "universe") {
        return std::make_shared<AlignAxisTask>(model, frame, axis, reference_frame);
    }
    vector3_t target;  // desired direction of the frame's axis (frame.hpp:307)

    int32_t abi_type() const override { return IKGPU_ALIGN_AXIS_X + static_cast<int32_t>(axis_); }
    void abi_target(number_t *o) const override {
        for (int i = 0; i < 9; ++i) o[i] = (i % 4 == 0) ? 1.0 : 0.0;
        for (int i = 0; i < 3; ++i) o[9 + i] = target[i];
    }

   protected:
    AlignAxisType axis_;
    string_t frame_, reference_frame_;
};

// ---- ik::CentreOfMassTask (ik/ik/centre_of_mass.hpp:14-62) --------------------------------------------
// e = oMr^-1 com(q) - target, J = R(oMr)^T Jcom (pinocchio::jacobianCenterOfMass, ik/ik/data.cpp:31-34).  Needs link
// masses in the model (URDF <inertial>); runs on the generic device kernel.
class CentreOfMassTask : public DeviceTask {
   public:
    CentreOfMassTask(const model_t &model, const string_t &reference_frame = "universe") : reference_frame_(reference_frame) {
        reference_id_ = model.getFrameId(reference_frame);
        if (reference_id_ >= static_cast<index_t>(model.nframes))
            throw std::invalid_argument("Reference frame not found in model: " + reference_frame);
        target << 0.0, 0.0, 0.0;  // the reference leaves it uninitialised; a caller sets it (cassie.cpp:101)
        set_dimension(index_t(3));
    }
    static std::shared_ptr<CentreOfMassTask> create(const model_t &model, const string_t &reference_frame = "universe") {
        return std::make_shared<CentreOfMassTask>(model, reference_frame);
    }
    vector3_t target;  // target point for the centre of mass in the task's reference frame (centre_of_mass.hpp:57)

    void abi_rows(int32_t priority, std::vector<ikgpu_task> &out) const override {
        ikgpu_task k;
        k.frame = 0;
        k.reference = static_cast<int32_t>(reference_id_);
        k.type = IKGPU_CENTRE_OF_MASS;
        k.priority = priority;
        for (int i = 0; i < 6; ++i) k.weight[i] = i < 3 ? weighting()[i] : 1.0;
        out.push_back(k);
    }
    void abi_targets(std::vector<number_t> &out) const override {
        number_t slot[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, target[0], target[1], target[2]};
        out.insert(out.end(), slot, slot + 12);
    }

   private:
    string_t reference_frame_;
    index_t reference_id_ = 0;
};

// ---- ik::Constraint / ik::FrameConstraint (ik/ik/constraint.hpp; ik/ik/frame.hpp:325-449) -------------
// The frame may not move relative to its reference frame in the selected coordinates: ik::dls keeps its step in the
// null space of the stacked constraint Jacobian (ik/ik/dls.cpp:26-34,43-53).  `target` is never read by the loop.
class Constraint {
   public:
    virtual ~Constraint() = default;
    index_t dimension() const { return dimension_; }
    virtual ikgpu_task abi_record() const = 0;

   protected:
    void set_dimension(const index_t &dimension) { dimension_ = dimension; }

   private:
    index_t dimension_ = 0;
};

class FrameConstraint : public Constraint {
   public:
    FrameConstraint(const model_t &model, const string_t &frame, const KinematicType &type = KinematicType::Full,
                    const string_t &reference_frame = "universe")
        : target(se3_t::Identity()), type_(type), frame_(frame), reference_frame_(reference_frame) {
        frame_id_ = model.getFrameId(frame);
        reference_id_ = model.getFrameId(reference_frame);
        if (frame_id_ >= static_cast<index_t>(model.nframes)) throw std::invalid_argument("Frame not found in model: " + frame);
        if (reference_id_ >= static_cast<index_t>(model.nframes))
            throw std::invalid_argument("Reference frame not found in model: " + reference_frame);
        set_dimension(type == KinematicType::Full ? 6 : 3);
    }
    static std::shared_ptr<FrameConstraint> create(const model_t &model, const string_t &frame,
                                                   const KinematicType &type = KinematicType::Full,
                                                   const string_t &reference_frame = "universe") {
        return std::make_shared<FrameConstraint>(model, frame, type, reference_frame);
    }
    se3_t target;

    ikgpu_task abi_record() const override {
        ikgpu_task k;
        k.frame = static_cast<int32_t>(frame_id_);
        k.reference = static_cast<int32_t>(reference_id_);
        k.type = type_ == KinematicType::Position ? IKGPU_POSITION : type_ == KinematicType::Orientation ? IKGPU_ORIENTATION : IKGPU_FULL;
        k.priority = 0;
        for (double &w : k.weight) w = 1.0;
        return k;
    }

   protected:
    KinematicType type_;
    string_t frame_, reference_frame_;
    index_t frame_id_ = 0, reference_id_ = 0;
};

// ---- ik::PostureTask (ik/ik/posture.hpp:17-85) --------------------------------------------------
// e = (q.bottomRows(nj) - target) .* mask, J.rightCols(nj) = I (the reference leaves the mask out of J).  Crosses the
// ABI as nj one-row tasks of type IKGPU_POSTURE_ROW; runs on the generic device kernel.
class PostureTask : public DeviceTask {
   public:
    PostureTask(const model_t &model, const index_t &nj)
        : target(vector_t::Zero(nj)), mask(vector_t::Ones(nj)), nj_(nj), q0_(model.nq - nj), v0_(model.nv - nj) {
        if (nj == 0 || nj > static_cast<index_t>(model.nv) || nj > static_cast<index_t>(model.nq))
            throw std::invalid_argument("PostureTask: nj out of range for this model");
        set_dimension(nj);
    }
    static std::shared_ptr<PostureTask> create(const model_t &model, const index_t &nj) {
        return std::make_shared<PostureTask>(model, nj);
    }
    vector_t target;  // posture.hpp:75
    vector_t mask;    // posture.hpp:82

    void abi_rows(int32_t priority, std::vector<ikgpu_task> &out) const override {
        for (index_t k = 0; k < nj_; ++k) {
            ikgpu_task r;
            r.frame = static_cast<int32_t>(v0_ + k);      // tangent column
            r.reference = static_cast<int32_t>(q0_ + k);  // index in q
            r.type = IKGPU_POSTURE_ROW;
            r.priority = priority;
            r.weight[0] = weighting()[k];
            r.weight[1] = mask[k];
            for (int i = 2; i < 6; ++i) r.weight[i] = 1.0;
            out.push_back(r);
        }
    }
    void abi_targets(std::vector<number_t> &out) const override {
        for (index_t k = 0; k < nj_; ++k) {
            number_t slot[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            slot[9] = target[k];
            out.insert(out.end(), slot, slot + 12);
        }
    }

   protected:
    index_t nj_, q0_, v0_;
};

// ---- ik::InverseKinematicsProblem (ik/ik/problem.hpp:9-206), frame + align-axis + posture tasks ------
class InverseKinematicsProblem {
   public:
    InverseKinematicsProblem(const model_t &model, const std::size_t &max_priority_level = 0)
        : model_(model), top_level_(max_priority_level), by_level_(max_priority_level + 1) {}

    const std::size_t &max_priority_level() const { return top_level_; }
    std::size_t e_size(const std::size_t &priority) const {  // rows of one priority level (problem.hpp:38-45)
        std::size_t rows = 0;
        for (const auto &t : get_all_tasks(priority)) rows += t->dimension();
        return rows;
    }
    std::size_t c_size() const {  // rows of all hard constraints (problem.hpp:47-53)
        std::size_t rows = 0;
        for (const auto &c : constraints_) rows += c->dimension();
        return rows;
    }

    // The reference keeps one vector + one name map per task class (problem.hpp:183-206).  Here ONE registry holds every named
    // object: (kind, name) -> the object, first registration of a name wins (as unordered_map::insert does there); the per-level
    // lists the solvers walk are kept beside it in insertion order.
    std::shared_ptr<FrameConstraint> add_frame_constraint(const string_t &name, const std::shared_ptr<FrameConstraint> &constraint) {
        remember(kConstraint, name, constraint);  // problem.hpp:68-77
        constraints_.push_back(constraint);
        ++generation_;
        return constraint;
    }
    std::shared_ptr<FrameConstraint> get_frame_constraint(const string_t &name) { return lookup<FrameConstraint>(kConstraint, name, "Frame constraint"); }
    std::vector<std::shared_ptr<Constraint>> get_all_constraints() const {  // problem.hpp:167-173
        return std::vector<std::shared_ptr<Constraint>>(constraints_.begin(), constraints_.end());
    }

    std::shared_ptr<FrameTask> add_frame_task(const string_t &name, const std::shared_ptr<FrameTask> &task, const std::size_t &priority = 0) {
        return enlist(kFrame, name, task, priority);  // problem.hpp:55-66
    }
    std::shared_ptr<FrameTask> get_frame_task(const string_t &name) { return lookup<FrameTask>(kFrame, name, "Frame task"); }
    std::size_t get_frame_task_index(const string_t &name) {  // position among the frame tasks, in registration order
        const auto it = registry_.find({kFrame, name});
        if (it == registry_.end()) throw std::out_of_range("Frame task does not exist: " + name);
        return it->second.ordinal;
    }
    std::shared_ptr<AlignAxisTask> add_align_axis_task(const string_t &name, const std::shared_ptr<AlignAxisTask> &task,
                                                       const std::size_t &priority = 0) {  // problem.hpp:94-105
        return enlist(kAlign, name, task, priority);
    }
    // (the reference looks the name up among the FRAME tasks, problem.hpp:109; here the align-axis tasks have their own names)
    std::shared_ptr<AlignAxisTask> get_align_axis_task(const string_t &name) { return lookup<AlignAxisTask>(kAlign, name, "Align-axis task"); }
    std::shared_ptr<PostureTask> add_posture_task(const string_t &name, const std::shared_ptr<PostureTask> &task,
                                                  const std::size_t &priority = 0) {  // problem.hpp:134-145
        return enlist(kPosture, name, task, priority);
    }
    std::shared_ptr<PostureTask> get_posture_task(const string_t &name) { return lookup<PostureTask>(kPosture, name, "Posture task"); }  // problem.hpp:147-149
    std::shared_ptr<CentreOfMassTask> add_centre_of_mass_task(const std::shared_ptr<CentreOfMassTask> &task,
                                                              const std::size_t &priority = 0) {  // problem.hpp:121-128
        if (registry_.count({kCom, string_t()})) throw std::logic_error("a problem holds one centre-of-mass task");
        return enlist(kCom, string_t(), task, priority);
    }
    std::shared_ptr<CentreOfMassTask> get_centre_of_mass_task() {  // problem.hpp:130-132 (null when none was added)
        const auto it = registry_.find({kCom, string_t()});
        return it == registry_.end() ? nullptr : std::static_pointer_cast<CentreOfMassTask>(it->second.object);
    }
    const std::vector<std::shared_ptr<DeviceTask>> &get_all_tasks(const std::size_t &priority) const { return by_level_.at(priority); }
    const model_t &model() const { return model_; }
    std::size_t generation() const { return generation_; }

   private:
    enum Kind { kFrame, kAlign, kPosture, kCom, kConstraint };
    struct Entry {
        std::shared_ptr<void> object;
        std::size_t ordinal;  // how many objects of this kind were registered before it
    };
    void remember(Kind kind, const string_t &name, std::shared_ptr<void> object) {
        registry_.insert({{kind, name}, Entry{std::move(object), count_[kind]}});
        ++count_[kind];
    }
    template <class T>
    std::shared_ptr<T> enlist(Kind kind, const string_t &name, const std::shared_ptr<T> &task, std::size_t priority) {
        if (priority > top_level_) throw std::out_of_range("Maximum priority level exceeded!");
        remember(kind, name, task);
        by_level_[priority].push_back(task);
        ++generation_;
        return task;
    }
    template <class T>
    std::shared_ptr<T> lookup(Kind kind, const string_t &name, const char *what) const {
        const auto it = registry_.find({kind, name});
        if (it == registry_.end()) throw std::out_of_range(string_t(what) + " does not exist: " + name);
        return std::static_pointer_cast<T>(it->second.object);
    }

    model_t model_;  // a copy, as the reference keeps (problem.hpp:183); the handle inside is shared
    std::size_t top_level_;
    std::vector<std::vector<std::shared_ptr<DeviceTask>>> by_level_;
    std::vector<std::shared_ptr<FrameConstraint>> constraints_;
    std::map<std::pair<int, string_t>, Entry> registry_;
    std::size_t count_[5] = {0, 0, 0, 0, 0};
    std::size_t generation_ = 0;
};

// ---- parameters and the stop rule ---------------------------------------------------------------
struct default_solver_parameters {  // ik/ik/common.hpp:59-66
    index_t max_iterations = 100;
    number_t max_time = 1.0;  // unused by the reference loop
    number_t step_length = 1.0;
};
struct dls_parameters : public default_solver_parameters {  // ik/ik/dls.hpp:24-28
    number_t damping = 1e-2;
    bool random_restart = false;  // unused by the reference loop
};

class inverse_kinematics_visitor {  // ik/ik/visitor.hpp:7-22
   public:
    inverse_kinematics_visitor() = default;
    virtual ~inverse_kinematics_visitor() = default;
    // should_stop(ik, e, dq) := e[0].squaredNorm() < stop_tolerance(); negative: never stop
    virtual number_t stop_tolerance() const { return 1e-4; }
    // The rest of what should_stop(ik, e, dq) is handed (ik/ik/visitor.hpp:15-21): a derived visitor may ALSO stop on a negligible
    // step, dq.squaredNorm() < step_tolerance() (<= 0: off), and may test every priority level instead of level 0 alone:
    // e[l].squaredNorm() < level_tolerances()[l] for every l (empty: the test above).  A C++ body cannot run inside the kernel;
    // these three numbers are what crosses the ABI (ikgpu_dls_params).
    virtual number_t step_tolerance() const { return 0.0; }
    virtual std::vector<number_t> level_tolerances() const { return {}; }
};
class default_inverse_kinematics_visitor : public inverse_kinematics_visitor {};
class never_stop_visitor : public inverse_kinematics_visitor {
   public:
    number_t stop_tolerance() const override { return -1.0; }
};

// ---- ik::dls_data (ik/ik/dls.hpp:34-65; ik/ik/data.hpp:8-29) ------------------------------------
class dls_data {
   public:
    explicit dls_data(const InverseKinematicsProblem &problem, int device = 0) : device_(device) { bind(problem); }
    dls_data(const dls_data &) = delete;
    dls_data &operator=(const dls_data &) = delete;

    bool success = false;  // data.success (ik/ik/data.hpp:18)
    index_t iterations = 0;
    vector_t q, dq;

    const ikgpu_problem *handle() const { return h_.get(); }
    const char *kernel() const { return ikgpu_problem_kernel(h_.get()); }

    // (Re)builds the device-side tables when tasks were added or weights changed since the last call.
    void bind(const InverseKinematicsProblem &problem) {
        std::vector<ikgpu_task> tasks;
        for (std::size_t p = 0; p <= problem.max_priority_level(); ++p)  // stacking order of ik/ik/dls.cpp:20-24
            for (const auto &t : problem.get_all_tasks(p)) t->abi_rows(static_cast<int32_t>(p), tasks);
        std::vector<ikgpu_task> constraints;
        for (const auto &c : problem.get_all_constraints()) constraints.push_back(c->abi_record());
        if (h_ && same(tasks, tasks_) && same(constraints, constraints_)) return;
        ikgpu_problem *h = nullptr;
        if (ikgpu_problem_create_constrained(problem.model().handle(), tasks.data(), static_cast<int32_t>(tasks.size()), constraints.data(),
                                             static_cast<int32_t>(constraints.size()), device_, &h) != IKGPU_OK)
            throw std::runtime_error(ikgpu_last_error());
        h_ = std::shared_ptr<ikgpu_problem>(h, ikgpu_problem_destroy);
        tasks_ = tasks;
        constraints_ = constraints;
        q = vector_t::Zero(problem.model().nq);
        dq = vector_t::Zero(problem.model().nv);
    }

   private:
    static bool same(const std::vector<ikgpu_task> &t, const std::vector<ikgpu_task> &u) {
        if (t.size() != u.size()) return false;
        for (std::size_t i = 0; i < t.size(); ++i) {
            const ikgpu_task &a = t[i], &b = u[i];
            if (a.frame != b.frame || a.reference != b.reference || a.type != b.type || a.priority != b.priority) return false;
            for (int k = 0; k < 6; ++k)
                if (a.weight[k] != b.weight[k]) return false;
        }
        return true;
    }
    int device_;
    std::shared_ptr<ikgpu_problem> h_;
    std::vector<ikgpu_task> tasks_, constraints_;
};

using Problem = InverseKinematicsProblem;   // the name BASELINE.json's north_star uses for ik/ik/problem.hpp:9's class

namespace detail {
inline ikgpu_dls_params to_abi(const inverse_kinematics_visitor &visitor, const dls_parameters &p) {
    ikgpu_dls_params a;
    a.max_iterations = static_cast<int32_t>(p.max_iterations);
    a.damping = p.damping;
    a.step_length = p.step_length;
    a.stop_sq_tol = visitor.stop_tolerance();
    a.dq_sq_tol = visitor.step_tolerance();
    const std::vector<number_t> lt = visitor.level_tolerances();
    if (lt.size() > IKGPU_MAX_VISITOR_LEVELS) throw std::invalid_argument("more level tolerances than the device visitor takes");
    a.num_level_tols = static_cast<int32_t>(lt.size());
    for (int l = 0; l < IKGPU_MAX_VISITOR_LEVELS; ++l) a.level_sq_tol[l] = l < a.num_level_tols ? lt[static_cast<std::size_t>(l)] : 0.0;
    return a;
}
inline std::vector<number_t> gather_targets(const InverseKinematicsProblem &problem) {
    std::vector<number_t> t;
    for (std::size_t p = 0; p <= problem.max_priority_level(); ++p)
        for (const auto &task : problem.get_all_tasks(p)) task->abi_targets(t);
    return t;
}
}  // namespace detail

// ---- ik::dls (ik/ik/dls.hpp:111-114; ik/ik/dls.cpp:5-78): one problem, a batch of one on the device
inline vector_t dls(InverseKinematicsProblem &problem, const vector_t &q0, dls_data &data,
                    const inverse_kinematics_visitor &visitor = inverse_kinematics_visitor(),
                    const dls_parameters &p = dls_parameters()) {
    data.bind(problem);
    if (q0.size() != static_cast<index_t>(problem.model().nq)) throw std::invalid_argument("q0 has the wrong size");
    const std::vector<number_t> targets = detail::gather_targets(problem);
    const ikgpu_dls_params prm = detail::to_abi(visitor, p);
    vector_t q(q0.size());
    uint8_t ok = 0;
    int32_t it = 0;
    if (ikgpu_dls_solve_batch_host(data.handle(), 1, q0.data(), targets.data(), &prm, q.data(), &ok, &it, IKGPU_AOS) != IKGPU_OK)
        throw std::runtime_error(ikgpu_last_error());
    data.success = ok != 0;
    data.iterations = static_cast<index_t>(it);
    data.q = q;
    return q;
}

// B problems in lockstep, HOST buffers: Q0 is nq x B column-major (problem b's q contiguous, as an
// Eigen::MatrixXd of that shape), targets B x ntasks x 12 in the stacking order of the tasks, Q likewise
// nq x B; success / iterations may be null.
inline void dls_batch(InverseKinematicsProblem &problem, std::int64_t B, const number_t *Q0, const number_t *targets,
                      dls_data &data, number_t *Q, std::uint8_t *success, std::int32_t *iterations,
                      const inverse_kinematics_visitor &visitor = inverse_kinematics_visitor(),
                      const dls_parameters &p = dls_parameters()) {
    data.bind(problem);
    const ikgpu_dls_params prm = detail::to_abi(visitor, p);
    if (ikgpu_dls_solve_batch_host(data.handle(), B, Q0, targets, &prm, Q, success, iterations, IKGPU_AOS) != IKGPU_OK)
        throw std::runtime_error(ikgpu_last_error());
}

// The same with DEVICE buffers on the problem's device (component-major: Q0 [nq][B], targets
// [ntasks][12][B]) and a hipStream_t; asynchronous.
inline void dls_batch_device(InverseKinematicsProblem &problem, std::int64_t B, const number_t *Q0, const number_t *targets,
                             dls_data &data, number_t *Q, std::uint8_t *success, std::int32_t *iterations, void *stream,
                             const inverse_kinematics_visitor &visitor = inverse_kinematics_visitor(),
                             const dls_parameters &p = dls_parameters()) {
    data.bind(problem);
    const ikgpu_dls_params prm = detail::to_abi(visitor, p);
    if (ikgpu_dls_solve_batch(data.handle(), B, Q0, targets, &prm, Q, success, iterations, IKGPU_SOA, stream) != IKGPU_OK)
        throw std::runtime_error(ikgpu_last_error());
}

// ---- ik::pik, the prioritised solver (ik/ik/pik.hpp:11-59; ik/ik/pik.cpp:31-103) ---------------------
struct pik_parameters {  // ik/ik/pik.hpp:11-16; the reference loop reads neither damping nor max_time
    int max_iterations = 100;
    double damping = 1e-2;
    double step_length = 1.0;
    double max_time = 1.0;
};

class pik_data : public dls_data {  // ik/ik/pik.hpp:21-50
   public:
    explicit pik_data(const InverseKinematicsProblem &problem, int device = 0)
        : dls_data(problem, device), da(vector_t::Zero(problem.model().nv)) {
        lambda.assign(problem.max_priority_level() + 1, 1.0);  // pik.hpp:24
    }
    vector_t da;                  // secondary step, projected into the null space of every level (pik.cpp:65)
    std::vector<double> lambda;   // damping factor of each priority level (pik.cpp:54-55)
    // the kernel the next ik::pik call runs with the current lambda / da: the problem's DLS kernel for one priority level
    // without a secondary step (ik::pik is then the DLS iteration, see ikgpu_pik_kernel), the PIK kernel otherwise
    const char *kernel() const;
};

namespace detail {
inline ikgpu_pik_params to_abi(const inverse_kinematics_visitor &visitor, const pik_parameters &p, const pik_data &data) {
    ikgpu_pik_params a;
    ikgpu_pik_params_default(&a, static_cast<int32_t>(data.lambda.size()));
    if (data.lambda.size() > IKGPU_MAX_PIK_LEVELS) throw std::invalid_argument("ik::pik on the device takes at most 8 priority levels");
    a.max_iterations = static_cast<int32_t>(p.max_iterations);
    a.step_length = p.step_length;
    a.stop_sq_tol = visitor.stop_tolerance();
    for (std::size_t l = 0; l < data.lambda.size(); ++l) a.lambda[l] = data.lambda[l];
    a.da = data.da.squaredNorm() > 0.0 ? data.da.data() : nullptr;
    return a;
}
}  // namespace detail

inline const char *pik_data::kernel() const {
    const ikgpu_pik_params prm = detail::to_abi(inverse_kinematics_visitor(), pik_parameters(), *this);
    return ikgpu_pik_kernel(handle(), &prm);
}

inline vector_t pik(InverseKinematicsProblem &problem, const vector_t &q0, pik_data &data,
                    const inverse_kinematics_visitor &visitor = inverse_kinematics_visitor(),
                    const pik_parameters &p = pik_parameters()) {
    data.bind(problem);
    if (q0.size() != static_cast<index_t>(problem.model().nq)) throw std::invalid_argument("q0 has the wrong size");
    const std::vector<number_t> targets = detail::gather_targets(problem);
    const ikgpu_pik_params prm = detail::to_abi(visitor, p, data);
    vector_t q(q0.size());
    uint8_t ok = 0;
    int32_t it = 0;
    if (ikgpu_pik_solve_batch_host(data.handle(), 1, q0.data(), targets.data(), &prm, q.data(), &ok, &it, IKGPU_AOS) != IKGPU_OK)
        throw std::runtime_error(ikgpu_last_error());
    data.success = ok != 0;
    data.iterations = static_cast<index_t>(it);
    data.q = q;
    return q;
}

// B problems in lockstep, HOST buffers laid out as for dls_batch.
inline void pik_batch(InverseKinematicsProblem &problem, std::int64_t B, const number_t *Q0, const number_t *targets,
                      pik_data &data, number_t *Q, std::uint8_t *success, std::int32_t *iterations,
                      const inverse_kinematics_visitor &visitor = inverse_kinematics_visitor(),
                      const pik_parameters &p = pik_parameters()) {
    data.bind(problem);
    const ikgpu_pik_params prm = detail::to_abi(visitor, p, data);
    if (ikgpu_pik_solve_batch_host(data.handle(), B, Q0, targets, &prm, Q, success, iterations, IKGPU_AOS) != IKGPU_OK)
        throw std::runtime_error(ikgpu_last_error());
}

// DEVICE buffers (component-major) and a hipStream_t; asynchronous.
inline void pik_batch_device(InverseKinematicsProblem &problem, std::int64_t B, const number_t *Q0, const number_t *targets,
                             pik_data &data, number_t *Q, std::uint8_t *success, std::int32_t *iterations, void *stream,
                             const inverse_kinematics_visitor &visitor = inverse_kinematics_visitor(),
                             const pik_parameters &p = pik_parameters()) {
    data.bind(problem);
    const ikgpu_pik_params prm = detail::to_abi(visitor, p, data);
    if (ikgpu_pik_solve_batch(data.handle(), B, Q0, targets, &prm, Q, success, iterations, IKGPU_SOA, stream) != IKGPU_OK)
        throw std::runtime_error(ikgpu_last_error());
}

}  // namespace ik
