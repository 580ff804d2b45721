// model.cpp -- URDF text -> flat kinematic model, reproducing the conventions of
// pinocchio::urdf::buildModelFromXML (called at reference ik_ros/src/cassie.cpp:34-35):
//   * joint 0 = "universe"; optional free-flyer "root_joint" as joint 1 (limits +-DBL_MAX);
//   * moving joints numbered by depth-first traversal, children visited in byte-lexicographic
//     order of the *joint name* (urdfdom keeps joints in a std::map);
//   * fixed joints folded into the placement of the frames / moving joints below them;
//   * <origin rpy> goes through a unit quaternion (urdfdom Rotation::setFromRPY) and then
//     Eigen's quaternion -> matrix formula;
//   * every URDF joint and link becomes a named frame, in visiting order, after
//     "universe" and "root_joint".
// Only robot-level <link> and <joint> elements are read (a <transmission> may hold its own
// <joint> children: reference ik/test/ur5.urdf:249-296).
#include "model.hpp"

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <stdexcept>

#include "xml_lite.hpp"

namespace ikgpu {

SE3 se3_mul(const SE3 &a, const SE3 &b) {
    SE3 c;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j)
            c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
        c[9 + i] = a[9 + i] + a[3 * i] * b[9] + a[3 * i + 1] * b[10] + a[3 * i + 2] * b[11];
    }
    return c;
}

namespace {

std::vector<double> parse_doubles(const std::string &s, const std::string &what) {
    std::vector<double> out;
    const char *p = s.data(), *end = s.data() + s.size();
    while (p < end) {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
        if (p >= end) break;
        if (*p == '+') ++p;
        double v = 0.0;
        auto r = std::from_chars(p, end, v);
        if (r.ec != std::errc()) throw std::runtime_error("cannot parse number in " + what + ": '" + s + "'");
        out.push_back(v);
        p = r.ptr;
    }
    return out;
}

std::array<double, 3> parse_vec3(const std::string *s, const std::string &what, std::array<double, 3> dflt) {
    if (!s) return dflt;
    auto v = parse_doubles(*s, what);
    if (v.size() != 3) throw std::runtime_error(what + " needs 3 numbers: '" + *s + "'");
    return {v[0], v[1], v[2]};
}

// urdfdom Rotation::setFromRPY followed by Eigen::Quaternion::toRotationMatrix
void rpy_to_rotation(const std::array<double, 3> &rpy, double *R) {
    const double phi = rpy[0] / 2.0, the = rpy[1] / 2.0, psi = rpy[2] / 2.0;
    double x = std::sin(phi) * std::cos(the) * std::cos(psi) - std::cos(phi) * std::sin(the) * std::sin(psi);
    double y = std::cos(phi) * std::sin(the) * std::cos(psi) + std::sin(phi) * std::cos(the) * std::sin(psi);
    double z = std::cos(phi) * std::cos(the) * std::sin(psi) - std::sin(phi) * std::sin(the) * std::cos(psi);
    double w = std::cos(phi) * std::cos(the) * std::cos(psi) + std::sin(phi) * std::sin(the) * std::sin(psi);
    const double n = std::sqrt(x * x + y * y + z * z + w * w);
    x /= n; y /= n; z /= n; w /= n;
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

struct UrdfJoint {
    std::string name, type, parent, child;
    SE3 origin;
    std::array<double, 3> axis;
    double lower = 0.0, upper = 0.0;
};

}  // namespace

int32_t Model::frame_id(const std::string &name) const {
    for (size_t i = 0; i < frame_names.size(); ++i)
        if (frame_names[i] == name) return static_cast<int32_t>(i);
    return nframes();
}

int32_t Model::joint_id(const std::string &name) const {
    for (size_t i = 0; i < joint_names.size(); ++i)
        if (joint_names[i] == name) return static_cast<int32_t>(i);
    return njoints();
}

// pinocchio::InertiaTpl::operator+= restricted to mass and lever, after placing the body: inertias[joint] += placement.act(Y)
void Model::append_body(int32_t joint, const SE3 &pl, double mass, const std::array<double, 3> &c) {
    joint_mass.resize(joint_type.size(), 0.0);
    joint_com.resize(joint_type.size(), {0.0, 0.0, 0.0});
    if (!(mass > 0.0)) return;  // a massless body adds nothing (and 0 / 0 is not worth reproducing)
    const std::array<double, 3> placed = {pl[0] * c[0] + pl[1] * c[1] + pl[2] * c[2] + pl[9], pl[3] * c[0] + pl[4] * c[1] + pl[5] * c[2] + pl[10],
                                          pl[6] * c[0] + pl[7] * c[1] + pl[8] * c[2] + pl[11]};
    const double mab = joint_mass[joint] + mass, mab_inv = 1.0 / mab;
    for (int i = 0; i < 3; ++i) {
        joint_com[joint][i] *= joint_mass[joint] * mab_inv;
        joint_com[joint][i] += (mass * mab_inv) * placed[i];
    }
    joint_mass[joint] = mab;
}

double Model::total_mass() const {
    double s = 0.0;
    for (size_t j = 1; j < joint_mass.size(); ++j) s += joint_mass[j];
    return s;
}

void Model::finalize() {
    joint_mass.resize(joint_type.size(), 0.0);
    joint_com.resize(joint_type.size(), {0.0, 0.0, 0.0});
    const size_t nj = joint_type.size(), nf = frame_parent.size();
    if (joint_parent.size() != nj || joint_idx_q.size() != nj || joint_idx_v.size() != nj ||
        joint_placement.size() != nj || joint_axis.size() != nj || joint_names.size() != nj ||
        frame_placement.size() != nf || frame_names.size() != nf || lower.size() != static_cast<size_t>(nq) ||
        upper.size() != static_cast<size_t>(nq))
        throw std::runtime_error("inconsistent model array sizes");
    if (nj == 0 || joint_type[0] != IKGPU_JOINT_UNIVERSE) throw std::runtime_error("joint 0 must be the universe");
    int32_t cq = 0, cv = 0;
    for (size_t j = 1; j < nj; ++j) {
        if (joint_parent[j] < 0 || joint_parent[j] >= static_cast<int32_t>(j))
            throw std::runtime_error("joint parents must precede their children");
        const int t = joint_type[j];
        if (t != IKGPU_JOINT_REVOLUTE && t != IKGPU_JOINT_PRISMATIC && t != IKGPU_JOINT_FREEFLYER && t != IKGPU_JOINT_REVOLUTE_UNBOUNDED)
            throw std::runtime_error("unknown joint type for joint " + joint_names[j]);
        if (joint_idx_q[j] != cq || joint_idx_v[j] != cv) throw std::runtime_error("idx_q / idx_v must be dense and ordered");
        cq += (t == IKGPU_JOINT_FREEFLYER) ? 7 : (t == IKGPU_JOINT_REVOLUTE_UNBOUNDED ? 2 : 1);
        cv += (t == IKGPU_JOINT_FREEFLYER) ? 6 : 1;
    }
    if (cq != nq || cv != nv) throw std::runtime_error("nq / nv do not match the joint list");
    for (size_t f = 0; f < nf; ++f)
        if (frame_parent[f] < 0 || frame_parent[f] >= static_cast<int32_t>(nj))
            throw std::runtime_error("frame parent joint out of range: " + frame_names[f]);
    joint_name_ptrs.clear();
    frame_name_ptrs.clear();
    for (const auto &s : joint_names) joint_name_ptrs.push_back(s.c_str());
    for (const auto &s : frame_names) frame_name_ptrs.push_back(s.c_str());
}

Model Model::from_flat(const ikgpu_flat_model &f) {
    if (f.njoints < 1 || f.nframes < 0 || !f.joint_type || !f.joint_parent || !f.joint_idx_q || !f.joint_idx_v ||
        !f.joint_placement || !f.joint_axis || (f.nq > 0 && (!f.lower || !f.upper)) ||
        (f.nframes > 0 && (!f.frame_parent || !f.frame_placement)))
        throw std::runtime_error("null array in ikgpu_flat_model");
    Model m;
    m.nq = f.nq;
    m.nv = f.nv;
    for (int j = 0; j < f.njoints; ++j) {
        m.joint_type.push_back(f.joint_type[j]);
        m.joint_parent.push_back(f.joint_parent[j]);
        m.joint_idx_q.push_back(f.joint_idx_q[j]);
        m.joint_idx_v.push_back(f.joint_idx_v[j]);
        SE3 p;
        std::memcpy(p.data(), f.joint_placement + 12 * j, sizeof(double) * 12);
        m.joint_placement.push_back(p);
        m.joint_axis.push_back({f.joint_axis[3 * j], f.joint_axis[3 * j + 1], f.joint_axis[3 * j + 2]});
        m.joint_names.push_back(f.joint_names && f.joint_names[j] ? f.joint_names[j] : ("joint" + std::to_string(j)));
        m.joint_mass.push_back(f.joint_mass ? f.joint_mass[j] : 0.0);
        if (f.joint_mass && f.joint_com) m.joint_com.push_back({f.joint_com[3 * j], f.joint_com[3 * j + 1], f.joint_com[3 * j + 2]});
        else m.joint_com.push_back({0.0, 0.0, 0.0});
    }
    m.lower.assign(f.lower, f.lower + f.nq);
    m.upper.assign(f.upper, f.upper + f.nq);
    for (int i = 0; i < f.nframes; ++i) {
        m.frame_parent.push_back(f.frame_parent[i]);
        SE3 p;
        std::memcpy(p.data(), f.frame_placement + 12 * i, sizeof(double) * 12);
        m.frame_placement.push_back(p);
        m.frame_names.push_back(f.frame_names && f.frame_names[i] ? f.frame_names[i] : ("frame" + std::to_string(i)));
    }
    m.finalize();
    return m;
}

Model Model::from_urdf(const char *xml_text, size_t len, bool free_flyer) {
    xml::Parser parser(xml_text, len);
    auto root = parser.parse_document();
    if (root->tag != "robot") throw std::runtime_error("URDF root element must be <robot>, got <" + root->tag + ">");

    std::vector<std::string> links;
    struct LinkInertial { double mass = 0.0; std::array<double, 3> com{0, 0, 0}; };
    std::map<std::string, LinkInertial> inertials;
    std::map<std::string, UrdfJoint> joints;  // ordered by name, as urdfdom's joints_ map
    for (const auto &el : root->children) {
        if (el->tag == "link") {
            const std::string *n = el->attr("name");
            if (!n) throw std::runtime_error("<link> without a name");
            links.push_back(*n);
            if (const xml::Element *in = el->child("inertial")) {  // mass and centre of mass; the rotational inertia is not on the path
                const xml::Element *ms = in->child("mass"), *org = in->child("origin");
                if (ms && ms->attr("value")) {
                    LinkInertial li;
                    li.mass = parse_doubles(*ms->attr("value"), "mass of " + *n).at(0);
                    li.com = parse_vec3(org ? org->attr("xyz") : nullptr, "inertial origin of " + *n, {0, 0, 0});
                    inertials[*n] = li;
                }
            }
        } else if (el->tag == "joint") {
            UrdfJoint j;
            const std::string *n = el->attr("name"), *t = el->attr("type");
            if (!n || !t) throw std::runtime_error("<joint> needs name and type");
            j.name = *n;
            j.type = *t;
            const xml::Element *par = el->child("parent"), *chi = el->child("child");
            if (!par || !chi || !par->attr("link") || !chi->attr("link"))
                throw std::runtime_error("joint " + j.name + " needs <parent link> and <child link>");
            j.parent = *par->attr("link");
            j.child = *chi->attr("link");
            const xml::Element *org = el->child("origin");
            auto xyz = parse_vec3(org ? org->attr("xyz") : nullptr, "origin xyz of " + j.name, {0, 0, 0});
            auto rpy = parse_vec3(org ? org->attr("rpy") : nullptr, "origin rpy of " + j.name, {0, 0, 0});
            rpy_to_rotation(rpy, j.origin.data());
            j.origin[9] = xyz[0]; j.origin[10] = xyz[1]; j.origin[11] = xyz[2];
            const xml::Element *ax = el->child("axis");
            j.axis = parse_vec3(ax ? ax->attr("xyz") : nullptr, "axis of " + j.name, {1, 0, 0});
            if (const xml::Element *lim = el->child("limit")) {
                if (const std::string *lo = lim->attr("lower")) j.lower = parse_doubles(*lo, "limit lower of " + j.name).at(0);
                if (const std::string *hi = lim->attr("upper")) j.upper = parse_doubles(*hi, "limit upper of " + j.name).at(0);
            }
            if (!joints.emplace(j.name, j).second) throw std::runtime_error("duplicate joint name " + j.name);
        }
    }
    if (links.empty()) throw std::runtime_error("URDF has no links");

    std::map<std::string, std::vector<const UrdfJoint *>> children;
    std::map<std::string, bool> has_parent;
    for (const auto &l : links) { children[l]; has_parent[l] = false; }
    for (const auto &kv : joints) {
        const UrdfJoint &j = kv.second;
        if (!children.count(j.parent) || !children.count(j.child))
            throw std::runtime_error("joint " + j.name + " refers to an unknown link");
        if (has_parent[j.child]) throw std::runtime_error("link " + j.child + " has two parent joints");
        children[j.parent].push_back(&j);
        has_parent[j.child] = true;
    }
    std::string root_link;
    for (const auto &l : links)
        if (!has_parent[l]) {
            if (!root_link.empty()) throw std::runtime_error("URDF has more than one root link (" + root_link + ", " + l + ")");
            root_link = l;
        }
    if (root_link.empty()) throw std::runtime_error("URDF has no root link");

    Model m;
    auto add_joint = [&](int32_t parent, int32_t type, const SE3 &placement, const std::string &name,
                         std::array<double, 3> axis, double lo, double hi) {
        m.joint_names.push_back(name);
        m.joint_type.push_back(type);
        m.joint_parent.push_back(parent);
        m.joint_placement.push_back(placement);
        m.joint_axis.push_back(axis);
        m.joint_idx_q.push_back(m.nq);
        m.joint_idx_v.push_back(m.nv);
        if (type == IKGPU_JOINT_FREEFLYER) {
            m.nq += 7; m.nv += 6;
            for (int i = 0; i < 7; ++i) {
                m.lower.push_back(-std::numeric_limits<double>::max());
                m.upper.push_back(std::numeric_limits<double>::max());
            }
        } else if (type == IKGPU_JOINT_REVOLUTE_UNBOUNDED) {   // (cos, sin): Pinocchio's limits for a continuous joint
            m.nq += 2; m.nv += 1;
            for (int i = 0; i < 2; ++i) { m.lower.push_back(-1.01); m.upper.push_back(1.01); }
        } else if (type != IKGPU_JOINT_UNIVERSE) {
            m.nq += 1; m.nv += 1;
            m.lower.push_back(lo);
            m.upper.push_back(hi);
        }
        return static_cast<int32_t>(m.joint_type.size()) - 1;
    };
    auto add_frame = [&](const std::string &name, int32_t parent, const SE3 &pl) {
        m.frame_names.push_back(name);
        m.frame_parent.push_back(parent);
        m.frame_placement.push_back(pl);
        return static_cast<int32_t>(m.frame_parent.size()) - 1;
    };

    add_joint(0, IKGPU_JOINT_UNIVERSE, se3_identity(), "universe", {0, 0, 0}, 0, 0);
    add_frame("universe", 0, se3_identity());
    std::map<std::string, int32_t> body_frame;  // link name -> its BODY frame
    if (free_flyer) {
        int32_t jid = add_joint(0, IKGPU_JOINT_FREEFLYER, se3_identity(), "root_joint", {0, 0, 0}, 0, 0);
        add_frame("root_joint", jid, se3_identity());
        body_frame[root_link] = add_frame(root_link, jid, se3_identity());
    } else {
        add_frame("root_joint", 0, se3_identity());
        body_frame[root_link] = add_frame(root_link, 0, se3_identity());
    }

    // depth-first, explicit stack replaced by recursion through a lambda
    struct Visitor {
        Model &m;
        std::map<std::string, std::vector<const UrdfJoint *>> &children;
        std::map<std::string, int32_t> &body_frame;
        decltype(add_joint) &add_joint_;
        decltype(add_frame) &add_frame_;
        void visit(const std::string &link, int depth) {
            if (depth > 4096) throw std::runtime_error("URDF tree too deep (cycle?)");
            const int32_t pf = body_frame.at(link);
            for (const UrdfJoint *j : children.at(link)) {
                const int32_t pj = m.frame_parent[pf];
                const SE3 pl = se3_mul(m.frame_placement[pf], j->origin);
                if (j->type == "fixed") {
                    add_frame_(j->name, pj, pl);
                    body_frame[j->child] = add_frame_(j->child, pj, pl);
                } else if (j->type == "revolute" || j->type == "prismatic" || j->type == "continuous") {
                    std::array<double, 3> a = j->axis;
                    const bool aligned = (a == std::array<double, 3>{1, 0, 0}) || (a == std::array<double, 3>{0, 1, 0}) ||
                                         (a == std::array<double, 3>{0, 0, 1});
                    if (!aligned) {
                        const double n = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
                        if (!(n > 0.0)) throw std::runtime_error("zero axis on joint " + j->name);
                        a = {a[0] / n, a[1] / n, a[2] / n};
                    }
                    // "continuous": Pinocchio's JointModelRevoluteUnbounded*, configuration (cos, sin), see include/ikgpu.h
                    const int32_t type = (j->type == "revolute") ? IKGPU_JOINT_REVOLUTE
                                       : (j->type == "continuous") ? IKGPU_JOINT_REVOLUTE_UNBOUNDED : IKGPU_JOINT_PRISMATIC;
                    const int32_t jid = add_joint_(pj, type, pl, j->name, a, j->lower, j->upper);
                    add_frame_(j->name, jid, se3_identity());
                    body_frame[j->child] = add_frame_(j->child, jid, se3_identity());
                } else {
                    // floating / planar joints are unused by the path
                    throw std::runtime_error("unsupported URDF joint type '" + j->type + "' on joint " + j->name);
                }
                visit(j->child, depth + 1);
            }
        }
    } visitor{m, children, body_frame, add_joint, add_frame};
    visitor.visit(root_link, 0);

    // appendBodyToJoint for every link that carries an <inertial>, in visiting order (= BODY frame order)
    std::vector<std::pair<int32_t, const std::string *>> bodies;
    for (const auto &kv : body_frame) bodies.push_back({kv.second, &kv.first});
    std::sort(bodies.begin(), bodies.end());
    for (const auto &b : bodies) {
        auto it = inertials.find(*b.second);
        if (it != inertials.end()) m.append_body(m.frame_parent[b.first], m.frame_placement[b.first], it->second.mass, it->second.com);
    }

    m.finalize();
    return m;
}

}  // namespace ikgpu
