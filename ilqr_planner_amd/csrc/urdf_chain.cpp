// urdf_chain.cpp -- URDF text -> kinematic chain of an ilqr_problem_desc.
//
// Replaces what sim::KDLRobot's constructor obtains from TinyURDFParser + orocos_kdl
// (reference src/sim/KDLRobot.cpp:45-66): the chain of joints from `base_frame` to `tip_frame`, each joint
// contributing one segment  T_seg(q) = Trans(origin xyz) * RPY(origin rpy) * Rot(axis, q)  (fixed joints: no Rot),
// followed by the user tool frame  Frame(EulerZYX(rpy[0], rpy[1], rpy[2]), xyz)  as a last fixed segment.
// tinyxml2 is not available offline, so this file carries a minimal XML reader (elements, attributes, comments,
// declarations; no entities beyond the five predefined ones, no CDATA) -- enough for URDF.
#include "../../include/ilqr_hip.h"

#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace {

struct Node {
    std::string name;
    std::map<std::string, std::string> attr;
    std::vector<std::unique_ptr<Node>> kids;
    const Node* child(const char* n) const {
        for (auto& k : kids)
            if (k->name == n) return k.get();
        return nullptr;
    }
    std::string get(const char* a, const char* def = "") const {
        auto it = attr.find(a);
        return it == attr.end() ? std::string(def) : it->second;
    }
};

struct Parser {
    const char* s;
    size_t n, i = 0;
    std::string err;
    explicit Parser(const char* t) : s(t), n(std::strlen(t)) {}
    bool starts(const char* w) const { return std::strncmp(s + i, w, std::strlen(w)) == 0; }
    void ws() {
        while (i < n && std::isspace((unsigned char)s[i])) i++;
    }
    static bool namech(char c) { return std::isalnum((unsigned char)c) || c == '_' || c == ':' || c == '-' || c == '.'; }
    static std::string unescape(const std::string& v) {
        std::string o;
        for (size_t k = 0; k < v.size(); k++) {
            if (v[k] == '&') {
                static const char* ent[][2] = {{"&lt;", "<"}, {"&gt;", ">"}, {"&amp;", "&"}, {"&quot;", "\""}, {"&apos;", "'"}};
                bool hit = false;
                for (auto& e : ent)
                    if (v.compare(k, std::strlen(e[0]), e[0]) == 0) { o += e[1]; k += std::strlen(e[0]) - 1; hit = true; break; }
                if (hit) continue;
            }
            o += v[k];
        }
        return o;
    }
    // skip text, comments, declarations and doctype until the next element start; false at end of input
    bool to_next_tag() {
        while (i < n) {
            if (s[i] != '<') { i++; continue; }
            if (starts("<!--")) {
                const char* e = std::strstr(s + i + 4, "-->");
                if (!e) { err = "unterminated comment"; return false; }
                i = (size_t)(e - s) + 3;
            } else if (starts("<?")) {
                const char* e = std::strstr(s + i + 2, "?>");
                if (!e) { err = "unterminated declaration"; return false; }
                i = (size_t)(e - s) + 2;
            } else if (starts("<!")) {
                const char* e = std::strchr(s + i, '>');
                if (!e) { err = "unterminated <! block"; return false; }
                i = (size_t)(e - s) + 1;
            } else {
                return true;
            }
        }
        return false;
    }
    std::unique_ptr<Node> element() {  // at '<' of an opening tag
        i++;
        auto nd = std::make_unique<Node>();
        size_t b = i;
        while (i < n && namech(s[i])) i++;
        nd->name.assign(s + b, i - b);
        if (nd->name.empty()) { err = "malformed tag"; return nullptr; }
        for (;;) {
            ws();
            if (i >= n) { err = "unexpected end inside <" + nd->name + ">"; return nullptr; }
            if (s[i] == '/') {
                if (i + 1 < n && s[i + 1] == '>') { i += 2; return nd; }
                err = "malformed tag end";
                return nullptr;
            }
            if (s[i] == '>') { i++; break; }
            size_t a = i;
            while (i < n && namech(s[i])) i++;
            std::string key(s + a, i - a);
            ws();
            if (key.empty() || i >= n || s[i] != '=') { err = "malformed attribute in <" + nd->name + ">"; return nullptr; }
            i++;
            ws();
            if (i >= n || (s[i] != '"' && s[i] != '\'')) { err = "attribute value must be quoted"; return nullptr; }
            char qc = s[i++];
            size_t v = i;
            while (i < n && s[i] != qc) i++;
            if (i >= n) { err = "unterminated attribute value"; return nullptr; }
            nd->attr[key] = unescape(std::string(s + v, i - v));
            i++;
        }
        // children until the matching close tag
        for (;;) {
            if (!to_next_tag()) { if (err.empty()) err = "missing </" + nd->name + ">"; return nullptr; }
            if (starts("</")) {
                i += 2;
                size_t c = i;
                while (i < n && namech(s[i])) i++;
                std::string cn(s + c, i - c);
                ws();
                if (i >= n || s[i] != '>' || cn != nd->name) { err = "mismatched </" + cn + "> for <" + nd->name + ">"; return nullptr; }
                i++;
                return nd;
            }
            auto k = element();
            if (!k) return nullptr;
            nd->kids.push_back(std::move(k));
        }
    }
    std::unique_ptr<Node> document() {
        if (!to_next_tag()) { if (err.empty()) err = "no root element"; return nullptr; }
        return element();
    }
};

bool vec3(const std::string& t, double o[3]) {
    const char* p = t.c_str();
    for (int k = 0; k < 3; k++) {
        char* e = nullptr;
        o[k] = std::strtod(p, &e);
        if (e == p) return false;
        p = e;
    }
    return true;
}

// KDL Rotation::RPY(r,p,y) = Rz(y) Ry(p) Rx(r); EulerZYX(a,b,c) = RPY(c,b,a)
void rpy_to_R(double r, double p, double y, double R[9]) {
    const double cr = std::cos(r), sr = std::sin(r), cp = std::cos(p), sp = std::sin(p), cy = std::cos(y), sy = std::sin(y);
    R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
    R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
    R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}

thread_local std::string g_err;

}  // namespace

extern "C" const char* ilqr_urdf_last_error(void) { return g_err.c_str(); }

// Fills desc->{dof,n_seg,seg_*} and, when non-null, lower/upper[dof] with the URDF joint limits.
// tool_rpy/tool_xyz may be NULL (identity tool frame).  Returns 0 on success; message in ilqr_urdf_last_error().
extern "C" int ilqr_chain_from_urdf(const char* urdf_text, const char* base_frame, const char* tip_frame, const double* tool_rpy,
                                    const double* tool_xyz, ilqr_problem_desc* desc, double* lower, double* upper) {
    g_err.clear();
    if (!urdf_text || !base_frame || !tip_frame || !desc) { g_err = "null argument"; return 1; }
    Parser ps(urdf_text);
    auto root = ps.document();
    if (!root) { g_err = "[KDLRobot] URDF parse error: " + ps.err; return 1; }
    if (root->name != "robot") { g_err = "[KDLRobot] URDF root element is <" + root->name + ">, expected <robot>"; return 1; }
    std::map<std::string, const Node*> by_child;
    for (auto& k : root->kids) {
        if (k->name != "joint") continue;
        const Node* ch = k->child("child");
        const Node* pa = k->child("parent");
        if (ch && pa) by_child[ch->get("link")] = k.get();
    }
    std::vector<const Node*> path;
    std::string link = tip_frame;
    const std::string fail_msg = std::string("[KDLRobot] Unable to build kinematic chain from ") + base_frame + " to " + tip_frame;
    while (link != base_frame) {
        auto it = by_child.find(link);
        if (it == by_child.end() || path.size() > 1000) { g_err = fail_msg; return 1; }  // KDLRobot.cpp:49,56
        path.push_back(it->second);
        link = it->second->child("parent")->get("link");
    }
    if ((int)path.size() + 1 > ILQR_MAX_SEG) { g_err = "chain has too many segments"; return 1; }
    int ns = 0, dof = 0;
    for (auto it = path.rbegin(); it != path.rend(); ++it, ++ns) {
        const Node* j = *it;
        double xyz[3] = {0, 0, 0}, rpy[3] = {0, 0, 0}, axis[3] = {1, 0, 0};
        if (const Node* o = j->child("origin")) {
            if (!vec3(o->get("xyz", "0 0 0"), xyz) || !vec3(o->get("rpy", "0 0 0"), rpy)) { g_err = "bad <origin> in joint " + j->get("name"); return 1; }
        }
        if (const Node* a = j->child("axis")) {
            if (!vec3(a->get("xyz", "1 0 0"), axis)) { g_err = "bad <axis> in joint " + j->get("name"); return 1; }
        }
        const std::string type = j->get("type");
        std::memcpy(desc->seg_xyz[ns], xyz, sizeof(xyz));
        rpy_to_R(rpy[0], rpy[1], rpy[2], desc->seg_R[ns]);
        if (type == "revolute" || type == "continuous") {
            const double nrm = std::sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
            if (nrm == 0) { g_err = "zero axis in joint " + j->get("name"); return 1; }
            for (int k = 0; k < 3; k++) desc->seg_axis[ns][k] = axis[k] / nrm;
            desc->seg_joint[ns] = dof;
            const Node* lim = j->child("limit");
            if (lower) lower[dof] = (lim && lim->attr.count("lower")) ? std::atof(lim->get("lower").c_str()) : -INFINITY;
            if (upper) upper[dof] = (lim && lim->attr.count("upper")) ? std::atof(lim->get("upper").c_str()) : INFINITY;
            dof++;
        } else if (type == "fixed") {
            desc->seg_joint[ns] = -1;
            desc->seg_axis[ns][0] = 0; desc->seg_axis[ns][1] = 0; desc->seg_axis[ns][2] = 1;
        } else {
            g_err = "unsupported joint type '" + type + "' in joint " + j->get("name");
            return 1;
        }
    }
    // "robot_custom_tip": Frame(EulerZYX(rpy0, rpy1, rpy2), xyz)   (KDLRobot.cpp:61-66; rpy[0] is the Z angle)
    desc->seg_joint[ns] = -1;
    for (int k = 0; k < 3; k++) desc->seg_xyz[ns][k] = tool_xyz ? tool_xyz[k] : 0.0;
    rpy_to_R(tool_rpy ? tool_rpy[2] : 0.0, tool_rpy ? tool_rpy[1] : 0.0, tool_rpy ? tool_rpy[0] : 0.0, desc->seg_R[ns]);
    desc->seg_axis[ns][0] = 0; desc->seg_axis[ns][1] = 0; desc->seg_axis[ns][2] = 1;
    ns++;
    desc->n_seg = ns;
    desc->dof = dof;
    return 0;
}
