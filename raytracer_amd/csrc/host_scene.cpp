// Host front end: .nff dialect -> primitives -> flattened bounding-sphere octree.
//
// Behavioural contract (file:line into the reference):
//   dialect and directive order ........ Driver.cpp:426-691
//   ear clipping of `p` / `pp` ......... Polygon.h:107-205, PolygonalPatch.h:78-157
//   triangle edge/normal precompute .... Triangle.h:40-46
//   octree geometry and membership ..... Octree.h:780-907, 1298-1320; Triangle.h:359-374; Sphere.h:211-232
//   root cube .......................... Driver.cpp:231,242
//   camera frame ....................... Camera.h:64-97
//   PPM output ......................... Driver.cpp:712-733
//   argv scanner ....................... Driver.cpp:167-216, 771-783
//
// Arithmetic that decides results (edge vectors, normals, octant bounds, camera
// frame) is written as single IEEE operations in the reference's evaluation order
// and this file is compiled without FMA contraction.
#include "host_scene.hpp"

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace rth {

// ---- small vector helpers (evaluation order: ((x0*y0 + x1*y1) + x2*y2)) ----
static inline Vec3 mk(double x, double y, double z) { Vec3 r; r.x = x; r.y = y; r.z = z; return r; }
static inline Vec3 operator-(const Vec3 &a, const Vec3 &b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline double dot(const Vec3 &a, const Vec3 &b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline Vec3 cross(const Vec3 &a, const Vec3 &b)
{
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline Vec3 unit(const Vec3 &a)
{
    double z = dot(a, a);
    if (z > 0) { double s = std::sqrt(z); return mk(a.x / s, a.y / s, a.z / s); }
    return a;
}
static inline double length(const Vec3 &a) { return std::sqrt(dot(a, a)); }

// ---- line cursor with `istream >> x` semantics on one text line ----
class Cursor {
public:
    explicit Cursor(const std::string &s) : p_(s.c_str()) {}
    void word()
    {
        blanks();
        while (*p_ && !std::isspace(static_cast<unsigned char>(*p_))) ++p_;
    }
    bool real(double &out)
    {
        blanks();
        if (!*p_) return false;
        char *end = nullptr;
        double v = std::strtod(p_, &end);
        if (end == p_) return false;
        out = v;
        p_ = end;
        return true;
    }
    bool integer(int &out)
    {
        blanks();
        if (!*p_) return false;
        char *end = nullptr;
        long v = std::strtol(p_, &end, 10);
        if (end == p_) return false;
        out = static_cast<int>(v);
        p_ = end;
        return true;
    }
    bool vec(Vec3 &v) { return real(v.x) && real(v.y) && real(v.z); }
    bool at_end() const { return *p_ == 0; }

private:
    void blanks()
    {
        while (*p_ && std::isspace(static_cast<unsigned char>(*p_))) ++p_;
    }
    const char *p_;
};

// ---- triangulation ----
namespace {

struct Ring {
    std::vector<Vec3> pos;
    std::vector<Vec3> nrm;  // empty for plain polygons
};

// 3x3 determinant with columns a, b, c, expanded the way Eigen 3.4 expands it.
double det_columns(const Vec3 &a, const Vec3 &b, const Vec3 &c)
{
    double m0 = a.x * (b.y * c.z - c.y * b.z);
    double m1 = b.x * (a.y * c.z - c.y * a.z);
    double m2 = c.x * (a.y * b.z - b.y * a.z);
    return (m0 - m1) + m2;
}

// Same-side test on the z component only (Polygon.h:97-105).
bool strictly_inside_xy(const Vec3 &a, const Vec3 &b, const Vec3 &c, const Vec3 &q)
{
    double s1 = cross(a - b, a - q).z;
    double s2 = cross(b - c, b - q).z;
    double s3 = cross(c - a, c - q).z;
    return (s1 > 0 && s2 > 0 && s3 > 0) || (s1 < 0 && s2 < 0 && s3 < 0);
}

void emit_triangle(std::vector<Primitive> &out, int material, const Ring &r, int ia, int ib, int ic)
{
    Primitive t;
    t.material = material;
    t.v[0] = r.pos[ia]; t.v[1] = r.pos[ib]; t.v[2] = r.pos[ic];
    if (!r.nrm.empty()) {
        t.has_vnormals = true;
        t.vn[0] = r.nrm[ia]; t.vn[1] = r.nrm[ib]; t.vn[2] = r.nrm[ic];
    }
    t.ab = t.v[0] - t.v[1];
    t.ac = t.v[0] - t.v[2];
    t.n = unit(cross(t.ab, t.ac));
    out.push_back(t);
}

// Restartable ear clipping: scan corners from index 0, cut the first convex corner
// that contains no other vertex (xy test), restart; the last three vertices close the fan.
void clip_ears(std::vector<Primitive> &out, int material, Ring ring)
{
    const Vec3 plane_n = unit(cross(ring.pos[0] - ring.pos[1], ring.pos[2] - ring.pos[1]));
    const double deg = 180.0 / 3.14159;
    int i = 0;
    while (static_cast<int>(ring.pos.size()) > 3 && i < static_cast<int>(ring.pos.size()) - 3) {
        const int n = static_cast<int>(ring.pos.size());
        const int prev = (i - 1 + n) % n, next = (i + 1) % n;
        const Vec3 &a = ring.pos[prev], &b = ring.pos[i], &c = ring.pos[next];
        const Vec3 ba = b - a, bc = b - c;
        double angle = std::atan2(det_columns(ba, bc, plane_n), dot(ba, bc));
        angle *= deg;
        bool cut = !(angle > 180 || angle < 0);
        if (cut) {
            for (int k = 0; k < n; ++k) {
                if (k == prev || k == i || k == next) continue;
                if (strictly_inside_xy(a, b, c, ring.pos[k])) { cut = false; break; }
            }
        }
        if (!cut) { ++i; continue; }
        emit_triangle(out, material, ring, prev, i, next);
        ring.pos.erase(ring.pos.begin() + i);
        if (!ring.nrm.empty()) ring.nrm.erase(ring.nrm.begin() + i);
        i = 0;
    }
    emit_triangle(out, material, ring, 0, 1, 2);
}

}  // namespace

bool Scene::load(const std::string &path, std::string &err)
{
    std::ifstream in(path, std::ios::binary);
    if (!in.is_open()) { err = "error opening file: " + path; return false; }
    std::vector<std::string> lines;
    for (std::string ln; std::getline(in, ln);) lines.push_back(ln);

    // The current material persists from one `f` line to the next one.
    int material = -1;
    auto current_material = [&]() {
        if (material < 0) {
            rt_material zero;
            std::memset(&zero, 0, sizeof zero);
            materials.push_back(zero);
            material = static_cast<int>(materials.size()) - 1;
        }
        return material;
    };
    const std::string blank;
    size_t li = 0;
    auto take = [&]() -> const std::string & { return (li + 1 < lines.size()) ? lines[++li] : (li = lines.size(), blank); };

    for (li = 0; li < lines.size(); ++li) {
        const std::string &line = lines[li];
        Cursor cur(line);
        const char kind = line.empty() ? '\0' : line[0];
        const bool patch = line.size() > 1 && line[0] == 'p' && line[1] == 'p';
        if (patch || kind == 'p') {
            int count = 0;
            cur.word();
            cur.integer(count);
            // a vertex count that runs past the end of the file is a truncated or corrupt file (the
            // reference would re-read its last line that many times)
            if (count > 0 && static_cast<size_t>(count) > lines.size() - li - 1) {
                err = "polygon at line " + std::to_string(li + 1) + " announces " + std::to_string(count) +
                      " vertices, the file has fewer lines left: " + path;
                return false;
            }
            Ring ring;
            Vec3 p, n;  // a short line keeps the previous values, like a failed stream extraction
            for (int k = 0; k < count; ++k) {
                Cursor vc(take());
                if (patch) { vc.vec(p) && vc.vec(n); }
                else vc.vec(p);
                ring.pos.push_back(p);
                if (patch) ring.nrm.push_back(unit(n));
            }
            if (count >= 3) clip_ears(prims, current_material(), ring);
            continue;
        }
        switch (kind) {
        case 'b': cur.word(); cur.vec(background); break;
        case 'Q': cur.word(); cur.integer(q_directive); break;
        case 'Z': cur.word(); cur.integer(lightcam_samples); break;
        case 'v': {
            { Cursor c(take()); c.word(); c.vec(from); }
            { Cursor c(take()); c.word(); c.vec(at); }
            { Cursor c(take()); c.word(); c.vec(up); }
            { Cursor c(take()); c.word(); c.real(angle); }
            { Cursor c(take()); c.word(); c.real(hither); }
            { Cursor c(take()); c.word(); c.integer(resx) && c.integer(resy); }
            break;
        }
        case 'l': {
            rt_light lt;
            std::memset(&lt, 0, sizeof lt);
            lt.color[0] = lt.color[1] = lt.color[2] = 1.0;
            lt.intensity = 1.0;
            Vec3 pos;
            cur.word();
            cur.vec(pos);
            lt.pos[0] = pos.x; lt.pos[1] = pos.y; lt.pos[2] = pos.z;
            if (!cur.at_end()) {  // optional colour: one token is skipped first
                Vec3 c;
                cur.word();
                if (cur.vec(c)) { lt.color[0] = c.x; lt.color[1] = c.y; lt.color[2] = c.z; }
            }
            lights.push_back(lt);
            break;
        }
        case 'T': {
            Ring ring;
            Vec3 p;
            for (int k = 0; k < 3; ++k) { Cursor vc(take()); vc.vec(p); ring.pos.push_back(p); }
            emit_triangle(prims, current_material(), ring, 0, 1, 2);
            break;
        }
        case 'f': {
            double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            cur.word();
            for (double &x : a) if (!cur.real(x)) break;
            rt_material m = {a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]};
            materials.push_back(m);
            material = static_cast<int>(materials.size()) - 1;
            break;
        }
        case 's': {
            Primitive s;
            s.sphere = true;
            s.material = current_material();
            cur.word();
            cur.vec(s.c) && cur.real(s.r);
            prims.push_back(s);
            break;
        }
        default: break;
        }
    }
    for (rt_light &l : lights) l.intensity = 1.0 / std::sqrt(static_cast<double>(lights.size()));
    build_octree();
    return true;
}

// ---- octree ----
namespace {

struct Box { Vec3 lo, hi; };

bool triangle_within(const Primitive &t, const Box &b)
{
    for (int i = 0; i < 3; ++i)
        for (int d = 0; d < 3; ++d)
            if (!(b.lo[d] <= t.v[i][d] && t.v[i][d] <= b.hi[d])) return false;
    return true;
}

// Arvo's sphere/box overlap; touching (distance^2 == r^2) counts as outside.
bool sphere_overlaps(const Primitive &s, const Box &b)
{
    double dist2 = 0;
    for (int d = 0; d < 3; ++d) {
        const double p = s.c[d];
        if (b.lo[d] > p) dist2 += (p - b.lo[d]) * (p - b.lo[d]);
        else if (b.hi[d] < p) dist2 += (p - b.hi[d]) * (p - b.hi[d]);
    }
    return !(dist2 >= s.r * s.r);
}

struct Builder {
    Scene &sc;
    explicit Builder(Scene &s) : sc(s) {}

    void node(const Vec3 &corner, double size, const std::vector<int> &members, int depth)
    {
        const int me = static_cast<int>(sc.nodes.size());
        sc.nodes.emplace_back();
        rt_node nd;
        std::memset(&nd, 0, sizeof nd);
        const Vec3 mid = mk(corner.x + size / 2, corner.y + size / 2, corner.z + size / 2);
        nd.center[0] = mid.x; nd.center[1] = mid.y; nd.center[2] = mid.z;
        nd.radius = (size * std::sqrt(3)) / 2.0;
        nd.depth = depth;

        // child cubes: x toggles fastest, then y, then z
        Box kid[8];
        const double h = size / 2;
        for (int k = 0; k < 8; ++k) {
            kid[k].lo = mk((k & 1) ? mid.x : mid.x - size / 2, (k & 2) ? mid.y : mid.y - size / 2,
                           (k & 4) ? mid.z : mid.z - size / 2);
            kid[k].hi = mk(kid[k].lo.x + h, kid[k].lo.y + h, kid[k].lo.z + h);
        }
        std::vector<int> below[8];
        std::vector<char> pushed(members.size(), 0);
        if (size / 2 > 0.01 && members.size() > 5) {
            for (size_t m = 0; m < members.size(); ++m) {
                const Primitive &p = sc.prims[members[m]];
                for (int k = 0; k < 8; ++k) {
                    const bool in = p.sphere ? sphere_overlaps(p, kid[k]) : triangle_within(p, kid[k]);
                    if (in) { below[k].push_back(members[m]); pushed[m] = 1; }
                }
            }
        }
        // residents: triangles first, then spheres, each in ascending primitive id
        nd.tri_begin = static_cast<int32_t>(sc.tri.size());
        for (size_t m = 0; m < members.size(); ++m) {
            if (pushed[m] || sc.prims[members[m]].sphere) continue;
            const Primitive &p = sc.prims[members[m]];
            rt_tri_rec r;
            std::memset(&r, 0, sizeof r);
            for (int d = 0; d < 3; ++d) { r.v0[d] = p.v[0][d]; r.ab[d] = p.ab[d]; r.ac[d] = p.ac[d]; }
            r.prim = members[m];
            sc.tri.push_back(r);
        }
        nd.tri_count = static_cast<int32_t>(sc.tri.size()) - nd.tri_begin;
        nd.sph_begin = static_cast<int32_t>(sc.sph.size());
        for (size_t m = 0; m < members.size(); ++m) {
            if (pushed[m] || !sc.prims[members[m]].sphere) continue;
            const Primitive &p = sc.prims[members[m]];
            rt_sph_rec r = {{p.c.x, p.c.y, p.c.z}, p.r};
            sc.sph.push_back(r);
            uint32_t tag = static_cast<uint32_t>(members[m]);
            if (sc.materials[p.material].T > 0) tag |= RT_SPH_TRANSPARENT;
            sc.sph_prim.push_back(tag);
        }
        nd.sph_count = static_cast<int32_t>(sc.sph.size()) - nd.sph_begin;
        sc.nodes[me] = nd;
        for (int k = 0; k < 8; ++k)
            if (!below[k].empty()) node(kid[k].lo, kid[k].hi.x - kid[k].lo.x, below[k], depth + 1);
        sc.nodes[me].skip = static_cast<int32_t>(sc.nodes.size());
    }
};

}  // namespace

void Scene::build_octree()
{
    nodes.clear(); tri.clear(); sph.clear(); sph_prim.clear();
    prim_flags.assign(prims.size(), 0);
    prim_material.assign(prims.size(), 0);
    prim_normals.assign(prims.size() * 12, 0.0);
    std::vector<int> all(prims.size());
    for (size_t i = 0; i < prims.size(); ++i) {
        all[i] = static_cast<int>(i);
        const Primitive &p = prims[i];
        prim_flags[i] = (p.sphere ? RT_PRIM_SPHERE : 0u) | (p.has_vnormals ? RT_PRIM_HAS_VNORMALS : 0u);
        prim_material[i] = p.material;
        if (!p.sphere) {
            double *o = &prim_normals[i * 12];
            o[0] = p.n.x; o[1] = p.n.y; o[2] = p.n.z;
            for (int k = 0; k < 3; ++k) { o[3 + k * 3] = p.vn[k].x; o[4 + k * 3] = p.vn[k].y; o[5 + k * 3] = p.vn[k].z; }
        }
    }
    const double size = 100;  // root cube: corner (-50,-50,-50), side 100
    Builder b(*this);
    b.node(mk(0 - size / 2, 0 - size / 2, 0 - size / 2), size, all, 0);
}

void Scene::fill_desc(rt_scene_desc &d) const
{
    std::memset(&d, 0, sizeof d);
    d.nodes = nodes.data(); d.n_nodes = static_cast<int32_t>(nodes.size());
    d.tri = tri.data(); d.n_tri_slots = static_cast<int32_t>(tri.size());
    d.sph = sph.data(); d.n_sph_slots = static_cast<int32_t>(sph.size());
    d.sph_prim = sph_prim.data();
    d.n_prims = static_cast<int32_t>(prims.size());
    d.prim_flags = prim_flags.data();
    d.prim_material = prim_material.data();
    d.prim_normals = prim_normals.data();
    d.materials = materials.data(); d.n_materials = static_cast<int32_t>(materials.size());
    d.lights = lights.data(); d.n_lights = static_cast<int32_t>(lights.size());
    d.background[0] = background.x; d.background[1] = background.y; d.background[2] = background.z;
    d.lightcam_samples = lightcam_samples;
}

void Scene::camera(int rx, int ry, rt_camera &cam) const
{
    if (rx <= 0) rx = resx;
    if (ry <= 0) ry = resy;
    std::memset(&cam, 0, sizeof cam);
    const Vec3 w = unit(from - at);
    const Vec3 u = unit(cross(up, w));
    const Vec3 v = unit(cross(w, u));
    const double aspect = static_cast<double>(ry != 0 ? rx / ry : 0);  // integer division on purpose
    const double d = length(from - at);
    const double half = d * std::tan(angle / 2 * 3.14159 / 180.0);
    const double pw = 2 * half / rx;
    cam.u[0] = u.x; cam.u[1] = u.y; cam.u[2] = u.z;
    cam.v[0] = v.x; cam.v[1] = v.y; cam.v[2] = v.z;
    cam.w[0] = w.x; cam.w[1] = w.y; cam.w[2] = w.z;
    cam.e[0] = from.x; cam.e[1] = from.y; cam.e[2] = from.z;
    cam.d = d;
    cam.pix_width = pw;
    cam.l = -half + 0.5 * pw;
    cam.t = half * aspect - 0.5 * pw;
    cam.resx = rx;
    cam.resy = ry;
}

}  // namespace rth

// ------------------------------------------------------------------ C ABI
extern "C" void rt_set_error_string(const char *msg);  // defined with the device path

static int host_fail(int code, const std::string &msg)
{
    rt_set_error_string(msg.c_str());
    return code;
}

struct rth_scene {
    rth::Scene sc;
};

extern "C" int rth_scene_load(const char *nff_path, rth_scene **out)
{
    if (!nff_path || !out) return host_fail(RT_ERR_INVALID, "rth_scene_load: null argument");
    rth_scene *s = new rth_scene();
    std::string err;
    if (!s->sc.load(nff_path, err)) {
        delete s;
        return host_fail(RT_ERR_IO, err);
    }
    *out = s;
    return RT_OK;
}

extern "C" int rth_scene_free(rth_scene *s)
{
    delete s;
    return RT_OK;
}

extern "C" int rth_scene_desc(const rth_scene *s, rt_scene_desc *out)
{
    if (!s || !out) return host_fail(RT_ERR_INVALID, "rth_scene_desc: null argument");
    s->sc.fill_desc(*out);
    return RT_OK;
}

extern "C" int rth_scene_camera(const rth_scene *s, int32_t resx, int32_t resy, rt_camera *out)
{
    if (!s || !out) return host_fail(RT_ERR_INVALID, "rth_scene_camera: null argument");
    s->sc.camera(resx, resy, *out);
    if (out->resx <= 0 || out->resy <= 0) return host_fail(RT_ERR_INVALID, "rth_scene_camera: scene has no resolution");
    return RT_OK;
}

extern "C" int rth_scene_resolution(const rth_scene *s, int32_t *resx, int32_t *resy)
{
    if (!s || !resx || !resy) return host_fail(RT_ERR_INVALID, "rth_scene_resolution: null argument");
    *resx = s->sc.resx;
    *resy = s->sc.resy;
    return RT_OK;
}

extern "C" int rth_scene_prim_geometry(const rth_scene *s, int32_t prim, double *out9)
{
    if (!s || !out9 || prim < 0 || prim >= static_cast<int32_t>(s->sc.prims.size()))
        return host_fail(RT_ERR_INVALID, "rth_scene_prim_geometry: bad argument");
    const rth::Primitive &p = s->sc.prims[prim];
    std::memset(out9, 0, sizeof(double) * 9);
    if (p.sphere) { out9[0] = p.c.x; out9[1] = p.c.y; out9[2] = p.c.z; out9[3] = p.r; }
    else for (int k = 0; k < 3; ++k) { out9[k * 3] = p.v[k].x; out9[k * 3 + 1] = p.v[k].y; out9[k * 3 + 2] = p.v[k].z; }
    return RT_OK;
}

static int write_ppm_bytes(const char *path, const uint8_t *b, int32_t resx, int32_t resy)
{
    FILE *f = std::fopen(path, "wb");
    if (!f) return host_fail(RT_ERR_IO, std::string("cannot write ") + path);
    std::fprintf(f, "P6\n%d %d\n%d\n", resx, resy, 255);
    std::fwrite(b, 1, static_cast<size_t>(resx) * resy * 3, f);
    std::fclose(f);
    return RT_OK;
}

extern "C" int rth_write_ppm_f64(const char *path, const double *rgb, int32_t resx, int32_t resy)
{
    if (!path || !rgb || resx <= 0 || resy <= 0) return host_fail(RT_ERR_INVALID, "rth_write_ppm_f64: bad argument");
    std::vector<uint8_t> b(static_cast<size_t>(resx) * resy * 3);
    for (size_t i = 0; i < b.size(); ++i) b[i] = static_cast<uint8_t>(static_cast<int>(rgb[i]));
    return write_ppm_bytes(path, b.data(), resx, resy);
}

extern "C" int rth_write_ppm_u8(const char *path, const uint8_t *rgb, int32_t resx, int32_t resy)
{
    if (!path || !rgb || resx <= 0 || resy <= 0) return host_fail(RT_ERR_INVALID, "rth_write_ppm_u8: bad argument");
    return write_ppm_bytes(path, rgb, resx, resy);
}

// The reference dispatches on the first character of every argument (after any
// leading '-'), so a positional that begins with s / a / p is taken as that flag;
// the input is argv[argc-2] (".nff" appended when missing), the output argv[argc-1].
extern "C" int rth_parse_args(int argc, const char *const *argv, int32_t *num_samples, double *aperture, int32_t *phong,
                              char *input_path, size_t input_cap, char *output_path, size_t output_cap)
{
    if (argc < 3 || !argv || !num_samples || !aperture || !phong || !input_path || !output_path)
        return host_fail(RT_ERR_INVALID, "usage: rt_driver [-s N] [-a A] [-p] input.nff output.ppm");
    *num_samples = 1; *aperture = 0; *phong = 0;
    int arg = 1;
    size_t pos = 0;
    while (arg < argc) {
        const char c = argv[arg][pos];
        if (c == '-') { ++pos; continue; }
        if (c == 's' || c == 'a') {
            const char *val = (arg + 1 < argc) ? argv[arg + 1] : "0";
            if (c == 's') *num_samples = std::atoi(val); else *aperture = std::atof(val);
            arg += 2;
        } else {
            if (c == 'p') *phong = 1;
            arg += 1;
        }
        pos = 0;
    }
    std::string in = argv[argc - 2];
    if (in.size() < 4 || in.compare(in.size() - 4, 4, ".nff") != 0) in += ".nff";
    if (in.size() + 1 > input_cap || std::strlen(argv[argc - 1]) + 1 > output_cap)
        return host_fail(RT_ERR_INVALID, "rth_parse_args: path buffer too small");
    std::strcpy(input_path, in.c_str());
    std::strcpy(output_path, argv[argc - 1]);
    return RT_OK;
}
