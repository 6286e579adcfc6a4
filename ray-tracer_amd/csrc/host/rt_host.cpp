// rt_host.cpp — host-side scene construction and BVH flattening.
// Build with -ffp-contract=off: the cached polygon fields and bounding shells
// must be the same f64 values the reference's constructors produce.
#include "rt_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

namespace rt {
namespace {

// geometry.rs:6-46 with the reference's evaluation order
inline double dot(const Vec3 &x, const Vec3 &y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; }
inline Vec3 cross(const Vec3 &x, const Vec3 &y) {
    return {x[1] * y[2] - x[2] * y[1], x[2] * y[0] - x[0] * y[2], x[0] * y[1] - x[1] * y[0]};
}
inline Vec3 add(const Vec3 &x, const Vec3 &y) { return {x[0] + y[0], x[1] + y[1], x[2] + y[2]}; }
inline Vec3 sub(const Vec3 &x, const Vec3 &y) { return {x[0] - y[0], x[1] - y[1], x[2] - y[2]}; }
inline double mag(const Vec3 &x) { return std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]); }
inline double dist(const Vec3 &x, const Vec3 &y) { return mag(sub(x, y)); }
inline Vec3 scale(double c, const Vec3 &x) { return {c * x[0], c * x[1], c * x[2]}; }
inline Vec3 normalize(const Vec3 &x) {
    double m = mag(x);
    if (m == 0.0) throw std::invalid_argument("normalize: zero vector (the reference panics here)");
    return scale(1.0 / m, x);
}
constexpr double EPSILON = 0.00000001; // geometry.rs:3

} // namespace

// ---------------------------------------------------------------------------
// Polygon
// ---------------------------------------------------------------------------
Polygon Polygon::new_cached(const std::vector<Vec3> &points, const Brdf &brdf, const Vec3 &direc_vec) {
    if (points.size() < 3) throw std::invalid_argument("Polygon::new_cached needs at least 3 points");
    Polygon p;
    Vec3 normal = cross(sub(points[1], points[0]), sub(points[2], points[1]));
    if (dot(normal, direc_vec) < 0.0) normal = scale(-1.0, normal);
    normal = normalize(normal);

    Vec3 b1 = sub(points[1], points[0]);
    Vec3 b2 = cross(normal, b1);
    b1 = normalize(b1);
    b2 = normalize(b2);

    p.points_in_basis.reserve(points.size());
    for (const Vec3 &pt : points) {
        Vec3 offset = sub(pt, points[0]);
        p.points_in_basis.push_back({dot(offset, b1), dot(offset, b2)});
    }
    Vec3 centroid{0.0, 0.0, 0.0};
    for (const Vec3 &pt : points) centroid = add(centroid, pt);
    centroid = scale(1.0 / static_cast<double>(points.size()), centroid);
    double radius = dist(centroid, points[0]);
    for (const Vec3 &pt : points) {
        double r = dist(centroid, pt);
        if (r > radius) radius = r;
    }
    p.points = points;
    p.material = brdf;
    p.bounding_center = centroid;
    p.bounding_radius = radius;
    p.normal = normal;
    p.basis_1 = b1;
    p.basis_2 = b2;
    return p;
}

std::vector<Polygon> Polygon::parallelpiped(const Vec3 &origin, const Vec3 &gen1, const Vec3 &gen2,
                                            const Vec3 &gen3, const Brdf &brdf) {
    auto quad = [&](const Vec3 &a, const Vec3 &b) {
        return std::vector<Vec3>{origin, add(origin, a), add(add(origin, a), b), add(origin, b)};
    };
    auto shifted = [](std::vector<Vec3> f, const Vec3 &g) {
        for (Vec3 &x : f) x = add(x, g);
        return f;
    };
    std::vector<Vec3> f1 = quad(gen1, gen2), f2 = quad(gen2, gen3), f3 = quad(gen1, gen3);
    std::vector<Polygon> out;
    out.push_back(new_cached(f1, brdf, scale(-1.0, gen3)));
    out.push_back(new_cached(f2, brdf, scale(-1.0, gen1)));
    out.push_back(new_cached(f3, brdf, scale(-1.0, gen2)));
    out.push_back(new_cached(shifted(f1, gen3), brdf, gen3));
    out.push_back(new_cached(shifted(f2, gen1), brdf, gen1));
    out.push_back(new_cached(shifted(f3, gen2), brdf, gen2));
    return out;
}

std::array<double, 4> Polygon::package() const {
    Vec3 mean{0.0, 0.0, 0.0};
    for (const Vec3 &pt : points) mean = add(mean, pt);
    mean = scale(1.0 / static_cast<double>(points.size()), mean);
    double radius = 0.0;
    for (const Vec3 &pt : points) {
        double r = dist(pt, mean);
        radius = radius < r ? r : radius; // geometry::max
    }
    return {mean[0], mean[1], mean[2], radius};
}

// ---------------------------------------------------------------------------
// OBJ, room
// ---------------------------------------------------------------------------
namespace {

std::string collapse_and_trim(const std::string &line) {
    // line.replace("  ", " ") is one non-overlapping left-to-right pass; then trim()
    std::string s;
    s.reserve(line.size());
    for (size_t i = 0; i < line.size();) {
        if (i + 1 < line.size() && line[i] == ' ' && line[i + 1] == ' ') {
            s.push_back(' ');
            i += 2;
        } else {
            s.push_back(line[i++]);
        }
    }
    const char *ws = " \t\r\n\f\v";
    size_t a = s.find_first_not_of(ws);
    if (a == std::string::npos) return std::string();
    size_t b = s.find_last_not_of(ws);
    return s.substr(a, b - a + 1);
}

std::vector<std::string> split(const std::string &s, char sep) {
    std::vector<std::string> out;
    size_t a = 0;
    while (true) {
        size_t b = s.find(sep, a);
        out.push_back(s.substr(a, b == std::string::npos ? std::string::npos : b - a));
        if (b == std::string::npos) break;
        a = b + 1;
    }
    return out;
}

double parse_f64(const std::string &t) {
    if (t.empty()) throw std::invalid_argument("OBJ: empty number");
    char *end = nullptr;
    double v = std::strtod(t.c_str(), &end);
    if (end == t.c_str() || *end != 0) throw std::invalid_argument("OBJ: bad number '" + t + "'");
    return v;
}
size_t parse_index(const std::string &t) {
    if (t.empty()) throw std::invalid_argument("OBJ: empty index");
    char *end = nullptr;
    unsigned long long v = std::strtoull(t.c_str(), &end, 10);
    if (end == t.c_str() || *end != 0) throw std::invalid_argument("OBJ: bad index '" + t + "'");
    return static_cast<size_t>(v);
}

} // namespace

std::vector<std::unique_ptr<Primitive>> polys_from_obj_text(const std::string &text, double z_shift,
                                                            const Brdf &brdf) {
    std::vector<std::unique_ptr<Primitive>> objects;
    std::vector<Vec3> vertices, normals;
    for (const std::string &line : split(text, '\n')) {
        bool is_v = line.rfind("v ", 0) == 0, is_vn = line.rfind("vn ", 0) == 0;
        if (is_v || is_vn) {
            Vec3 c{0.0, 0.0, 0.0};
            std::vector<std::string> items = split(collapse_and_trim(line), ' ');
            // main.rs:187-190 indexes past its three coordinates (a panic) on a fourth number. Deviation:
            // `v x y z r g b`, the vertex colours MeshLab writes (the reference's own duck.obj), is read by
            // its first three numbers; anything else with more than three is still an error.
            if (items.size() > 4 && !(is_v && items.size() == 7))
                throw std::invalid_argument("OBJ: more than three coordinates");
            for (size_t k = 1; k < items.size(); k++) {
                double value = parse_f64(items[k]);
                if (k <= 3) c[k - 1] = value;
            }
            c[2] += z_shift; // main.rs:193 (applied to normals too, which are never read)
            (is_v ? vertices : normals).push_back(c);
        } else if (line.rfind("f ", 0) == 0) {
            std::vector<std::string> items = split(collapse_and_trim(line), ' ');
            std::vector<Vec3> pts;
            for (size_t k = 1; k < items.size(); k++) {
                std::vector<std::string> sub_items = split(items[k], '/');
                if (sub_items.size() < 3) throw std::invalid_argument("OBJ: face item needs v/t/n");
                size_t vi = parse_index(sub_items[0]);
                (void)parse_index(sub_items[2]); // parsed, never used (main.rs:211)
                if (vi == 0 || vi > vertices.size()) throw std::invalid_argument("OBJ: vertex index out of range");
                pts.push_back(vertices[vi - 1]);
            }
            if (pts.size() < 3) throw std::invalid_argument("OBJ: face with fewer than 3 vertices");
            Vec3 dir = cross(sub(pts[1], pts[0]), sub(pts[2], pts[1])); // main.rs:222-225
            objects.push_back(std::make_unique<Polygon>(Polygon::new_cached(pts, brdf, dir)));
        }
    }
    return objects;
}

std::vector<std::unique_ptr<Primitive>> polys_from_obj(const std::string &file_name, double z_shift) {
    std::ifstream in(file_name, std::ios::binary);
    if (!in) throw std::runtime_error("something went wrong with reading the file: " + file_name);
    std::stringstream ss;
    ss << in.rdbuf();
    Brdf b;
    b.reflection_spectrum = {0.65, 0.65, 0.65}; // main.rs:229-233
    return polys_from_obj_text(ss.str(), z_shift, b);
}

std::vector<std::unique_ptr<Primitive>> reference_room() {
    const double z = -1.0, x = 2.0, y = 2.0, h = 0.5;
    std::vector<std::unique_ptr<Primitive>> o;
    Brdf floor_b{{0.75, 0.95, 0.5}, 0.0, 1.0}, light{{1.0, 1.0, 1.0}, 25.0, 1.0}, grey{{0.95, 0.95, 0.95}, 0.0, 1.0};
    o.push_back(std::make_unique<Polygon>(Polygon::new_cached(
        {{x, y, z}, {-x, y, z}, {-x, -y, z}, {x, -y, z}}, floor_b, {0.0, 0.0, 1.0})));
    o.push_back(std::make_unique<Polygon>(Polygon::new_cached(
        {{-x - 1.0, y, z + h}, {-x - 1.0, -y, z + h}, {-x - 1.0, -y, -z + h}, {-x - 1.0, y, -z + h}}, light,
        {1.0, 0.0, 0.0})));
    o.push_back(std::make_unique<Polygon>(Polygon::new_cached(
        {{x, y, z}, {x, -y, z}, {x, -y, z + 3.0}, {x, y, z + 3.0}}, grey, {-1.0, 0.0, 0.0})));
    o.push_back(std::make_unique<Polygon>(Polygon::new_cached(
        {{x, y, z}, {-x, y, z}, {-x, y, -z + 1.0}, {x, y, -z + 1.0}}, grey, {0.0, -1.0, 0.0})));
    return o;
}

// ---------------------------------------------------------------------------
// BVH::from_list -> flattened nodes
// ---------------------------------------------------------------------------
rt_scene_desc FlatScene::desc() const {
    rt_scene_desc d;
    std::memset(&d, 0, sizeof d);
    d.struct_size = sizeof(rt_scene_desc);
    d.n_nodes = static_cast<uint32_t>(nodes.size());
    d.nodes = nodes.data();
    d.n_slots = static_cast<uint32_t>(slots.size());
    d.slots = slots.data();
    d.n_polygons = static_cast<uint32_t>(polygons.size());
    d.polygons = polygons.data();
    d.points2d = points2d.data();
    d.n_points2d = static_cast<uint32_t>(points2d.size() / 2);
    d.n_spheres = static_cast<uint32_t>(spheres.size());
    d.spheres = spheres.data();
    d.materials = materials.data();
    d.n_materials = static_cast<uint32_t>(materials.size());
    return d;
}

std::array<double, 4> BVH::package(const std::vector<const Primitive *> &objects) {
    if (objects.empty()) return {0.0, 0.0, 0.0, 0.0};
    Vec3 mean{0.0, 0.0, 0.0};
    for (const Primitive *p : objects) {
        auto b = p->package();
        mean = add(mean, {b[0], b[1], b[2]});
    }
    mean = scale(1.0 / static_cast<double>(objects.size()), mean);
    double radius = 0.0;
    for (const Primitive *p : objects) {
        auto b = p->package();
        double r = dist({b[0], b[1], b[2]}, mean) + b[3];
        radius = radius < r ? r : radius;
    }
    return {mean[0], mean[1], mean[2], radius};
}

namespace {

struct Flattener {
    const std::vector<std::unique_ptr<Primitive>> &objects;
    std::vector<std::array<double, 4>> pk; // Primitive::package(), computed once per primitive
    FlatScene &out;
    std::map<std::array<uint64_t, 6>, uint32_t> material_index;

    Flattener(const std::vector<std::unique_ptr<Primitive>> &o, FlatScene &f) : objects(o), out(f) {
        pk.reserve(o.size());
        for (const auto &p : o) pk.push_back(p->package());
    }

    uint32_t material_of(const Brdf &b) {
        std::array<uint64_t, 6> key;
        double vals[6] = {b.reflection_spectrum[0], b.reflection_spectrum[1], b.reflection_spectrum[2], b.emissivity,
                          b.inside_index_of_refraction, b.gloss};
        std::memcpy(key.data(), vals, sizeof vals);
        auto it = material_index.find(key);
        if (it != material_index.end()) return it->second;
        rt_material m;
        std::memset(&m, 0, sizeof m);
        std::memcpy(m.reflection_spectrum, vals, 3 * sizeof(double));
        m.emissivity = b.emissivity;
        m.inside_index_of_refraction = b.inside_index_of_refraction;
        m.gloss = b.gloss;
        uint32_t idx = static_cast<uint32_t>(out.materials.size());
        out.materials.push_back(m);
        material_index.emplace(key, idx);
        return idx;
    }

    std::array<double, 4> shell_of(const std::vector<uint32_t> &ids) const { // bvh.rs:77-95
        if (ids.empty()) return {0.0, 0.0, 0.0, 0.0};
        Vec3 mean{0.0, 0.0, 0.0};
        for (uint32_t i : ids) mean = add(mean, {pk[i][0], pk[i][1], pk[i][2]});
        mean = scale(1.0 / static_cast<double>(ids.size()), mean);
        double radius = 0.0;
        for (uint32_t i : ids) {
            double r = dist({pk[i][0], pk[i][1], pk[i][2]}, mean) + pk[i][3];
            radius = radius < r ? r : radius;
        }
        return {mean[0], mean[1], mean[2], radius};
    }

    void emit_leaf(uint32_t node, const std::vector<uint32_t> &ids) {
        rt_node &n = out.nodes[node];
        n.slot_begin = static_cast<uint32_t>(out.slots.size());
        n.slot_count = static_cast<uint32_t>(ids.size());
        n.left = 0;
        for (uint32_t id : ids) {
            const Primitive *p = objects[id].get();
            if (p->is_sphere()) {
                const Sphere *s = static_cast<const Sphere *>(p);
                rt_sphere r;
                std::memset(&r, 0, sizeof r);
                std::memcpy(r.center, s->center.data(), sizeof r.center);
                r.radius = s->radius;
                r.material = material_of(s->material);
                r.prim_id = id;
                out.slots.push_back(RT_SLOT_SPHERE | static_cast<uint32_t>(out.spheres.size()));
                out.spheres.push_back(r);
            } else {
                const Polygon *g = static_cast<const Polygon *>(p);
                rt_polygon r;
                std::memset(&r, 0, sizeof r);
                std::memcpy(r.normal, g->normal.data(), 24);
                r.bounding_radius = g->bounding_radius;
                std::memcpy(r.p0, g->points[0].data(), 24);
                std::memcpy(r.bounding_center, g->bounding_center.data(), 24);
                std::memcpy(r.basis_1, g->basis_1.data(), 24);
                std::memcpy(r.basis_2, g->basis_2.data(), 24);
                r.n_points = static_cast<uint32_t>(g->points_in_basis.size());
                r.points_offset = static_cast<uint32_t>(out.points2d.size() / 2);
                r.material = material_of(g->material);
                r.prim_id = id;
                for (const auto &q : g->points_in_basis) {
                    out.points2d.push_back(q[0]);
                    out.points2d.push_back(q[1]);
                }
                out.slots.push_back(static_cast<uint32_t>(out.polygons.size()));
                out.polygons.push_back(r);
            }
        }
        out.info.n_leaves++;
        if (ids.size() > out.info.max_leaf) out.info.max_leaf = static_cast<uint32_t>(ids.size());
    }

    // bvh.rs:21-75. Emits node `ids` and its subtree in right-first pre-order.
    void build(std::vector<uint32_t> ids, uint32_t depth) {
        uint32_t node = static_cast<uint32_t>(out.nodes.size());
        out.nodes.emplace_back();
        std::memset(&out.nodes[node], 0, sizeof(rt_node));
        if (depth > out.info.depth) out.info.depth = depth;
        auto shell = shell_of(ids);
        std::memcpy(out.nodes[node].shell, shell.data(), sizeof shell);

        if (ids.size() <= 4) {
            emit_leaf(node, ids);
            out.nodes[node].skip = static_cast<uint32_t>(out.nodes.size());
            return;
        }
        Vec3 mean_pos{shell[0], shell[1], shell[2]};
        Vec3 mean_dir{0.0, 0.0, 0.0};
        for (uint32_t i : ids) {
            Vec3 obj_dir = sub({pk[i][0], pk[i][1], pk[i][2]}, mean_pos);
            if (mag(obj_dir) > EPSILON) obj_dir = normalize(obj_dir);
            else obj_dir = {0.0, 0.0, 0.01};
            if (dot(obj_dir, mean_dir) < 0.0) obj_dir = scale(-1.0, obj_dir);
            mean_dir = add(mean_dir, obj_dir);
        }
        std::vector<uint32_t> left_ids, right_ids;
        for (uint32_t i : ids) {
            if (dot(sub({pk[i][0], pk[i][1], pk[i][2]}, mean_pos), mean_dir) > 0.0) left_ids.push_back(i);
            else right_ids.push_back(i);
        }
        if (left_ids.empty() || right_ids.empty()) {
            // Every centre fell on one side: the reference recurses forever here.
            // Keeping the list as one oversized leaf cannot change any closest hit.
            out.info.guard_leaves++;
            emit_leaf(node, ids);
            out.nodes[node].skip = static_cast<uint32_t>(out.nodes.size());
            return;
        }
        ids.clear();
        ids.shrink_to_fit();
        build(std::move(right_ids), depth + 1); // lands at node + 1
        out.nodes[node].left = static_cast<uint32_t>(out.nodes.size());
        build(std::move(left_ids), depth + 1);
        out.nodes[node].skip = static_cast<uint32_t>(out.nodes.size());
    }
};

} // namespace

BVH BVH::from_list(std::vector<std::unique_ptr<Primitive>> objects) {
    BVH b;
    b.objects_ = std::move(objects);
    b.flat_ = std::make_unique<FlatScene>();
    Flattener f(b.objects_, *b.flat_);
    std::vector<uint32_t> ids(b.objects_.size());
    for (uint32_t i = 0; i < ids.size(); i++) ids[i] = i;
    b.flat_->info.n_prims = static_cast<uint32_t>(ids.size());
    f.build(std::move(ids), 1);
    b.flat_->info.n_nodes = static_cast<uint32_t>(b.flat_->nodes.size());
    return b;
}

BVH::BVH(BVH &&o) noexcept : objects_(std::move(o.objects_)), flat_(std::move(o.flat_)), scene_(o.scene_) {
    o.scene_ = nullptr;
}
BVH &BVH::operator=(BVH &&o) noexcept {
    if (this != &o) {
        if (scene_) rt_scene_destroy(scene_);
        objects_ = std::move(o.objects_);
        flat_ = std::move(o.flat_);
        scene_ = o.scene_;
        o.scene_ = nullptr;
    }
    return *this;
}
BVH::~BVH() {
    if (scene_) rt_scene_destroy(scene_);
}

void BVH::upload(int device) {
    if (scene_) {
        rt_scene_destroy(scene_);
        scene_ = nullptr;
    }
    rt_scene_desc d = flat_->desc();
    if (rt_scene_create(&d, device, &scene_) != RT_OK) throw std::runtime_error(rt_last_error());
}
rt_scene *BVH::scene() {
    if (!scene_) upload(0);
    return scene_;
}

std::vector<BVH::Hit> BVH::intersect_batch(const std::vector<double> &rays, rt_traverse_mode mode) {
    size_t n = rays.size() / 6;
    std::vector<uint32_t> ids(n);
    std::vector<double> dist_(n), pos(3 * n), nrm(3 * n);
    if (rt_intersect_batch(scene(), rays.data(), n, mode, ids.data(), dist_.data(), pos.data(), nrm.data(),
                           nullptr) != RT_OK)
        throw std::runtime_error(rt_last_error());
    std::vector<Hit> out(n);
    for (size_t i = 0; i < n; i++) {
        if (ids[i] == RT_MISS) continue;
        Hit &h = out[i];
        h.some = true;
        h.prim_id = ids[i];
        h.distance = dist_[i];
        for (int k = 0; k < 3; k++) {
            h.pos[k] = pos[3 * i + k];
            h.normal[k] = nrm[3 * i + k];
        }
        h.brdf = &objects_[ids[i]]->brdf();
    }
    return out;
}

BVH::Hit BVH::intersect(const Vec3 &o, const Vec3 &d, rt_traverse_mode mode) {
    std::vector<double> ray{o[0], o[1], o[2], d[0], d[1], d[2]};
    return intersect_batch(ray, mode)[0];
}

Vec3 trace_ray(const Vec3 &o, const Vec3 &d, BVH &bvh, uint64_t seed, uint32_t pixel, uint32_t sample,
               uint32_t sub_ray) {
    double ray[6] = {o[0], o[1], o[2], d[0], d[1], d[2]};
    uint32_t key[3] = {pixel, sample, sub_ray};
    Vec3 out{};
    if (rt_trace_paths(bvh.scene(), ray, key, 1, seed, RT_MAX_RAY_DEPTH, RT_TRAVERSE_FAST, out.data(), nullptr) !=
        RT_OK)
        throw std::runtime_error(rt_last_error());
    return out;
}

namespace camera {

std::vector<double> ray_bundle(const rt_camera &cam, uint32_t width, uint32_t height, uint64_t seed) {
    std::vector<double> rays(static_cast<size_t>(width) * height * 30);
    if (rt_ray_bundle(&cam, width, height, seed, 0, rays.data()) != RT_OK) throw std::runtime_error(rt_last_error());
    return rays;
}

std::vector<double> ray_bundle(const Vec3 &camera_loc, const std::array<Vec3, 3> &camera_dir,
                               double camera_focal_length, double camera_sensor_to_lens, double sensor_diagonal,
                               double aperture_radius, size_t image_resolution, uint64_t seed) {
    rt_camera c;
    std::memcpy(c.loc, camera_loc.data(), 24);
    for (int r = 0; r < 3; r++)
        for (int k = 0; k < 3; k++) c.dir[3 * r + k] = camera_dir[r][k];
    c.focal_length = camera_focal_length;
    c.sensor_to_lens = camera_sensor_to_lens;
    c.sensor_diagonal = sensor_diagonal;
    c.aperture_radius = aperture_radius;
    return ray_bundle(c, static_cast<uint32_t>(image_resolution), static_cast<uint32_t>(image_resolution), seed);
}

std::array<uint8_t, 4> spectral_radiance_to_color(const Vec3 &sr) {
    std::array<uint8_t, 4> out{};
    if (rt_tone_map(sr.data(), 1, 0, out.data()) != RT_OK) throw std::runtime_error(rt_last_error());
    return out;
}

} // namespace camera
// ---------------------------------------------------------------------------
// write_to_png (main.rs:242-244)
// ---------------------------------------------------------------------------
namespace {

uint32_t crc32_of(const uint8_t *data, size_t n, uint32_t crc) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ data[i]) & 0xFFu] ^ (crc >> 8);
    return ~crc;
}

void put_be32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}

void put_chunk(std::vector<uint8_t> &png, const char type[4], const std::vector<uint8_t> &body) {
    put_be32(png, (uint32_t)body.size());
    size_t start = png.size();
    png.insert(png.end(), type, type + 4);
    png.insert(png.end(), body.begin(), body.end());
    put_be32(png, crc32_of(png.data() + start, png.size() - start, 0u));
}

} // namespace

void write_to_png(const std::string &file_name, const uint8_t *rgba, uint32_t width, uint32_t height) {
    if (!rgba || width == 0 || height == 0 || (uint64_t)width * height > (1ull << 28))
        throw std::invalid_argument("write_to_png: bad image");
    // scanlines: filter byte 0 (none) + RGBA
    const size_t row = (size_t)width * 4;
    std::vector<uint8_t> raw;
    raw.reserve((row + 1) * height);
    for (uint32_t y = 0; y < height; y++) {
        raw.push_back(0);
        raw.insert(raw.end(), rgba + y * row, rgba + (y + 1) * row);
    }
    // zlib stream of stored blocks (at most 65,535 bytes each)
    std::vector<uint8_t> z;
    z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0; // Adler-32
    for (size_t pos = 0; pos < raw.size();) {
        size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
        for (size_t i = 0; i < n; i++) {
            a = (a + raw[pos + i]) % 65521u;
            b = (b + a) % 65521u;
        }
        z.insert(z.end(), raw.begin() + (std::ptrdiff_t)pos, raw.begin() + (std::ptrdiff_t)(pos + n));
        pos += n;
    }
    put_be32(z, (b << 16) | a);
    std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, width);
    put_be32(ihdr, height);
    ihdr.push_back(8); // bit depth
    ihdr.push_back(6); // colour type: RGBA (image::RGBA(8), main.rs:243)
    ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    put_chunk(png, "IHDR", ihdr);
    put_chunk(png, "IDAT", z);
    put_chunk(png, "IEND", {});
    std::FILE *f = std::fopen(file_name.c_str(), "wb");
    if (!f) throw std::runtime_error("write_to_png: cannot open " + file_name);
    size_t written = std::fwrite(png.data(), 1, png.size(), f);
    if (std::fclose(f) != 0 || written != png.size()) throw std::runtime_error("write_to_png: cannot write " + file_name);
}

} // namespace rt
