// rt_host.hpp — host-side mirror of the reference crate's scene and BVH entry
// points (rust_tracing/src/brdf.rs, primitives.rs, bvh.rs, main.rs:117-240).
//
// The host builds primitives exactly as the reference constructors do, runs
// BVH::from_list's split rule, and emits the tree directly in its flattened
// form (rt_scene_desc of rt_amd.h). It never intersects anything on the CPU:
// BVH::intersect and friends forward to the GPU library.
#pragma once

#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt_amd.h"

namespace rt {

using Vec3 = std::array<double, 3>;

// brdf.rs:4-10
struct Brdf {
    Vec3 reflection_spectrum{0.0, 0.0, 0.0};
    double emissivity = 0.0;
    double inside_index_of_refraction = 1.0;
    double gloss = 0.0; // 0 = ideal diffuse (brdf.rs:14-29); (0, 1] = the lobe of geometry.js:172-184 with n = 1 - gloss
};

// primitives.rs:5-14 (intersect/normal_at run on the GPU, so only the
// build-time half of the trait exists on the host)
class Primitive {
public:
    virtual ~Primitive() = default;
    virtual std::array<double, 4> package() const = 0;
    virtual const Brdf &brdf() const = 0;
    virtual bool is_sphere() const = 0;
};

// primitives.rs:35-39
class Sphere final : public Primitive {
public:
    Vec3 center;
    double radius;
    Brdf material;
    Sphere(const Vec3 &c, double r, const Brdf &b) : center(c), radius(r), material(b) {}
    std::array<double, 4> package() const override { return {center[0], center[1], center[2], radius}; }
    const Brdf &brdf() const override { return material; }
    bool is_sphere() const override { return true; }
};

// primitives.rs:99-114
class Polygon final : public Primitive {
public:
    std::vector<Vec3> points;
    Brdf material;
    Vec3 bounding_center;
    double bounding_radius;
    Vec3 normal;
    Vec3 basis_1, basis_2;
    std::vector<std::array<double, 2>> points_in_basis;

    // primitives.rs:144-199; throws std::invalid_argument where the reference panics
    static Polygon new_cached(const std::vector<Vec3> &points, const Brdf &brdf, const Vec3 &direc_vec);
    // primitives.rs:117-141
    static std::vector<Polygon> parallelpiped(const Vec3 &origin, const Vec3 &gen1, const Vec3 &gen2,
                                              const Vec3 &gen3, const Brdf &brdf);
    std::array<double, 4> package() const override; // primitives.rs:255-267
    const Brdf &brdf() const override { return material; }
    bool is_sphere() const override { return false; }
};

// main.rs:172-240 on an in-memory OBJ text
std::vector<std::unique_ptr<Primitive>> polys_from_obj_text(const std::string &text, double z_shift,
                                                            const Brdf &brdf);
// main.rs:172-240 reading a file; throws std::runtime_error if it cannot be read
std::vector<std::unique_ptr<Primitive>> polys_from_obj(const std::string &file_name, double z_shift = -0.717);
// write_to_png (main.rs:242-244): an RGBA8 buffer as a PNG file. The reference goes through the `image`
// crate; this writer needs no library: one IDAT chunk holding a zlib stream of stored (uncompressed)
// blocks, CRC-32 per chunk, Adler-32 over the scanlines. Throws std::runtime_error if the file cannot be
// written.
void write_to_png(const std::string &file_name, const uint8_t *rgba, uint32_t width, uint32_t height);
// the four quads of init_environment (main.rs:117-161)
std::vector<std::unique_ptr<Primitive>> reference_room();

struct BuildInfo {
    uint32_t n_nodes = 0, n_leaves = 0, depth = 0, n_prims = 0, max_leaf = 0, guard_leaves = 0;
};

// The flattened tree. Owns the arrays an rt_scene_desc points into.
class FlatScene {
public:
    std::vector<rt_node> nodes;
    std::vector<uint32_t> slots;
    std::vector<rt_polygon> polygons;
    std::vector<double> points2d;
    std::vector<rt_sphere> spheres;
    std::vector<rt_material> materials;
    BuildInfo info;
    rt_scene_desc desc() const;
};

// bvh.rs:10-18
class BVH {
public:
    // bvh.rs:21-75; consumes the list. prim ids are list positions.
    static BVH from_list(std::vector<std::unique_ptr<Primitive>> objects);
    // bvh.rs:77-95
    static std::array<double, 4> package(const std::vector<const Primitive *> &objects);

    const FlatScene &flat() const { return *flat_; }
    const std::vector<std::unique_ptr<Primitive>> &objects() const { return objects_; }

    // Uploads on first use (device 0 unless upload() was called). GPU only:
    // throws std::runtime_error carrying rt_last_error() when no device exists.
    void upload(int device = 0);
    rt_scene *scene();

    struct Hit {
        bool some = false;  // Option::Some
        Vec3 pos{}, normal{};
        const Brdf *brdf = nullptr;
        uint32_t prim_id = RT_MISS;
        double distance = 0.0;
    };
    // bvh.rs:97: one ray (a whole GPU launch for one ray: use intersect_batch for throughput)
    Hit intersect(const Vec3 &ray_origin, const Vec3 &ray_dir, rt_traverse_mode mode = RT_TRAVERSE_FAST);
    std::vector<Hit> intersect_batch(const std::vector<double> &rays, rt_traverse_mode mode = RT_TRAVERSE_FAST);

    BVH(BVH &&) noexcept;
    BVH &operator=(BVH &&) noexcept;
    ~BVH();

private:
    BVH() = default;
    std::vector<std::unique_ptr<Primitive>> objects_;
    std::unique_ptr<FlatScene> flat_;
    rt_scene *scene_ = nullptr;
};

// tracer.rs:8: one path; the random stream is (seed, pixel, sample, sub_ray)
Vec3 trace_ray(const Vec3 &ray_origin, const Vec3 &ray_dir, BVH &bvh, uint64_t seed = 0, uint32_t pixel = 0,
               uint32_t sample = 0, uint32_t sub_ray = 0);

namespace camera {
// camera.rs:16 (square image, as the reference) and its width x height generalisation
std::vector<double> ray_bundle(const Vec3 &camera_loc, const std::array<Vec3, 3> &camera_dir,
                               double camera_focal_length, double camera_sensor_to_lens, double sensor_diagonal,
                               double aperture_radius, size_t image_resolution, uint64_t seed = 0);
std::vector<double> ray_bundle(const rt_camera &cam, uint32_t width, uint32_t height, uint64_t seed = 0);
// camera.rs:80
std::array<uint8_t, 4> spectral_radiance_to_color(const Vec3 &spectral_radiance);
} // namespace camera

} // namespace rt
