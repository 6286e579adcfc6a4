// rt_host_c.cpp — C wrappers (rt_host.h) over the C++ host mirror.
#include <cstring>
#include <exception>

#include "rt_error.hpp"
#include "rt_host.h"
#include "rt_host.hpp"

struct rt_builder {
    std::vector<std::unique_ptr<rt::Primitive>> pending; // the list that from_list will consume
    std::unique_ptr<rt::BVH> bvh;
};

namespace {

rt::Brdf to_brdf(const rt_material *m) {
    rt::Brdf b;
    b.reflection_spectrum = {m->reflection_spectrum[0], m->reflection_spectrum[1], m->reflection_spectrum[2]};
    b.emissivity = m->emissivity;
    b.inside_index_of_refraction = m->inside_index_of_refraction;
    b.gloss = m->gloss;
    return b;
}
rt::Vec3 v3(const double *p) { return {p[0], p[1], p[2]}; }

template <class F> rt_status guarded(F &&f) {
    try {
        return f();
    } catch (const std::invalid_argument &e) {
        return rt::fail(RT_ERR_INVALID_ARGUMENT, e.what());
    } catch (const std::bad_alloc &) {
        return rt::fail(RT_ERR_OUT_OF_MEMORY, "host allocation failed");
    } catch (const std::exception &e) {
        return rt::fail(RT_ERR_INVALID_ARGUMENT, e.what());
    }
}

} // namespace

extern "C" {

rt_status rtb_create(rt_builder **out) {
    if (!out) return rt::fail(RT_ERR_INVALID_ARGUMENT, "rtb_create: out is NULL");
    return guarded([&] {
        *out = new rt_builder();
        return RT_OK;
    });
}

rt_status rtb_destroy(rt_builder *b) {
    delete b;
    return RT_OK;
}

rt_status rtb_add_sphere(rt_builder *b, const double center[3], double radius, const rt_material *brdf,
                         uint32_t *prim_id_out) {
    if (!b || !center || !brdf) return rt::fail(RT_ERR_INVALID_ARGUMENT, "rtb_add_sphere: NULL argument");
    return guarded([&] {
        if (prim_id_out) *prim_id_out = static_cast<uint32_t>(b->pending.size());
        b->pending.push_back(std::make_unique<rt::Sphere>(v3(center), radius, to_brdf(brdf)));
        return RT_OK;
    });
}

rt_status rtb_add_polygon(rt_builder *b, const double *points, uint32_t n_points, const rt_material *brdf,
                          const double direc_vec[3], uint32_t *prim_id_out) {
    if (!b || !points || !brdf || !direc_vec) return rt::fail(RT_ERR_INVALID_ARGUMENT, "rtb_add_polygon: NULL argument");
    return guarded([&] {
        std::vector<rt::Vec3> pts(n_points);
        for (uint32_t i = 0; i < n_points; i++) pts[i] = v3(points + 3 * i);
        auto poly = std::make_unique<rt::Polygon>(rt::Polygon::new_cached(pts, to_brdf(brdf), v3(direc_vec)));
        if (prim_id_out) *prim_id_out = static_cast<uint32_t>(b->pending.size());
        b->pending.push_back(std::move(poly));
        return RT_OK;
    });
}

rt_status rtb_add_parallelpiped(rt_builder *b, const double origin[3], const double gen1[3], const double gen2[3],
                                const double gen3[3], const rt_material *brdf, uint32_t *first_prim_id_out) {
    if (!b || !origin || !gen1 || !gen2 || !gen3 || !brdf)
        return rt::fail(RT_ERR_INVALID_ARGUMENT, "rtb_add_parallelpiped: NULL argument");
    return guarded([&] {
        auto faces = rt::Polygon::parallelpiped(v3(origin), v3(gen1), v3(gen2), v3(gen3), to_brdf(brdf));
        if (first_prim_id_out) *first_prim_id_out = static_cast<uint32_t>(b->pending.size());
        for (auto &f : faces) b->pending.push_back(std::make_unique<rt::Polygon>(std::move(f)));
        return RT_OK;
    });
}

rt_status rtb_add_mesh(rt_builder *b, const double *vertices, uint32_t nv, const uint32_t *faces, uint32_t nf,
                       uint32_t face_len, double z_shift, const rt_material *brdf, uint32_t *first_prim_id_out) {
    if (!b || !vertices || !faces || !brdf) return rt::fail(RT_ERR_INVALID_ARGUMENT, "rtb_add_mesh: NULL argument");
    if (face_len < 3) return rt::fail(RT_ERR_INVALID_ARGUMENT, "rtb_add_mesh: face_len < 3");
    return guarded([&] {
        rt::Brdf mat = to_brdf(brdf);
        std::vector<std::unique_ptr<rt::Primitive>> made;
        made.reserve(nf);
        std::vector<rt::Vec3> pts(face_len);
        for (uint32_t f = 0; f < nf; f++) {
            for (uint32_t k = 0; k < face_len; k++) {
                uint32_t vi = faces[static_cast<size_t>(f) * face_len + k];
                if (vi >= nv) throw std::invalid_argument("rtb_add_mesh: vertex index out of range");
                pts[k] = {vertices[3 * static_cast<size_t>(vi)], vertices[3 * static_cast<size_t>(vi) + 1],
                          vertices[3 * static_cast<size_t>(vi) + 2] + z_shift};
            }
            rt::Vec3 e0{pts[1][0] - pts[0][0], pts[1][1] - pts[0][1], pts[1][2] - pts[0][2]};
            rt::Vec3 e1{pts[2][0] - pts[1][0], pts[2][1] - pts[1][1], pts[2][2] - pts[1][2]};
            rt::Vec3 dir{e0[1] * e1[2] - e0[2] * e1[1], e0[2] * e1[0] - e0[0] * e1[2], e0[0] * e1[1] - e0[1] * e1[0]};
            made.push_back(std::make_unique<rt::Polygon>(rt::Polygon::new_cached(pts, mat, dir)));
        }
        if (first_prim_id_out) *first_prim_id_out = static_cast<uint32_t>(b->pending.size());
        for (auto &p : made) b->pending.push_back(std::move(p));
        return RT_OK;
    });
}

rt_status rtb_add_obj(rt_builder *b, const char *text, size_t len, double z_shift, const rt_material *brdf,
                      uint32_t *first_prim_id_out, uint32_t *n_added_out) {
    if (!b || !text || !brdf) return rt::fail(RT_ERR_INVALID_ARGUMENT, "rtb_add_obj: NULL argument");
    return guarded([&] {
        auto made = rt::polys_from_obj_text(std::string(text, len), z_shift, to_brdf(brdf));
        if (first_prim_id_out) *first_prim_id_out = static_cast<uint32_t>(b->pending.size());
        if (n_added_out) *n_added_out = static_cast<uint32_t>(made.size());
        for (auto &p : made) b->pending.push_back(std::move(p));
        return RT_OK;
    });
}

rt_status rtb_add_reference_room(rt_builder *b) {
    if (!b) return rt::fail(RT_ERR_INVALID_ARGUMENT, "rtb_add_reference_room: NULL builder");
    return guarded([&] {
        for (auto &p : rt::reference_room()) b->pending.push_back(std::move(p));
        return RT_OK;
    });
}

rt_status rtb_build(rt_builder *b) {
    if (!b) return rt::fail(RT_ERR_INVALID_ARGUMENT, "rtb_build: NULL builder");
    if (b->bvh) return rt::fail(RT_ERR_INVALID_ARGUMENT, "rtb_build: already built (from_list consumes its list)");
    return guarded([&] {
        b->bvh = std::make_unique<rt::BVH>(rt::BVH::from_list(std::move(b->pending)));
        b->pending.clear();
        return RT_OK;
    });
}

rt_status rtb_get_info(const rt_builder *b, rt_build_info *out) {
    if (!b || !out || !b->bvh) return rt::fail(RT_ERR_INVALID_ARGUMENT, "rtb_get_info: not built");
    const rt::BuildInfo &i = b->bvh->flat().info;
    std::memset(out, 0, sizeof *out);
    out->n_nodes = i.n_nodes;
    out->n_leaves = i.n_leaves;
    out->depth = i.depth;
    out->n_prims = i.n_prims;
    out->max_leaf = i.max_leaf;
    out->guard_leaves = i.guard_leaves;
    return RT_OK;
}

rt_status rtb_get_desc(const rt_builder *b, rt_scene_desc *out) {
    if (!b || !out || !b->bvh) return rt::fail(RT_ERR_INVALID_ARGUMENT, "rtb_get_desc: not built");
    *out = b->bvh->flat().desc();
    return RT_OK;
}

rt_status rtb_package(const rt_builder *b, uint32_t prim_id, double out[4]) {
    if (!b || !out) return rt::fail(RT_ERR_INVALID_ARGUMENT, "rtb_package: NULL argument");
    const auto &list = b->bvh ? b->bvh->objects() : b->pending;
    if (prim_id >= list.size()) return rt::fail(RT_ERR_INVALID_ARGUMENT, "rtb_package: prim_id out of range");
    auto p = list[prim_id]->package();
    std::memcpy(out, p.data(), sizeof p);
    return RT_OK;
}

rt_status rtb_write_png(const char *file_name, const uint8_t *rgba, uint32_t width, uint32_t height) {
    if (!file_name || !rgba) return rt::fail(RT_ERR_INVALID_ARGUMENT, "rtb_write_png: NULL argument");
    try {
        rt::write_to_png(file_name, rgba, width, height);
    } catch (const std::invalid_argument &e) {
        return rt::fail(RT_ERR_INVALID_ARGUMENT, e.what());
    } catch (const std::exception &e) {
        return rt::fail(RT_ERR_INTERNAL, e.what());
    }
    return RT_OK;
}

} // extern "C"
