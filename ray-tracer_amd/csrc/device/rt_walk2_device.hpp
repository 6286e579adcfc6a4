// rt_walk2_device.hpp — the fast walk with TWO rays per lane (traverse_fast2).
//
// traverse_fast (rt_walk_device.hpp) gives every lane one ray and lets the wavefront vote on the kind of round
// to run; a lane whose ray is of another kind, or waits for its kind to gather enough lanes, sits the round
// out, and about half of the lanes of an issued vector instruction carry no ray. Here every lane owns two
// rays with independent state, and in a round of kind K it works on whichever of its two rays is of kind K: a
// lane whose first ray waits for an exact test goes on walking with its second. The rounds are the same code
// on the same records in the same f64 arithmetic, so the hits are the same bits.
//
// What makes room for the second ray:
//  * registers: the kernels are compiled for 4 wavefronts per SIMD (128 registers) instead of 5 (96), i.e.
//    8 rays per SIMD lane in flight instead of 5;
//  * LDS: a stack entry is ONE word instead of a 4-byte reference + a 2-byte distance. That needs compact
//    child references (20 bits), which the scene provides when it has fewer than 65,528 primitives and at most
//    16,384 wide nodes of at most 15 levels (`Scene::qnodes_c`; larger scenes keep the one-ray walk). An entry
//    is (entry distance, top 12 bits of the float, rounded toward zero) | reference; push and pop are one LDS
//    operation each. The closest hit's position is not kept at all: the step along the ray is (8 bytes per
//    ray in LDS), and the position is recomputed from it when the ray is stored, with the expression the exact
//    test itself uses. 2 x 16 entries x 4 B + 2 x 8 B = 144 B per lane, four blocks per CU.
#pragma once

#include "rt_walk_device.hpp"

namespace rtd {
namespace {

// compact child reference (20 bits):
//   leaf   1 | count-1 (3 bits) | first slot (16 bits)
//   inner  0 | 0 | need (4 bits) | node index (14 bits)
//   group  0 | 1 | mask (4 bits) | node index (14 bits)      (stack entries only)
//   none   0xFFFFF
constexpr uint32_t C_LEAF = 1u << 19, C_GROUP = 1u << 18, C_CNT_SHIFT = 16, C_SLOT_MASK = 0xFFFFu;
constexpr uint32_t C_IDX_MASK = 0x3FFFu, C_NEED_SHIFT = 14, C_MASK_SHIFT = 14;
constexpr uint32_t C_NONE = 0xFFFFFu, C_REF_MASK = 0xFFFFFu, C_T_MASK = 0xFFF00000u;

#define RT_FAST2_BOUNDS __launch_bounds__(TBLOCK, 4)

// The state of one of a lane's two rays between rounds. Flags and kind are kept as bits of ordinary
// registers: as separate bools they would be lane masks in scalar register pairs, of which two rays' worth
// do not fit (the compiler then parks them in vector register lanes, one instruction per access).
struct Ray2 {
    float ox, oy, oz, dx, dy, dz, ix, iy, iz, oix, oiy, oiz, tlimit;
    double best_dist;
    uint32_t best_slot, cur, cand, ri, sp;
    uint32_t flags; // F_*
    uint32_t kind;  // K_*: what the ray waits for
};
constexpr uint32_t F_ACTIVE = 1u, F_PRE_OK = 2u, F_HAVE = 4u, F_OD = 8u;
// K_IDLE: the slot is empty, or its ray is finished (F_ACTIVE still set) and waits to be stored
constexpr uint32_t K_IDLE = 0u, K_NODE = 1u, K_LEAF = 2u, K_EXACT = 3u;
__device__ __forceinline__ uint32_t kind_of_ref(uint32_t cur) {
    return cur == C_NONE ? K_IDLE : ((cur & C_LEAF) ? K_LEAF : K_NODE);
}

// LDS of a block: [2][stack_depth][TBLOCK] stack words, then [2][TBLOCK] doubles (step of the closest hit)
inline size_t stack2_bytes(const Scene &sc) {
    return (size_t)2 * sc.stack_depth * TBLOCK * sizeof(uint32_t) + (size_t)2 * TBLOCK * sizeof(double);
}

// NOTE on the loop below: every exit and every `continue` of the main loop hangs on a value that is the same
// in all lanes (ballots, the ray supply). A per-lane loop counter as an extra exit makes the compiler treat the
// loop's exits as divergent, and the wavefront then never leaves it (measured: the kernel hangs).
template <class IO>
__device__ __forceinline__ void traverse_fast2(const Scene &sc, const IO &io, RaySupply supply, uint32_t *lds,
                                               const uint32_t vote_exact, const uint32_t vote_leaf,
                                               const uint32_t vote_refill) {
    const uint32_t D = sc.stack_depth;
    uint32_t *const stk = lds + threadIdx.x;                      // ray r, entry e: stk[(r * D + e) * TBLOCK]
    double *const bstep = reinterpret_cast<double *>(lds + (size_t)2 * D * TBLOCK) + threadIdx.x; // ray r: bstep[r * TBLOCK]
    const QNode *const nodes = sc.qnodes_c;
    Ray2 a;
    a.ox = a.oy = a.oz = a.dx = a.dy = a.dz = a.ix = a.iy = a.iz = a.oix = a.oiy = a.oiz = a.tlimit = 0.0f;
    a.best_dist = 0.0;
    a.best_slot = RT_MISS; a.cur = C_NONE; a.cand = 0; a.ri = 0; a.sp = 0; a.flags = 0; a.kind = K_IDLE;
    Ray2 b = a;
#ifdef RT_COUNT_VISITS
    uint32_t n_steps = 0, n_screened = 0, n_tested = 0, n_done = 0;
    auto tally = [&](int k, bool lane_takes_part) {
        unsigned long long m = lane_mask(lane_takes_part);
        if (lane_rank(~0ull) == 0) {
            atomicAdd(sc.visits + k, 1ull);
            atomicAdd(sc.visits + k + 1, (unsigned long long)__popcll(m));
        }
    };
#endif

    auto pop = [&](uint32_t *base, uint32_t &sp, float tlimit) -> uint32_t {
        while (sp > 0) {
            sp -= 1;
            const uint32_t e = base[sp * TBLOCK];
            if (__uint_as_float(e & C_T_MASK) <= tlimit) return e & C_REF_MASK;
        }
        return C_NONE;
    };

    for (;;) {
        const uint32_t n_exact = __popcll((lane_mask(a.kind == K_EXACT) | lane_mask(b.kind == K_EXACT)));
        const uint32_t n_leaf = __popcll((lane_mask(a.kind == K_LEAF) | lane_mask(b.kind == K_LEAF)));
        uint32_t n_node = __popcll((lane_mask(a.kind == K_NODE) | lane_mask(b.kind == K_NODE)));

        // ---- store finished rays, hand new rays to idle slots (one slot per lane and round) ----
        const unsigned long long idle = (lane_mask(a.kind == K_IDLE) | lane_mask(b.kind == K_IDLE));
        const uint32_t n_idle = __popcll(idle);
        if (n_idle >= vote_refill || n_exact + n_leaf + n_node == 0) {
            const uint64_t avail = supply.available();
            const uint32_t rank = lane_rank(idle);
            const bool a_idle = a.kind == K_IDLE, b_idle = b.kind == K_IDLE;
            const bool mine = a_idle || b_idle;
            // the slot this lane deals with in this round: a finished ray first (it must be stored whether or
            // not new rays are left), an empty slot otherwise
            const bool a_fin = a_idle && (a.flags & F_ACTIVE), b_fin = b_idle && (b.flags & F_ACTIVE);
            const bool in_a = a_fin || (!b_fin && a_idle);
            const bool take = mine && rank < avail;
            const uint32_t ri_new = (uint32_t)supply.next + rank;
            V3 o_new = {0.0, 0.0, 0.0}, dir_new = {0.0, 0.0, 0.0};
            if (take) io.load(ri_new, o_new, dir_new);
            const uint32_t fl = in_a ? a.flags : b.flags;
            if (mine && (fl & F_ACTIVE)) {
                const uint32_t ri = in_a ? a.ri : b.ri;
                if (fl & F_OD) {
                    io.store_flag(ri); // k_rewalk_* repeats this ray in reference order
                } else {
                    const bool have = (fl & F_HAVE) != 0u;
                    Closest best = {have, have ? (in_a ? a.best_slot : b.best_slot) : RT_MISS,
                                    have ? (in_a ? a.best_dist : b.best_dist) : 0.0, {0.0, 0.0, 0.0}};
                    if (have && io.needs_pos()) { // o + step * dir, as the exact test evaluated it
                        V3 o, dir;
                        io.load(ri, o, dir);
                        best.pos = hit_position(o, dir, bstep[in_a ? 0 : TBLOCK]);
                    }
                    io.store(sc, ri, best);
                }
                if (in_a) a.flags = 0;
                else b.flags = 0;
#ifdef RT_COUNT_VISITS
                n_done++;
#endif
            }
            if (avail != 0) {
#ifdef RT_COUNT_VISITS
                tally(10, take);
#endif
                if (take) {
                    const FRay fr = make_fray(sc, o_new, dir_new);
                    Ray2 r;
                    r.ox = fr.ox; r.oy = fr.oy; r.oz = fr.oz; r.dx = fr.dx; r.dy = fr.dy; r.dz = fr.dz;
                    r.ix = fr.ix; r.iy = fr.iy; r.iz = fr.iz; r.oix = fr.oix; r.oiy = fr.oiy; r.oiz = fr.oiz;
                    r.tlimit = __builtin_inff();
                    r.best_dist = 0.0;
                    r.best_slot = RT_MISS;
                    r.cur = fr.unit ? sc.root_ref_c : C_NONE; // not a unit direction: finished at once, for the re-walk
                    r.cand = 0;
                    r.ri = ri_new;
                    r.sp = 0;
                    r.flags = F_ACTIVE | (fr.pre_ok ? F_PRE_OK : 0u) | (fr.unit ? 0u : F_OD);
                    r.kind = kind_of_ref(r.cur);
                    if (in_a) a = r;
                    else b = r;
                }
                supply.next += (n_idle < avail) ? n_idle : avail;
                continue; // vote again with the new rays on board
            }
            if (lane_mask(((a.flags | b.flags) & F_ACTIVE) != 0u) == 0ull) { // supply exhausted and every ray stored
#ifdef RT_COUNT_VISITS
                atomicAdd(sc.visits + 0, (unsigned long long)n_done);
                atomicAdd(sc.visits + 1, (unsigned long long)n_steps);
                atomicAdd(sc.visits + 2, (unsigned long long)n_screened);
                atomicAdd(sc.visits + 3, (unsigned long long)n_tested);
#endif
                break;
            }
            if (n_exact + n_leaf + n_node == 0) continue;
        }

        if (n_exact >= vote_exact || (n_exact != 0 && n_leaf == 0 && n_node == 0)) {
            // ---- exact f64 test of one queued primitive per lane -------------------
            const bool ua = a.kind == K_EXACT;
#ifdef RT_COUNT_VISITS
            tally(4, ua || b.kind == K_EXACT);
#endif
            if (ua || b.kind == K_EXACT) {
                uint32_t cand = ua ? a.cand : b.cand, cur = ua ? a.cur : b.cur;
                const uint32_t ri = ua ? a.ri : b.ri;
                uint32_t best_slot = ua ? a.best_slot : b.best_slot;
                double best_dist = ua ? a.best_dist : b.best_dist;
                uint32_t fl = ua ? a.flags : b.flags;
                float tlimit = ua ? a.tlimit : b.tlimit;
                uint32_t sp = ua ? a.sp : b.sp;
                const uint32_t k = __ffs(cand) - 1;
                cand &= cand - 1;
                const uint32_t slot = (cur & C_SLOT_MASK) + k;
                PrimHot h = sc.hot[slot];
                V3 o, dir;
                double step;
                io.load(ri, o, dir);
#ifdef RT_COUNT_VISITS
                n_tested++;
#endif
                bool od = false;
                const bool is_hit = prim_intersect_step(sc, slot, h, o, dir, step, od);
                if (od) fl |= F_OD;
                if (is_hit) {
                    const V3 pos = hit_position(o, dir, step);
                    const double dd = mag(sub(pos, o));
                    // strict '<' keeps the first hit in reference visiting order; reference slots
                    // are numbered in that order, so an exact tie goes to the smaller one
                    if (!(fl & F_HAVE) || dd < best_dist ||
                        (dd == best_dist && sc.ref_of_fast[slot] < sc.ref_of_fast[best_slot])) {
                        fl |= F_HAVE;
                        best_slot = slot;
                        best_dist = dd;
                        bstep[ua ? 0 : TBLOCK] = step;
                        // the f32 ray starts `shift` along the ray (make_fray): the same expression
                        const V3 rc = {sc.root_c[0] - o.x, sc.root_c[1] - o.y, sc.root_c[2] - o.z};
                        double shift = dot(rc, dir) - 2.0 * sc.root_r;
                        shift = (shift > 0.0) ? shift : 0.0;
                        tlimit = prune_limit(sc, dd, shift);
                    }
                }
                uint32_t kind = K_EXACT;
                if (cand == 0) {
                    cur = pop(stk + (ua ? 0u : D * TBLOCK), sp, tlimit);
                    kind = kind_of_ref(cur);
                }
                if (ua) {
                    a.cand = cand; a.cur = cur; a.best_slot = best_slot; a.best_dist = best_dist;
                    a.flags = fl; a.tlimit = tlimit; a.sp = sp; a.kind = kind;
                } else {
                    b.cand = cand; b.cur = cur; b.best_slot = best_slot; b.best_dist = best_dist;
                    b.flags = fl; b.tlimit = tlimit; b.sp = sp; b.kind = kind;
                }
            }
        } else if (n_leaf >= vote_leaf || (n_leaf != 0 && n_node == 0)) {
            // ---- leaf: f32 screen of its primitives ---------------------------------
            const bool ua = a.kind == K_LEAF;
#ifdef RT_COUNT_VISITS
            tally(6, ua || b.kind == K_LEAF);
#endif
            if (ua || b.kind == K_LEAF) {
                const uint32_t cur0 = ua ? a.cur : b.cur;
                const float ox = ua ? a.ox : b.ox, oy = ua ? a.oy : b.oy, oz = ua ? a.oz : b.oz;
                const float dx = ua ? a.dx : b.dx, dy = ua ? a.dy : b.dy, dz = ua ? a.dz : b.dz;
                const bool pre_ok = ((ua ? a.flags : b.flags) & F_PRE_OK) != 0u;
                const uint32_t sb = cur0 & C_SLOT_MASK;
                const uint32_t cnt = ((cur0 >> C_CNT_SHIFT) & 7u) + 1u;
                uint32_t c = (1u << cnt) - 1u;
#ifdef RT_COUNT_VISITS
                n_screened += cnt;
#endif
                if (pre_ok) {
                    const uint32_t first = prefilter_pass2(sc, reinterpret_cast<const float4 *>(sc.pre2 + sb), ox, oy, oz, dx, dy, dz);
                    c = cnt > 1u ? first : first & 1u;
                    if (cnt > 2u)
                        for (uint32_t k = 2; k < cnt; k += 2) {
                            const uint32_t two = prefilter_pass2(sc, reinterpret_cast<const float4 *>(sc.pre2 + (sb + k)), ox,
                                                                 oy, oz, dx, dy, dz);
                            c |= (k + 1 < cnt ? two : two & 1u) << k;
                        }
                }
                uint32_t cur = cur0, kind = K_EXACT;
                uint32_t sp = ua ? a.sp : b.sp;
                if (c == 0) {
                    cur = pop(stk + (ua ? 0u : D * TBLOCK), sp, ua ? a.tlimit : b.tlimit);
                    kind = kind_of_ref(cur);
                }
                if (ua) {
                    a.cand = c; a.cur = cur; a.sp = sp; a.kind = kind;
                } else {
                    b.cand = c; b.cur = cur; b.sp = sp; b.kind = kind;
                }
            }
        } else {
            // ---- node steps (see traverse_fast: the same step on the compact references) ----
            const uint32_t cap = D;
            const uint32_t keep = n_node - (n_node * sc.node_drop >> 4);
            for (uint32_t it = 0; it < sc.node_iters; it++) {
                const bool ua = a.kind == K_NODE;
#ifdef RT_COUNT_VISITS
                tally(8, ua || b.kind == K_NODE);
#endif
                if (ua || b.kind == K_NODE) {
                    const uint32_t cur0 = ua ? a.cur : b.cur;
                    const float ix = ua ? a.ix : b.ix, iy = ua ? a.iy : b.iy, iz = ua ? a.iz : b.iz;
                    const float oix = ua ? a.oix : b.oix, oiy = ua ? a.oiy : b.oiy, oiz = ua ? a.oiz : b.oiz;
                    const float tlimit = ua ? a.tlimit : b.tlimit;
                    uint32_t sp = ua ? a.sp : b.sp;
                    uint32_t *const base = stk + (ua ? 0u : D * TBLOCK);
                    const bool grp = (cur0 & C_GROUP) != 0u;
                    const uint32_t idx = cur0 & C_IDX_MASK;
                    const uint32_t need = (cur0 >> C_NEED_SHIFT) & 15u; // meaningless for a group entry
#ifdef RT_COUNT_VISITS
                    n_steps++;
#endif
                    const NodeHits h = node_hits(nodes + idx, ix, iy, iz, oix, oiy, oiz, tlimit, grp,
                                                 (cur0 >> C_MASK_SHIFT) & 0xFu, C_NONE);
                    if (h.k1 != NO_CHILD) { // more than one child: the others wait on the stack
                        if (!grp && sp + 3u + need <= cap + 1u) {
                            if (h.k3 != NO_CHILD) { // farthest first, so that the nearest is popped first
                                base[sp * TBLOCK] = (h.k3 & C_T_MASK) | h.r3;
                                sp += 1;
                            }
                            if (h.k2 != NO_CHILD) {
                                base[sp * TBLOCK] = (h.k2 & C_T_MASK) | h.r2;
                                sp += 1;
                            }
                            base[sp * TBLOCK] = (h.k1 & C_T_MASK) | h.r1;
                            sp += 1;
                        } else {
                            const uint32_t rest = (1u << (h.k1 & 3u)) | (h.k2 != NO_CHILD ? 1u << (h.k2 & 3u) : 0u) |
                                                  (h.k3 != NO_CHILD ? 1u << (h.k3 & 3u) : 0u);
                            base[sp * TBLOCK] = (h.k1 & C_T_MASK) | C_GROUP | (rest << C_MASK_SHIFT) | idx;
                            sp += 1;
                        }
                    }
                    const uint32_t cur = (h.k0 != NO_CHILD) ? h.r0 : pop(base, sp, tlimit);
                    const uint32_t kind = kind_of_ref(cur);
                    if (ua) {
                        a.cur = cur; a.sp = sp; a.kind = kind;
                    } else {
                        b.cur = cur; b.sp = sp; b.kind = kind;
                    }
                }
                n_node = __popcll((lane_mask(a.kind == K_NODE) | lane_mask(b.kind == K_NODE)));
                if (n_node < keep) break;
            }
        }
    }
}

} // namespace
} // namespace rtd
