// rt_walk3_device.hpp — the fast walk with a PARKED next ray per lane (traverse_fast3).
//
// In traverse_fast (rt_walk_device.hpp) a lane whose ray is finished waits until enough lanes are in the same
// position to make a refill round worth its price (ray loads, the f64 set-up of the f32 ray, stores). Measured on
// the 1080p hippo render: while a node round runs, 12.9 of the 64 lanes have no ray at all (36.6 take part, 10.0
// wait for a leaf screen, 4.5 for an exact test): the largest single loss of lane utilisation.
//
// Here every lane holds, besides the ray it walks, the NEXT ray it will walk, already set up and parked in LDS
// (40 bytes: inverse direction, origin, direction, index). A lane that finishes stores its result and goes on with
// the parked ray after five LDS reads ("switch" round: no global load, so nothing to wait for). The expensive part,
// loading a ray and setting it up in f64, runs for all lanes that have used up their parked ray at once ("prep"
// round): it is the old refill round, but it runs when half of the lanes have SPENT their spare ray, not when half
// of them are idle. Finishing and refilling are decoupled.
//
// The LDS for the parked ray comes from one-word stack entries, i.e. from the compact 20-bit child references of
// rt_walk2_device.hpp: 15 entries x 4 B + 24 B (position of the closest hit) + 40 B = 124 B per lane, five blocks
// per CU as before; scenes without compact references, or with more than 14 wide levels, keep traverse_fast.
// The rounds themselves are traverse_fast's code on the same records in the same f64 arithmetic: same bits.
#pragma once

#include "rt_walk2_device.hpp"

namespace rtd {
namespace {

// bits of a lane's flag word; P_N_* describe the parked ray
constexpr uint32_t P_ACTIVE = 1u, P_PRE_OK = 2u, P_HAVE = 4u, P_OD = 8u, P_N_PRE_OK = 16u, P_N_OD = 32u, P_N_PRESENT = 0x80000000u;

// LDS of a block: [3][TBLOCK] doubles (position of the closest hit), [5][TBLOCK] 8-byte words (the parked ray),
// [stack_depth3][TBLOCK] stack words: 24 B + 40 B + 15 x 4 B = 124 B per lane = 31 KB per block, five blocks per CU
constexpr uint32_t WALK3_MAX_STACK = 15;
inline size_t stack3_bytes(const Scene &sc) {
    return (size_t)TBLOCK * (3 * sizeof(double) + 5 * sizeof(uint2) + (size_t)sc.stack_depth3 * sizeof(uint32_t));
}

// vote_switch: lanes with a finished (or no) ray and a parked one before a switch round runs; vote_prep: lanes
// without a parked ray before a prep round runs; vote_stuck: lanes that have nothing to do and nothing parked.
// NOTE: every exit of the main loop hangs on a value that is the same in all lanes.
template <class IO>
__device__ __forceinline__ void traverse_fast3(const Scene &sc, const IO &io, RaySupply supply, uint32_t *lds,
                                               const uint32_t vote_exact, const uint32_t vote_leaf,
                                               const uint32_t vote_switch, const uint32_t vote_prep,
                                               const uint32_t vote_stuck) {
    double *const best_pos = reinterpret_cast<double *>(lds) + threadIdx.x;               // [k * TBLOCK], k < 3
    uint2 *const parked = reinterpret_cast<uint2 *>(lds + 6 * TBLOCK) + threadIdx.x;       // [k * TBLOCK], k < 5
    uint32_t *const stk = lds + 16 * TBLOCK + threadIdx.x;                                  // [e * TBLOCK]
    const QNode *const nodes = sc.qnodes_c;
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, ix = 0, iy = 0, iz = 0, oix = 0, oiy = 0, oiz = 0, tlimit = 0;
    uint32_t cur = C_NONE, cand = 0, best_slot = RT_MISS, ri = 0, flags = 0;
    int sp = 0;
    double best_dist = 0.0;
    uint32_t n_need = 64;     // lanes without a parked ray (the same in every lane)
    uint32_t supply_done = 0; // the ray supply has run out (the same in every lane)
    auto uniform = [](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
#ifdef RT_COUNT_VISITS
    uint32_t n_steps = 0, n_screened = 0, n_tested = 0, n_done = 0;
    // rounds and lanes: [4..5] exact, [6..7] leaf, [8..9] node iterations, [10..11] prep rounds and rays handed
    // out, [96..97] switch rounds; census of the lanes while a round runs: [100 + 2k + j] as in traverse_fast
    auto tally = [&](int k, bool lane_takes_part) {
        unsigned long long m = lane_mask(lane_takes_part);
        const bool c_idle = cur == C_NONE;
        const bool c_exact = cand != 0, c_node = !c_idle && !(cur & C_LEAF), c_leaf = !c_idle && !c_exact && !c_node;
        const unsigned long long mi = lane_mask(c_idle), mn = lane_mask(c_node), ml = lane_mask(c_leaf), me = lane_mask(c_exact);
        if (lane_rank(~0ull) == 0) {
            atomicAdd(sc.visits + k, 1ull);
            atomicAdd(sc.visits + k + 1, (unsigned long long)__popcll(m));
            if (k < 10) {
                atomicAdd(sc.visits + 100 + 2 * k + 0, (unsigned long long)__popcll(mi));
                atomicAdd(sc.visits + 100 + 2 * k + 1, (unsigned long long)__popcll(mn));
                atomicAdd(sc.visits + 100 + 2 * k + 2, (unsigned long long)__popcll(ml));
                atomicAdd(sc.visits + 100 + 2 * k + 3, (unsigned long long)__popcll(me));
            }
        }
    };
#endif

    // an entry is (entry distance, top 12 bits of the float, rounded toward zero) | reference: a stored
    // distance never exceeds the true one, so a pop can only keep too much
    auto pop = [&]() -> uint32_t {
        while (sp > 0) {
            sp -= 1;
            const uint32_t e = stk[sp * TBLOCK];
            if (__uint_as_float(e & C_T_MASK) <= tlimit) return e & C_REF_MASK;
        }
        return C_NONE;
    };

    for (;;) {
        // cur == C_NONE: the lane has no ray in work (none at all, or a finished one); cand != 0 only with a leaf in cur
        {
            // ---- switch: lanes at rest store their result and go on with their parked ray ----
            // (votes are taken on plain comparisons and combined as 64-bit masks: a vote on a combination of
            // predicates costs two vector instructions more; every lane of the wavefront is alive here)
            const bool resting = cur == C_NONE;
            const unsigned long long m_rest = lane_mask(resting), m_parked = lane_mask((int32_t)flags < 0); // P_N_PRESENT is the sign bit
            const uint32_t n_switch = __popcll(m_rest & m_parked);
            if (n_switch >= vote_switch || (~m_rest == 0ull && (m_parked | lane_mask((flags & P_ACTIVE) != 0u)) != 0ull)) {
#ifdef RT_COUNT_VISITS
                tally(96, resting && (int32_t)flags < 0);
#endif
                if (resting) {
                    if (flags & P_ACTIVE) { // a finished ray
                        if (flags & P_OD) {
                            io.store_flag(ri); // k_rewalk_* repeats this ray in reference order
                        } else {
                            const bool have = (flags & P_HAVE) != 0u;
                            Closest best = {have, have ? best_slot : RT_MISS, have ? best_dist : 0.0, {0.0, 0.0, 0.0}};
                            if (have) best.pos = V3{best_pos[0], best_pos[TBLOCK], best_pos[2 * TBLOCK]};
                            io.store(sc, ri, best);
                        }
                        flags &= ~(P_ACTIVE | P_PRE_OK | P_HAVE | P_OD);
#ifdef RT_COUNT_VISITS
                        n_done++;
#endif
                    }
                    if ((int32_t)flags < 0) {
                        const uint2 w0 = parked[0], w1 = parked[TBLOCK], w2 = parked[2 * TBLOCK], w3 = parked[3 * TBLOCK],
                                    w4 = parked[4 * TBLOCK];
                        ix = __uint_as_float(w0.x); iy = __uint_as_float(w0.y); iz = __uint_as_float(w1.x);
                        ox = __uint_as_float(w1.y); oy = __uint_as_float(w2.x); oz = __uint_as_float(w2.y);
                        dx = __uint_as_float(w3.x); dy = __uint_as_float(w3.y); dz = __uint_as_float(w4.x);
                        ri = w4.y;
                        oix = ox * ix; oiy = oy * iy; oiz = oz * iz; // make_fray's own products
                        tlimit = __builtin_inff();
                        best_slot = RT_MISS;
                        best_dist = 0.0;
                        sp = 0;
                        cand = 0;
                        cur = (flags & P_N_OD) ? C_NONE : sc.root_ref_c; // not a unit direction: for the re-walk
                        flags = P_ACTIVE | ((flags & P_N_PRE_OK) ? P_PRE_OK : 0u) | ((flags & P_N_OD) ? P_OD : 0u);
                    }
                }
                n_need = uniform(n_need + n_switch);
            }
        }
        const bool is_exact = cand != 0;
        const bool at_rest = cur == C_NONE;
        const bool leaf_bit = cur >= C_LEAF; // references are 20 bits wide; C_NONE has the bit too
        const unsigned long long m_exact = lane_mask(is_exact), m_at_rest = lane_mask(at_rest), m_leaf_bit = lane_mask(leaf_bit);
        const unsigned long long m_leaf = m_leaf_bit & ~m_at_rest & ~m_exact;
        const bool is_node = !leaf_bit;
        const bool is_leaf = leaf_bit && !at_rest && !is_exact;
        const uint32_t n_exact = __popcll(m_exact);
        const uint32_t n_leaf = __popcll(m_leaf);
        const uint32_t n_node = __popcll(~m_leaf_bit);
        const uint32_t work = n_exact + n_leaf + n_node;
        // lanes that have nothing to do and nothing parked
        const uint32_t n_stuck = __popcll(m_at_rest & ~lane_mask((int32_t)flags < 0));

        if (work == 0 || (!supply_done && (n_need >= vote_prep || n_stuck >= vote_stuck))) {
            // ---- prep: a new parked ray for every lane that has none ----
            const uint64_t left = supply.available();
            const uint32_t avail = uniform(left > 64u ? 64u : (uint32_t)left);
            if (avail == 0) supply_done = 1u;
            const bool need = (int32_t)flags >= 0;
            const unsigned long long need_mask = lane_mask(need);
            if (avail != 0) {
                const uint32_t rank = lane_rank(need_mask);
                const bool take = need && rank < avail;
#ifdef RT_COUNT_VISITS
                tally(10, take);
#endif
                if (take) {
                    const uint32_t ri_new = (uint32_t)supply.next + rank;
                    V3 o_new, dir_new;
                    io.load(ri_new, o_new, dir_new);
                    const FRay fr = make_fray(sc, o_new, dir_new);
                    parked[0] = uint2{__float_as_uint(fr.ix), __float_as_uint(fr.iy)};
                    parked[TBLOCK] = uint2{__float_as_uint(fr.iz), __float_as_uint(fr.ox)};
                    parked[2 * TBLOCK] = uint2{__float_as_uint(fr.oy), __float_as_uint(fr.oz)};
                    parked[3 * TBLOCK] = uint2{__float_as_uint(fr.dx), __float_as_uint(fr.dy)};
                    parked[4 * TBLOCK] = uint2{__float_as_uint(fr.dz), ri_new};
                    flags = (flags & ~(P_N_PRE_OK | P_N_OD)) | P_N_PRESENT | (fr.pre_ok ? P_N_PRE_OK : 0u) | (fr.unit ? 0u : P_N_OD);
                }
                const uint32_t wanted = (uint32_t)__popcll(need_mask);
                const uint32_t taken = wanted < avail ? wanted : avail;
                supply.next += taken;
                n_need = uniform(wanted - taken);
            } else if ((lane_mask((flags & P_ACTIVE) != 0u) | lane_mask((int32_t)flags < 0)) == 0ull) { // nothing walked or parked
#ifdef RT_COUNT_VISITS
                atomicAdd(sc.visits + 0, (unsigned long long)n_done);
                atomicAdd(sc.visits + 1, (unsigned long long)n_steps);
                atomicAdd(sc.visits + 2, (unsigned long long)n_screened);
                atomicAdd(sc.visits + 3, (unsigned long long)n_tested);
#endif
                break;
            }
        } else if (n_exact >= vote_exact || (n_exact != 0 && n_leaf == 0 && n_node == 0)) {
            // ---- exact f64 test of one queued primitive per lane -------------------
#ifdef RT_COUNT_VISITS
            tally(4, is_exact);
#endif
            if (is_exact) {
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
                if (od) flags |= P_OD;
                if (is_hit) {
                    const V3 pos = hit_position(o, dir, step);
                    const double dd = mag(sub(pos, o));
                    // strict '<' keeps the first hit in reference visiting order; reference slots
                    // are numbered in that order, so an exact tie goes to the smaller one
                    if (!(flags & P_HAVE) || dd < best_dist ||
                        (dd == best_dist && sc.ref_of_fast[slot] < sc.ref_of_fast[best_slot])) {
                        flags |= P_HAVE;
                        best_slot = slot;
                        best_dist = dd;
                        best_pos[0] = pos.x; best_pos[TBLOCK] = pos.y; best_pos[2 * TBLOCK] = pos.z;
                        // the f32 ray starts `shift` along the ray (make_fray; 0 for a ray the screen may be used on)
                        double shift = 0.0;
                        if (!(flags & P_PRE_OK)) {
                            const V3 rc = {sc.root_c[0] - o.x, sc.root_c[1] - o.y, sc.root_c[2] - o.z};
                            shift = dot(rc, dir) - 2.0 * sc.root_r;
                            shift = (shift > 0.0) ? shift : 0.0;
                        }
                        tlimit = prune_limit(sc, dd, shift);
                    }
                }
                if (cand == 0) cur = pop();
            }
        } else if (n_leaf >= vote_leaf || (n_leaf != 0 && n_node == 0)) {
            // ---- leaf: f32 screen of its primitives ---------------------------------
#ifdef RT_COUNT_VISITS
            tally(6, is_leaf);
#endif
            if (is_leaf) {
                const uint32_t sb = cur & C_SLOT_MASK;
                const uint32_t cnt = ((cur >> C_CNT_SHIFT) & 7u) + 1u;
                uint32_t c = (1u << cnt) - 1u;
#ifdef RT_COUNT_VISITS
                n_screened += cnt;
#endif
                if (flags & P_PRE_OK) {
                    const uint32_t first = prefilter_pass2(sc, reinterpret_cast<const float4 *>(sc.pre2 + sb), ox, oy, oz, dx, dy, dz);
                    c = cnt > 1u ? first : first & 1u;
                    if (cnt > 2u)
                        for (uint32_t k = 2; k < cnt; k += 2) {
                            const uint32_t two = prefilter_pass2(sc, reinterpret_cast<const float4 *>(sc.pre2 + (sb + k)), ox,
                                                                 oy, oz, dx, dy, dz);
                            c |= (k + 1 < cnt ? two : two & 1u) << k;
                        }
                }
                cand = c;
                if (cand == 0) cur = pop();
            }
        } else {
            // ---- node steps (see traverse_fast: the same step on the compact references) ----
            const uint32_t cap = sc.stack_depth3;
            const uint32_t keep = n_node - (n_node * sc.node_drop >> 4);
            bool node_now = is_node;
            for (uint32_t it = 0; it < sc.node_iters; it++) {
#ifdef RT_COUNT_VISITS
                tally(8, node_now);
#endif
                if (node_now) {
                    const bool grp = (cur & C_GROUP) != 0u;
                    const uint32_t idx = cur & C_IDX_MASK;
                    const uint32_t need = (cur >> C_NEED_SHIFT) & 15u; // meaningless for a group entry
#ifdef RT_COUNT_VISITS
                    n_steps++;
#endif
                    const NodeHits h = node_hits(nodes + idx, ix, iy, iz, oix, oiy, oiz, tlimit, grp,
                                                 (cur >> C_MASK_SHIFT) & 0xFu, C_NONE);
                    if (h.k1 != NO_CHILD) { // more than one child: the others wait on the stack
                        if (!grp && (uint32_t)sp + 3u + need <= cap + 1u) {
                            if (h.k3 != NO_CHILD) { // farthest first, so that the nearest is popped first
                                stk[sp * TBLOCK] = (h.k3 & C_T_MASK) | h.r3;
                                sp += 1;
                            }
                            if (h.k2 != NO_CHILD) {
                                stk[sp * TBLOCK] = (h.k2 & C_T_MASK) | h.r2;
                                sp += 1;
                            }
                            stk[sp * TBLOCK] = (h.k1 & C_T_MASK) | h.r1;
                            sp += 1;
                        } else {
                            const uint32_t rest = (1u << (h.k1 & 3u)) | (h.k2 != NO_CHILD ? 1u << (h.k2 & 3u) : 0u) |
                                                  (h.k3 != NO_CHILD ? 1u << (h.k3 & 3u) : 0u);
                            stk[sp * TBLOCK] = (h.k1 & C_T_MASK) | C_GROUP | (rest << C_MASK_SHIFT) | idx;
                            sp += 1;
                        }
                    }
                    cur = (h.k0 != NO_CHILD) ? h.r0 : pop();
                    node_now = cur < C_LEAF;
                }
                if ((uint32_t)__popcll(lane_mask(node_now)) < keep) break;
            }
        }
    }
}

} // namespace
} // namespace rtd
