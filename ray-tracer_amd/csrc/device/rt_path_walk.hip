// rt_path_walk.hip — the fast walk over the live paths of one bounce (k_traverse_paths_fast), in a translation
// unit of its own: it is compiled with -mllvm -amdgpu-sched-strategy=max-memory-clause, which makes this kernel
// 4 % faster and the batch kernel (same source, rt_kernels.hip) 6 % slower (measured on MI355X, ROCm 7.2).
#include "rt_walk2_device.hpp"
#include "rt_walk3_device.hpp"

namespace rtd {
namespace {

__global__ void RT_FAST_BOUNDS k_traverse_paths_fast(Scene sc, Paths p, uint32_t depth,
                                                                 uint32_t rays_per_wave, unsigned long long *ticket) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    uint32_t wave = blockIdx.x * (TBLOCK / 64) + (threadIdx.x >> 6);
    RaySupplyT<uint32_t> supply; // a pass holds fewer than 2^31 paths
    supply.init(wave, gridDim.x * (TBLOCK / 64), rays_per_wave, p.counters[depth], ticket);
    PathIO io = path_io(p, depth);
    FastLds l = fast_lds(sc, lds_stack);
    traverse_fast<uint32_t>(sc, io, supply, l.sref, l.st, l.pos, sc.vote_exact_paths, sc.vote_leaf_paths,
                            sc.vote_refill_paths); // a pass holds fewer than 2^32 paths
}

__global__ void RT_FAST2_BOUNDS k_traverse_paths_fast2(Scene sc, Paths p, uint32_t depth, uint32_t rays_per_wave,
                                                       unsigned long long *ticket) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    uint64_t wave = (uint64_t)blockIdx.x * (TBLOCK / 64) + (threadIdx.x >> 6);
    RaySupply supply;
    supply.init(wave, (uint64_t)gridDim.x * (TBLOCK / 64), rays_per_wave, p.counters[depth], ticket);
    PathIO io = path_io(p, depth);
    traverse_fast2(sc, io, supply, lds_stack, sc.vote2_exact_paths, sc.vote2_leaf_paths, sc.vote2_refill_paths);
}

__global__ void RT_FAST_BOUNDS k_traverse_paths_fast3(Scene sc, Paths p, uint32_t depth, uint32_t rays_per_wave,
                                                      unsigned long long *ticket) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    // readfirstlane: the compiler then knows that the ray supply, and with it every exit of the walk's loop, is the same in all lanes
    uint64_t wave = (uint64_t)blockIdx.x * (TBLOCK / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    RaySupply supply;
    supply.init(wave, (uint64_t)gridDim.x * (TBLOCK / 64), rays_per_wave, p.counters[depth], ticket);
    PathIO io = path_io(p, depth);
    traverse_fast3(sc, io, supply, lds_stack, sc.vote_exact_paths, sc.vote_leaf_paths, sc.vote3_switch_paths,
                   sc.vote3_prep_paths, sc.vote3_stuck_paths);
}

} // namespace

void launch_path_walk_fast(const Scene &sc, const Paths &p, uint32_t depth, uint64_t expected, unsigned long long *ticket,
                           hipStream_t stream, uint32_t walk_blocks_per_cu) {
    if (ticket) (void)hipMemsetAsync(ticket, 0, sizeof *ticket, stream);
    if (sc.walk3 & 2u) { // a pass holds fewer than 2^32 paths
        allow_large_lds(k_traverse_paths_fast3);
        uint64_t waves = resident_waves(k_traverse_paths_fast3, stack3_bytes(sc), walk_blocks_per_cu);
        uint32_t rpw = rays_per_wave_for(expected, waves);
        hipLaunchKernelGGL(k_traverse_paths_fast3, dim3(blocks_for_waves(expected, rpw, waves)), dim3(TBLOCK),
                           stack3_bytes(sc), stream, sc, p, depth, rpw, ticket);
        return;
    }
    if ((sc.two_rays & 2u) && ticket) { // a pass holds fewer than 2^32 paths
        uint64_t waves = resident_waves(k_traverse_paths_fast2, stack2_bytes(sc), walk_blocks_per_cu);
        uint32_t rpw = rays_per_wave_for(expected, waves);
        hipLaunchKernelGGL(k_traverse_paths_fast2, dim3(blocks_for_waves(expected, rpw, waves)), dim3(TBLOCK),
                           stack2_bytes(sc), stream, sc, p, depth, rpw, ticket);
        return;
    }
    allow_large_lds(k_traverse_paths_fast);
    uint64_t waves = resident_waves(k_traverse_paths_fast, stack_bytes(sc), walk_blocks_per_cu);
    uint32_t rpw = rays_per_wave_for(expected, waves);
    hipLaunchKernelGGL(k_traverse_paths_fast, dim3(blocks_for_waves(expected, rpw, waves)), dim3(TBLOCK), stack_bytes(sc),
                       stream, sc, p, depth, rpw, ticket);
}

} // namespace rtd
