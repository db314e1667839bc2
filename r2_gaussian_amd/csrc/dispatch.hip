// dispatch.hip -- the chain rules of dispatch.hpp and their counters (r2_path_stats / r2_path_stat_name in r2hip.h).
#include "dispatch.hpp"
#include "voxel_state.hpp"

#include <atomic>

namespace r2 {

namespace {
std::atomic<long long> g_path[PS_COUNT];
const char *const PATH_NAMES[PS_COUNT] = {
    "raster.tile_first", "raster.general.debug", "raster.general.switched_off", "raster.general.grid", "raster.general.instances",
    "raster.general.device_lds", "raster.general.no_prediction", "raster.general.no_workspace", "raster.event.seeded",
    "raster.event.second_pass", "raster.event.depth_slabs", "raster.event.deferred",
    "voxel.small_grid", "voxel.stick_first", "voxel.general.debug", "voxel.general.switched_off", "voxel.general.grid",
    "voxel.general.slab", "voxel.general.instances", "voxel.general.device_lds", "voxel.general.no_workspace",
    "voxel.general.remembered", "voxel.general.long_lists", "voxel.general.small_overflow",
};
}  // namespace

void path_count(PathStat s) { g_path[s].fetch_add(1, std::memory_order_relaxed); }

RasterChoice raster_forward_choice(size_t P, size_t V, int width, int height, bool debug, bool enabled, bool lds_ok, size_t wgs)
{
    const size_t gx = (size_t)(width + TILE2D - 1) / TILE2D, gy = (size_t)(height + TILE2D - 1) / TILE2D, T = gx * gy * V;
    if (debug) return {false, PS_RAS_GENERAL_DEBUG};                     // n_contrib and the per-stage checks live in the general chain
    if (!enabled) return {false, PS_RAS_GENERAL_SWITCHED_OFF};
    // one LDS histogram over the lists; rectangles packed into bytes of the stacked grid
    if (T > DISPATCH_RAS_MAX_TILES || gx > DISPATCH_RAS_MAX_AXIS || gy * V > DISPATCH_RAS_MAX_AXIS) return {false, PS_RAS_GENERAL_GRID};
    // ids share a word with the block mask; per-(workgroup, list) offsets are 32-bit
    if (P * V >= DISPATCH_RAS_MAX_INSTANCES || wgs * T > ((size_t)1 << 25)) return {false, PS_RAS_GENERAL_INSTANCES};
    if (!lds_ok) return {false, PS_RAS_GENERAL_DEVICE_LDS};
    return {true, PS_RAS_TILE_FIRST};
}

VoxelChoice voxel_forward_choice(const VoxelGrid &v, size_t P, bool debug, bool small_on, bool small_lds_ok, bool sticks_on,
                                 bool sticks_lds_ok)
{
    const size_t T = (size_t)v.gx * v.gy * v.gz;
    if (debug) return {VOX_CHAIN_GENERAL, PS_VOX_GENERAL_DEBUG, 0u};
    if (T <= DISPATCH_VOX_SMALL_TILES) {   // patches: survivors only
        if (v.is_slab()) return {VOX_CHAIN_GENERAL, PS_VOX_GENERAL_SLAB, 0u};
        if ((size_t)v.gx > DISPATCH_VOX_SMALL_AXIS || (size_t)v.gy > DISPATCH_VOX_SMALL_AXIS || (size_t)v.gz > DISPATCH_VOX_SMALL_AXIS)
            return {VOX_CHAIN_GENERAL, PS_VOX_GENERAL_GRID, 0u};
        // {workgroups done : 12 | survivors : 20 | rows : 32} in one 64-bit atomic
        if (P >= DISPATCH_VOX_SMALL_P) return {VOX_CHAIN_GENERAL, PS_VOX_GENERAL_INSTANCES, 0u};
        if (!small_on) return {VOX_CHAIN_GENERAL, PS_VOX_GENERAL_SWITCHED_OFF, 0u};
        if (!small_lds_ok) return {VOX_CHAIN_GENERAL, PS_VOX_GENERAL_DEVICE_LDS, 0u};
        return {VOX_CHAIN_SMALL, PS_VOX_SMALL_GRID, 0u};
    }
    // sticks of 2^shift consecutive tile ids: at most 4096 lists
    uint32_t sh = 0;
    while (sh <= DISPATCH_VOX_STICK_MAX_SHIFT && ((T + ((size_t)1 << sh) - 1) >> sh) > DISPATCH_VOX_STICK_LISTS) ++sh;
    if (sh > DISPATCH_VOX_STICK_MAX_SHIFT || v.gx > 65535 || v.gy > 65535 || v.gz > 65535) return {VOX_CHAIN_GENERAL, PS_VOX_GENERAL_GRID, 0u};
    if (P >= DISPATCH_VOX_STICK_P) return {VOX_CHAIN_GENERAL, PS_VOX_GENERAL_INSTANCES, 0u};   // ids share a word with the tile-in-stick bits
    if (!sticks_on) return {VOX_CHAIN_GENERAL, PS_VOX_GENERAL_SWITCHED_OFF, 0u};
    if (!sticks_lds_ok) return {VOX_CHAIN_GENERAL, PS_VOX_GENERAL_DEVICE_LDS, 0u};
    return {VOX_CHAIN_STICKS, PS_VOX_STICK_FIRST, sh};
}

}  // namespace r2

extern "C" int r2_path_stat_count(void) { return (int)r2::PS_COUNT; }
extern "C" const char *r2_path_stat_name(int i) { return (i >= 0 && i < (int)r2::PS_COUNT) ? r2::PATH_NAMES[i] : nullptr; }
extern "C" int r2_path_stats(long long *out, int n, int reset)
{
    for (int i = 0; i < (int)r2::PS_COUNT; ++i) {
        if (out && i < n) out[i] = r2::g_path[i].load(std::memory_order_relaxed);
        if (reset) r2::g_path[i].store(0, std::memory_order_relaxed);
    }
    return (int)r2::PS_COUNT;
}
