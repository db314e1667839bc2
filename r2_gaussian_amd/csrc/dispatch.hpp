// dispatch.hpp -- which chain a forward takes, decided in ONE place (round 6).
//
// Every forward entry point has several chains (rounds 3-6 added them one by one: a fast chain for the common case next to the
// general chain that serves everything).  The static part of each rule -- what (P, views, grid, debug, switches, device) admits -- is
// the two functions below; what only the call itself can find out (no prediction yet, a list too long, more survivors than the
// small path holds) is counted where it is found, under the same names.  r2_path_stats() returns every counter: per operator the
// calls each chain served and WHY the others went to the general chain.
//
//   operator           chain        serves                                            static rule (this file)              dynamic hand-overs (counted at their site)
//   rasterizer fwd     tile-first   1..V stacked views, <= 4096 tiles in all          raster_forward_choice()              no prediction for this (P, V, detector) yet; no counter block
//                      general      everything (first call of a size, debug, > 4096 tiles, >= 2^24 view instances)
//   voxelizer fwd      small-grid   <= 64 tiles, <= 8 per axis (the 32^3 TV patch)    voxel_forward_choice()               more than 8192 survivors / state beyond the temp's capacity
//                      stick-first  65 .. 32768 tiles (64^3 .. the 256^3 query)       voxel_forward_choice()               a list beyond the long-list rule (r2_voxel_sticks_limits), remembered per (P, grid)
//                      general      everything (debug, x-slabs of <= 64 tiles, larger grids)
//   render kernels     one-wave forward / four-wave forward (ids >= 2^28, R2_FWD_WAVE=0) / debug (n_contrib)
// Results never depend on the chain: point_list, ranges, images / volumes and gradients are identical (tests/test_*_gpu.py compare them).
#pragma once
#include "r2_common.hpp"

namespace r2 {

enum PathStat {
    // rasterizer forward
    PS_RAS_TILE_FIRST = 0,          // calls served by the tile-first chain
    PS_RAS_GENERAL_DEBUG,           // general chain because: debug mode
    PS_RAS_GENERAL_SWITCHED_OFF,    //   r2_tile_first_control(0) / R2_TILE_FIRST=0
    PS_RAS_GENERAL_GRID,            //   more than 4096 (stacked) tiles, or a grid beyond 256 tiles along an axis
    PS_RAS_GENERAL_INSTANCES,       //   2^24 view instances or more (ids share a word with the block mask), or offsets beyond 32 bits
    PS_RAS_GENERAL_DEVICE_LDS,      //   the device cannot give the sort kernel its 78 KB of LDS
    PS_RAS_GENERAL_NO_PREDICTION,   //   first call of this (P, V, detector) on this thread and no call on the same detector to seed from
    PS_RAS_GENERAL_NO_WORKSPACE,    //   the per-(thread, device, stream) counter block could not be had
    PS_RAS_EVENT_SEEDED,            // events of the tile-first chain: prediction seeded from another Gaussian count
    PS_RAS_EVENT_SECOND_PASS,       //   prediction short: chain enqueued again with the exact size
    PS_RAS_EVENT_DEPTH_SLABS,       //   lists cut into depth slabs (long lists expected)
    PS_RAS_EVENT_DEFERRED,          //   returned a token instead of waiting (r2_defer_count_control)
    // voxelizer forward
    PS_VOX_SMALL_GRID,              // calls served by the small-grid path
    PS_VOX_STICK_FIRST,             // ... by the stick-first chain
    PS_VOX_GENERAL_DEBUG,           // general chain because: debug mode
    PS_VOX_GENERAL_SWITCHED_OFF,    //   the chain that would serve the grid is switched off (R2_VOXEL_SMALL=0, r2_voxel_sticks_control(0))
    PS_VOX_GENERAL_GRID,            //   more than 32 768 x 2^k tiles / 65535 tiles along an axis / an axis beyond 8 tiles on a <= 64-tile grid
    PS_VOX_GENERAL_SLAB,            //   an x-slab call of <= 64 tiles (the small path rebuilds tile cubes without the slab's clip)
    PS_VOX_GENERAL_INSTANCES,       //   P beyond the id bits of the chain (2^20 small-grid, 2^29 stick-first)
    PS_VOX_GENERAL_DEVICE_LDS,      //   the device cannot give the sort kernels their LDS
    PS_VOX_GENERAL_NO_WORKSPACE,    //   no counter block
    PS_VOX_GENERAL_REMEMBERED,      //   this thread has seen this (P, grid) hand over before (long lists), within the last VS_NOTE_RETRY calls
    PS_VOX_GENERAL_LONG_LISTS,      //   handed over after the scan: a list beyond the long-list rule
    PS_VOX_GENERAL_SMALL_OVERFLOW,  //   handed over after the preprocess: more survivors / instances than the small path holds
    PS_COUNT
};
void path_count(PathStat s);

struct RasterChoice { bool tile_first; PathStat why; };
// enabled / lds_ok: the chain's switch and the device's LDS opt-in (raster_tilefirst.hip knows both); wgs = producer workgroups
RasterChoice raster_forward_choice(size_t P, size_t V, int width, int height, bool debug, bool enabled, bool lds_ok, size_t wgs);

enum VoxelChain { VOX_CHAIN_GENERAL = 0, VOX_CHAIN_SMALL, VOX_CHAIN_STICKS };
struct VoxelChoice { VoxelChain chain; PathStat why; uint32_t stick_shift; };
struct VoxelGrid;
VoxelChoice voxel_forward_choice(const VoxelGrid &v, size_t P, bool debug, bool small_on, bool small_lds_ok, bool sticks_on,
                                 bool sticks_lds_ok);

// limits the rules are made of (the kernels' own constants, gathered here)
constexpr size_t DISPATCH_RAS_MAX_TILES = 4096, DISPATCH_RAS_MAX_AXIS = 256, DISPATCH_RAS_MAX_INSTANCES = (size_t)1 << 24;
constexpr size_t DISPATCH_VOX_SMALL_TILES = 64, DISPATCH_VOX_SMALL_AXIS = 8, DISPATCH_VOX_SMALL_P = (size_t)1 << 20;
constexpr size_t DISPATCH_VOX_STICK_LISTS = 4096, DISPATCH_VOX_STICK_P = (size_t)1 << 29;
constexpr uint32_t DISPATCH_VOX_STICK_MAX_SHIFT = 3;

}  // namespace r2
