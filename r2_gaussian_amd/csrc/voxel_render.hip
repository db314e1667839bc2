// voxel_render.hip -- per-tile evaluation of the Gaussian mixture on the voxel grid and its backward.
//
// Reference: renderCUDA forward VOX/forward.cu:183-315, renderCUDA backward VOX/backward.cu:216-374.
// VALU/exp-bound (512 voxel-Gaussian pairs per 48 bytes gathered): FMA contraction ON, tolerance-checked.
// The reference evaluates `power` partly in double (quirk Q6); here it is a float FMA chain on the
// log2e-pre-scaled inverse covariance -- the difference is ~1e-7 relative, inside the 1e-4 budget.
//
// Forward : a tile's list is cut into work items of VOX_CHUNK instances; one workgroup = one work item =
//           8 waves over the 8x8x8 tile.  Lanes are mapped z-fastest (lane = y*8+z, wave = x) so the volume is
//           written in 32-byte runs of [nx,ny,nz] (the reference's x-fastest thread order strides by ny*nz
//           floats between lanes).  Records (48 bytes packed) are staged through LDS in 512-record batches;
//           partial sums per work item are added in list order by a second kernel (deterministic volume).
// Backward: item-parallel as in raster_render.hip: one LANE owns one x-slab of one (tile, Gaussian) instance; dL/dvol of
//           the wave's tiles is staged in LDS.  The 10 gradient sums of the reference are linear in 10 moments of
//           w = G*dL: sum w, sum w d_i, sum w d_i d_j.  No atomics: each instance adds its slabs' rows in a fixed order
//           and stores its moment row at its emission index, the geometry backward reduces each Gaussian's contiguous
//           run in a fixed order (bit-reproducible gradients).
#include "voxel_state.hpp"

// experiment builds (-DR2_EXP_TS): stamps inside the item workgroups of the forward kernel, read by scripts/cbench (phase durations of
// the first 2047 work items by item length: profiles/r06_voxel_render_stamps.txt).  15 holds the item's length, not a time.
R2_TS_DEFINE(vrender)
#ifdef R2_EXP_TS
#define VR_TS(ph) do { if (hb < 2047u * 2u && (hb & 1u) == 0u && threadIdx.x == 0) r2::g_ts_vrender[ph][hb >> 1] = wall_clock64(); } while (0)
#define VR_TSV(ph, val) do { if (hb < 2047u * 2u && (hb & 1u) == 0u && threadIdx.x == 0) r2::g_ts_vrender[ph][hb >> 1] = (unsigned long long)(val); } while (0)
#else
#define VR_TS(ph)
#define VR_TSV(ph, val)
#endif

namespace r2 {

// ------------------------------------------------------------------------------------------------ forward
// Production kernel, item-parallel like the rasterizer's (raster_render.hip): a workgroup = one work item (<= VOX_CHUNK
// instances of one tile list), its 8 waves own the 8 x-slabs (1 x 8 x 8 voxels) of the tile and never synchronise.
// One LANE owns one list entry and accumulates the slab's 64 voxels in registers; the workgroup stages 512 entries at a
// time in LDS and every wave compacts those whose alpha >= 1e-6 bounding box touches its slab; along a z row the
// Gaussian is walked with the recurrence G(c+1) = G(c) r(c), r(c+1) = r(c) exp2(2 F2), re-anchored every 4 voxels (two
// v_exp_f32 per 4 voxels instead of four); a 64x64 transpose-reduction leaves voxel (y, z) of the slab in lane y*8+z.
constexpr int VFWD_BATCH = 256;   // list entries staged per round: one per thread of the workgroup
constexpr int VFWD_MIN_STEP = 25;  // tile lists shorter than this go to the one-wave-per-tile kernel (voxel-parallel)
// Within a work item, a wave's remainder of fewer live entries than VFWD_STEP_MIN is evaluated voxel-parallel instead of as a
// (partly filled) lane-per-entry step.  Round 4: the remainder loop is a chain of dependent LDS reads that costs about as much
// as 80 instructions per entry (fitted from SQ_INSTS_VALU and time of two builds), a step ~420 whatever its fill: break-even
// at 5-6 entries, not at the 25 this used until round 4 (profiles/r04f_voxel_steps.txt).
#ifndef R2_VFWD_STEP_MIN
#define R2_VFWD_STEP_MIN 6
#endif
constexpr int VFWD_STEP_MIN = R2_VFWD_STEP_MIN;

__device__ __forceinline__ bool slab_live(float px, float py, float pz, float4 h, float kz, float xc, float y0, float z0)
{
    // slab = voxel centres x = xc, y in [y0+0.5, y0+7.5], z in [z0+0.5, z0+7.5]; h = {hx, hyc, hzc, ky} (VoxelGeom::ext).
    // At the offset dx the cut-off ellipsoid's cross-section is an ellipse around (py - ky dx, pz - kz dx) inside the box
    // +-(hyc, hzc) * sqrt(1 - (dx/hx)^2): the slab is live if that box touches its 8x8 voxel centres.  Conservative: hx, hyc,
    // hzc are padded (0.4 % + 0.05 voxel), so the root is >= 0.09 wherever the true ellipsoid reaches.  hx = +inf (no
    // culling): dx/hx = 0, hyc = hzc = +inf; hx = -inf (nothing passes): the first comparison fails.
    const float dx = px - xc;
    const float u = dx * __builtin_amdgcn_rcpf(h.x);
    const float t = __builtin_amdgcn_sqrtf(fmaxf(1.0f - u * u, 0.0f));   // raw v_sqrt_f32 (1 ulp): sqrtf costs ~20 instructions here
    const float cy = py - h.w * dx, cz = pz - kz * dx;
#ifdef R2_EXP_SLAB_HALF   // sensitivity experiment of profiles/r04f_voxel_steps.txt: halved cross-section (WRONG volumes)
    const float ey = 0.5f * h.y * t, ez = 0.5f * h.z * t;
#else
    const float ey = h.y * t, ez = h.z * t;
#endif
    return (fabsf(dx) <= h.x) && (cy - ey <= y0 + 7.5f) && (cy + ey >= y0 + 0.5f) && (cz - ez <= z0 + 7.5f) &&
           (cz + ez >= z0 + 0.5f);
}

// Measured and left out: a second step body that walks whole rows of 8 voxels with one recurrence for the entries that allow
// it (85 % here; three compaction queues, one tier per step): 0.666 -> 0.718 ms at 256^3 -- two ~2000-instruction bodies and
// more partly filled steps cost more than the two v_exp_f32 per row it saves.
// Measured and left out (round 4): the recurrence state of two rows in one register pair and v_pk_mul_f32 for its two
// multiplications per voxel (4 -> 3 VALU instructions per voxel): 425 -> 435 us.  Packed f32 multiplies do not issue at twice
// the scalar rate here.
__device__ __forceinline__ void vfwd_item(const float4 p, const float4 q, const float4 r, float xc, float y0, float z0,
                                          float (&acc)[64])
{
    // log2(alpha) = E(y, z) = a2 dx^2 + b2 dx dy + c2 dx dz + d2 dy^2 + e2 dy dz + f2 dz^2 + L, dy = p.y - (y0 + y + 0.5), ...
    // Round 4: a second recurrence ACROSS the rows.  Until then every row of a slab took 4 v_exp_f32 (two z segments x {start
    // value, ratio}): 32 quarter-rate instructions and their arguments per step, a quarter of its cycles.  Now per z segment:
    //   rows 4..7: g(4) = 2^E(4, s), then g(y+1) = g(y) rho(y), rho(y+1) = rho(y) kappa    rho(4) = 2^(E(5,s) - E(4,s))
    //   rows 3..0: g(3) = 2^E(3, s), then g(y-1) = g(y) rho'(y), rho'(y-1) = rho'(y) kappa  rho'(3) = 2^(E(2,s) - E(3,s))
    //   z ratio:   rt(4) = 2^(E(4,s+1) - E(4,s)), rt(y+1) = rt(y) chi, rt(y-1) = rt(y) / chi
    // with kappa = 2^(2 d2), chi = 2^e2 (one v_exp_f32 each per entry): 5 exponentials per segment instead of 16.  The clamps
    // keep the ratios finite where the Gaussian is long dead (0 * inf); needs_exact_slab3 (voxel_state.hpp) keeps out the
    // entries for which a walk could start from an underflowed value and climb back above the cut-off.
    const float dx = p.x - xc;
    const float adx2L = q.x * dx * dx + r.z;
    const float bdx = q.y * dx, cdx = q.z * dx;
    const float dz0 = p.z - (z0 + 0.5f);
    const float kf1 = r.y * (1.0f - 2.0f * dz0);
    const float rr = __builtin_amdgcn_exp2f(2.0f * r.y);
    const float kap = __builtin_amdgcn_exp2f(2.0f * q.w);
    const float chi = __builtin_amdgcn_exp2f(r.x), chii = __builtin_amdgcn_exp2f(-r.x);
    const float dy4 = p.y - (y0 + 4.5f), dy3 = p.y - (y0 + 3.5f);
    const float k0u = dy4 * (q.w * dy4 + bdx) + adx2L, k1u = r.x * dy4 + cdx;
    const float k0d = dy3 * (q.w * dy3 + bdx) + adx2L, k1d = r.x * dy3 + cdx;
    const float eu = q.w * (1.0f - 2.0f * dy4) - bdx;   // E(5, z) - E(4, z) + e2 dz
    const float ed = q.w * (1.0f + 2.0f * dy3) + bdx;   // E(2, z) - E(3, z) - e2 dz
#if defined(__AMDGCN_WAVEFRONT_SIZE) && __AMDGCN_WAVEFRONT_SIZE != 64
#error "the inline asm below assumes a 64-lane EXEC mask (wave64)"
#endif
    const unsigned long long full_exec = __builtin_amdgcn_read_exec();
    (void)full_exec;
    // one row of a segment: acc += (g >= 1e-6) ? g : 0 for VOX_RECUR_STEPS voxels.  power <= 0 (VOX/forward.cu:288) holds:
    // entries that reach this path have a positive definite conic; alpha >= 1e-6: VOX/forward.cu:293.  The EXEC-mask form
    // (compare + masked add instead of compare + select + add) took 487 -> 467 us at 256^3 in round 3.
#ifndef R2_EXP_NO_CMPX
#define R2_VFWD_ROW(ROW, SEG, G0, RT0)                                                                      \
    {                                                                                                       \
        float g_ = (G0), rt_ = (RT0);                                                                       \
        _Pragma("unroll") for (int c = 0; c < VOX_RECUR_STEPS; ++c) {                                       \
            asm volatile("v_cmpx_le_f32_e32 %[thr], %[g]\n\t"                                               \
                         "v_add_f32_e32 %[a], %[a], %[g]\n\t"                                               \
                         "s_mov_b64 exec, %[ex]"                                                            \
                         : [a] "+v"(acc[(ROW) * TILE3D + (SEG) + c])                                        \
                         : [thr] "n"(0x358637bd), [g] "v"(g_), [ex] "s"(full_exec)                          \
                         : "vcc");                                                                          \
            g_ *= rt_;                                                                                      \
            rt_ *= rr;                                                                                      \
        }                                                                                                   \
    }
#else   // the portable statement of the same arithmetic (built as libr2hip_nocmpx.so and compared in tests/test_variants_gpu.py)
#define R2_VFWD_ROW(ROW, SEG, G0, RT0)                                                                      \
    {                                                                                                       \
        float g_ = (G0), rt_ = (RT0);                                                                       \
        _Pragma("unroll") for (int c = 0; c < VOX_RECUR_STEPS; ++c) {                                       \
            acc[(ROW) * TILE3D + (SEG) + c] += (g_ >= ALPHA_MIN_3D) ? g_ : 0.f;                             \
            g_ *= rt_;                                                                                      \
            rt_ *= rr;                                                                                      \
        }                                                                                                   \
    }
#endif
#pragma unroll
    for (int seg = 0; seg < TILE3D; seg += VOX_RECUR_STEPS) {   // z segments: the z walk is re-anchored every VOX_RECUR_STEPS voxels
        const float dzs = dz0 - (float)seg;
        const float zq = r.y * dzs, ez = r.x * dzs;
        float gu = __builtin_amdgcn_exp2f(dzs * (zq + k1u) + k0u), gd = __builtin_amdgcn_exp2f(dzs * (zq + k1d) + k0d);
        float ru = __builtin_amdgcn_exp2f(fminf(eu - ez, 100.0f)), rd = __builtin_amdgcn_exp2f(fminf(ed + ez, 100.0f));
        float rtu = __builtin_amdgcn_exp2f(fminf(kf1 + (2.0f * (float)seg) * r.y - k1u, 100.0f));
        float rtd = rtu * chii;
#pragma unroll
        for (int j = 0; j <= VOX_RECUR_YSTEPS; ++j) {
            R2_VFWD_ROW(4 + j, seg, gu, rtu)
            gu *= ru; ru *= kap; rtu *= chi;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j <= VOX_RECUR_YSTEPS; ++j) {
            R2_VFWD_ROW(3 - j, seg, gd, rtd)
            gd *= rd; rd *= kap; rtd *= chii;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#undef R2_VFWD_ROW
}

// number of set bits of a ballot below this lane (v_mbcnt: no 64-bit lane mask to keep in registers)
__device__ __forceinline__ uint32_t lanes_below(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// one entry, voxel-parallel: lane = voxel y*8+z of the slab at x = xc (exact exp; ~20 instructions per entry)
__device__ __forceinline__ float vfwd_voxel_parallel(const float4 p, const float4 q, const float4 r, float xc, float y0, float z0, int lane)
{
    const float dx = p.x - xc, dy = p.y - (y0 + (float)(lane >> 3) + 0.5f), dz = p.z - (z0 + (float)(lane & 7) + 0.5f);
    const float pl = dx * (q.x * dx + q.y * dy + q.z * dz) + dy * (q.w * dy + r.x * dz) + ((r.y * dz) * dz + r.z);
    const float al = __builtin_amdgcn_exp2f(pl);
    return ((pl <= r.z) && (al >= ALPHA_MIN_3D)) ? al : 0.f;
}

#ifndef R2_VFWD_WGS
#define R2_VFWD_WGS 4
#endif
__device__ __forceinline__ void vfwd_item_body(
    const uint32_t hb /* half-item: work item hb / 2, x-slabs 4 (hb & 1) .. + 3 */,
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ chunk_base, const uint4 *__restrict__ work_tile,
    uint32_t T, const uint32_t *__restrict__ point_list, const float4 *__restrict__ rec, const float4 *__restrict__ ext,
    VoxelGrid v, float *__restrict__ partial, float *__restrict__ out)
{
    // two workgroups of 4 waves per work item (x-slabs 0-3 and 4-7): tile lists are short (~150 entries at 256^3), so
    // a workgroup is one dependent chain of gathers followed by 2-3 evaluation steps -- small workgroups let 4+ of them
    // overlap on a CU
    const uint32_t w = hb >> 1;
    const int half = (int)(hb & 1u);
    VR_TS(0);
    const uint4 wd = work_tile[w];   // {tile, first instance, one past the last, items of the tile}
    const uint32_t tile = wd.x, beg = wd.y, end = wd.z;
    VR_TSV(15, end - beg);
    VR_TS(1);   // the descriptor is here
    const int tx = tile % v.gx, ty = (tile / v.gx) % v.gy, tz = tile / (v.gx * v.gy);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slab = half * 4 + wave;
    // (v.ox: x offset of an x-slab call inside the full grid, 0 otherwise -- the arithmetic sees the FULL grid's voxel coordinates,
    //  only the output index is slab-local: voxel_state.hpp)
    const float xc = (float)(tx * TILE3D + v.ox + slab) + 0.5f, y0 = (float)(ty * TILE3D), z0 = (float)(tz * TILE3D);
    const float xc0 = (float)(tx * TILE3D + v.ox + half * 4) + 0.5f;   // the workgroup's first slab

    // the workgroup stages VFWD_BATCH entries at a time (one per thread); every wave then picks the entries whose
    // bounding box touches ITS x-slab
    // p, q, r (+ per wave 64 carry rows behind the batch: records of the live entries a batch left over), extents
    __shared__ float4 s0[VFWD_BATCH + 256], s1[VFWD_BATCH + 256], s2[VFWD_BATCH + 256], s3[VFWD_BATCH];
    __shared__ uint16_t sQ[4][VFWD_BATCH];
    __shared__ unsigned long long sKeep[4][4], sExact[4][4];   // [slab of the workgroup][staging wave]: ballots of the slab tests

    if (end - beg < (uint32_t)VFWD_MIN_STEP) {
        // Very short list (the median tile at 256^3 holds 8 entries): no lane-per-entry step can fill up, so ONE
        // workgroup evaluates all 8 slabs voxel-parallel (two per wave) and the second half-item exits at once.
        if (half) return;
        const int n = (int)(end - beg);
        if (tid < n) {
            const uint32_t id = point_list[beg + (uint32_t)tid];
            s0[tid] = rec[3 * id]; s1[tid] = rec[3 * id + 1]; s2[tid] = rec[3 * id + 2]; s3[tid] = ext[id];
        }
        __syncthreads();
        for (int sl = wave; sl < TILE3D; sl += 4) {
            const float xs = (float)(tx * TILE3D + v.ox + sl) + 0.5f;
            float sum = 0.f;
            for (int j = 0; j < n; ++j) {
                const float4 p = s0[j], h = s3[j];
                const float4 q = s1[j], r = s2[j];
                if (!slab_live(p.x, p.y, p.z, h, r.w, xs, y0, z0)) continue;   // wave-uniform
                const float dx = p.x - xs, dy = p.y - (y0 + (float)(lane >> 3) + 0.5f), dz = p.z - (z0 + (float)(lane & 7) + 0.5f);
                const float pl = dx * (q.x * dx + q.y * dy + q.z * dz) + dy * (q.w * dy + r.x * dz) + ((r.y * dz) * dz + r.z);
                const float al = __builtin_amdgcn_exp2f(pl);
                sum += ((pl <= r.z) && (al >= ALPHA_MIN_3D)) ? al : 0.f;
            }
            if (wd.w == 1u) {
                const int vx = tx * TILE3D + sl, vy = ty * TILE3D + (lane >> 3), vz = tz * TILE3D + (lane & 7);
                if (vx < v.nx && vy < v.ny && vz < v.nz) out[((size_t)vx * v.ny + vy) * v.nz + vz] = sum;
            } else {
                partial[(size_t)w * 512 + sl * 64 + lane] = sum;
            }
        }
        return;
    }

    float acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = 0.f;
    float tail = 0.f;   // voxel-parallel contributions: already in the final (lane = voxel) layout
    bool stepped = false;   // wave-uniform: did any lane-per-entry step run (else acc is still all zero)
    int ncarry = 0;         // wave-uniform: entries in the wave's carry rows
    const int carry0 = VFWD_BATCH + wave * 64;

    for (uint32_t base = beg;; base += VFWD_BATCH) {
        // one extra pass after the last batch flushes the carry buffer through the same step code (a second copy of the step
        // body costs registers: 126 -> 128 VGPRs + scratch)
        const bool flush = base >= end;   // workgroup-uniform
        if (!flush) {
            const uint32_t k = base + (uint32_t)tid;
            const uint32_t id = point_list[k < end ? k : beg];
            const float4 np = rec[3 * id], nq = rec[3 * id + 1], nr = rec[3 * id + 2], nh = ext[id];
            // The thread that stages an entry also tests it against the workgroup's four slabs, while the record is in its
            // registers; the ballots go to LDS and every wave later reads the four masks of ITS slab (until round 4 every wave
            // re-read each entry from LDS and tested it for its own slab; 363 -> 357 us at 256^3, a third fewer LDS reads).
            const bool valid = k < end;
            const bool exact = needs_exact_slab3(nq.w, nr.y, nr.z, nh.z);
            unsigned long long mk[4], mx[4];
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const bool live = valid && slab_live(np.x, np.y, np.z, nh, nr.w, xc0 + (float)sl, y0, z0);
                mk[sl] = __ballot(live && !exact);
                mx[sl] = __ballot(live && exact);
            }
            // (nq is not needed by the tests above: it waits in this thread's own slot of s3 -- which only the short-list path
            // uses otherwise -- instead of in four registers across the barrier.  With it in registers the kernel needed 131 of
            // the 128 VGPRs that four waves per SIMD allow and spilled three of them around this barrier: 12 bytes written
            // and read back per staged entry, 96 MB of scratch writes per 256^3 query in the round-4 counters.)
            s3[tid] = nq;
            if (base == beg) VR_TS(2); else VR_TS(8);   // ids -> records -> slab tests of the first / of the latest batch
            __syncthreads();   // the previous batch has been consumed
            s0[tid] = np; s1[tid] = s3[tid]; s2[tid] = nr;
            if (lane == 0) {
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) { sKeep[sl][wave] = mk[sl]; sExact[sl][wave] = mx[sl]; }
            }
            __syncthreads();
        }
        // (opaque copy of the flag for the code below: otherwise the compiler threads the flush pass into a second copy of
        // the step body -- 2 x 500 instructions)
        int flush_pass = __builtin_amdgcn_readfirstlane(flush ? 1 : 0);
        asm volatile("" : "+s"(flush_pass));
        if (base == beg) VR_TS(3); else if (!flush_pass) VR_TS(9);   // staged
        // compaction: entries that may use the row recurrences queue up from the front of sQ, the few that need the exact
        // evaluation (needs_exact_slab3: very thin along y or z, or no finite culling box) from the back -- they are evaluated
        // voxel-parallel, so that the lane-per-entry step is straight-line code (the exact variant of the step cost 24 VGPRs
        // = one wave per SIMD, and 50 us at 256^3 whenever a single lane of a step asked for it)
        int cnt = 0, cntx = 0;
        if (!flush_pass) {
#pragma unroll
            for (int r = 0; r < VFWD_BATCH / 64; ++r) {   // entries r*64 .. r*64+63 were staged by wave r
                const unsigned long long m = sKeep[wave][r], mxr = sExact[wave][r];
                const int e = r * 64 + lane;
                if ((m >> lane) & 1ull) sQ[wave][cnt + (int)lanes_below(m)] = (uint16_t)e;
                if ((mxr >> lane) & 1ull) sQ[wave][VFWD_BATCH - 1 - (cntx + (int)lanes_below(mxr))] = (uint16_t)e;
                cnt += __popcll(m);
                cntx += __popcll(mxr);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // Round 4: a batch's remainder of fewer than 64 live entries is not flushed as a partly filled step; its records move
        // to the wave's carry buffer and open the next batch's first step (fill of the steps at 256^3: 80 % -> 91 %).
        int head = 0;
#pragma nounroll   // (also keeps the first pass -- the only one with carry rows -- from being peeled into a second copy)
        while (ncarry + (cnt - head) >= (flush_pass ? VFWD_STEP_MIN : 64)) {
            float4 ep = make_float4(0.f, 0.f, 0.f, 0.f), eq = ep, er = make_float4(0.f, 0.f, -INFINITY, 0.f);   // idle lane
            if (head + lane - ncarry < cnt) {
                const int e = lane < ncarry ? carry0 + lane : (int)sQ[wave][head + lane - ncarry];
                ep = s0[e]; eq = s1[e]; er = s2[e];
            }
            vfwd_item(ep, eq, er, xc, y0, z0, acc);
            stepped = true;
            head += 64 - ncarry;
            ncarry = 0;
        }
        if (base == beg) VR_TS(4); else if (!flush_pass) VR_TS(10);   // the batch's steps are done
        if (flush_pass) {
            // fewer than VFWD_STEP_MIN entries left: voxel-parallel (a step would be mostly idle)
            for (int t = 0; t < ncarry; ++t) tail += vfwd_voxel_parallel(s0[carry0 + t], s1[carry0 + t], s2[carry0 + t], xc, y0, z0, lane);
            break;
        }
        {
            const int rem = cnt - head;   // < 64 - ncarry
            __builtin_amdgcn_wave_barrier();   // (the step above has read the carry rows this overwrites)
            if (lane < rem) {
                const int e = sQ[wave][head + lane];
                s0[carry0 + ncarry + lane] = s0[e]; s1[carry0 + ncarry + lane] = s1[e]; s2[carry0 + ncarry + lane] = s2[e];
            }
            ncarry += rem;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // the exact entries are evaluated voxel-parallel (lane = voxel y*8+z of the slab, entries broadcast from LDS, exact exp)
        for (int t = 0; t < cntx; ++t) {
            const int e = sQ[wave][VFWD_BATCH - 1 - t];
            tail += vfwd_voxel_parallel(s0[e], s1[e], s2[e], xc, y0, z0, lane);
        }
    }

    VR_TS(5);   // the flush pass is done
    // 64x64 transpose-reduction (see raster_render.hip): acc[0] ends up as the slab's voxel number `lane` = y*8 + z.
    // Skipped when the (short) list was handled entirely by the voxel-parallel tail -- most tiles at 256^3.
    if (stepped) {
        // widest two steps: v_permlane32_swap / v_permlane16_swap (see raster_render.hip)
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i]), __float_as_uint(acc[32 + i]), false, false);
            acc[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[i]), __float_as_uint(acc[16 + i]), false, false);
            acc[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) {
            const bool up = (lane & d) != 0;
#pragma unroll
            for (int i = 0; i < d; ++i) {
                const float keep = up ? acc[d + i] : acc[i];
                const float send = up ? acc[i] : acc[d + i];
                acc[i] = keep + __shfl_xor(send, d);
            }
        }
    }
    const float value = acc[0] + tail;
    VR_TS(6);
    if (wd.w == 1u) {
        // the tile's only work item (almost every tile at 256^3): write the volume directly, no partial + combine pass
        const int vx = tx * TILE3D + slab, vy = ty * TILE3D + (lane >> 3), vz = tz * TILE3D + (lane & 7);
        if (vx < v.nx && vy < v.ny && vz < v.nz) out[((size_t)vx * v.ny + vy) * v.nz + vz] = value;
    } else {
        partial[(size_t)w * 512 + slab * 64 + lane] = value;   // x*64 + y*8 + z: the layout voxel_combine_kernel expects
    }
}

// The item workgroups of a launch: workgroup `bid` of `item_grid` (a multiple of 1024) takes the half-items hb(bid),
// hb(bid) + item_grid, ...  The host sizes item_grid from an estimate of the work list's length (the list itself exists only on
// the device): with the upper bound it used until round 4, two thirds of the workgroups of a 256^3 query found nothing to do and
// cost 10 us of dispatch; an estimate that falls short just means a second pass for some workgroups.
__device__ __forceinline__ void vfwd_items(
    const uint32_t bid, const uint32_t item_grid,
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ chunk_base, const uint4 *__restrict__ work_tile,
    uint32_t T, const uint32_t *__restrict__ point_list, const float4 *__restrict__ rec, const float4 *__restrict__ ext,
    VoxelGrid v, float *__restrict__ partial, float *__restrict__ out)
{
    // XCD-aware order: workgroup b runs on XCD b % 8.  Runs of 128 consecutive half-items (64 tiles in list order:
    // neighbours that share most of their Gaussians) go to one XCD, so their record gathers hit that XCD's L2; the
    // runs are dealt round-robin over the 8 XCDs, which keeps the dense middle of the volume spread over all of them.
    const uint32_t nhalf = 2u * chunk_base[T];
    const uint32_t seq = bid >> 3;
#pragma nounroll
    for (uint32_t hb = ((seq >> 7) * 8u + (bid & 7u)) * 128u + (seq & 127u); hb < nhalf; hb += item_grid) {
        vfwd_item_body(hb, ranges, chunk_base, work_tile, T, point_list, rec, ext, v, partial, out);
        __syncthreads();   // the next item reuses the LDS buffers
    }
}

__global__ void __launch_bounds__(256, R2_VFWD_WGS) voxel_render_forward_kernel(
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ chunk_base, const uint4 *__restrict__ work_tile,
    uint32_t T, const uint32_t *__restrict__ point_list, const float4 *__restrict__ rec, const float4 *__restrict__ ext,
    VoxelGrid v, float *__restrict__ partial, float *__restrict__ out)
{
    vfwd_items(blockIdx.x, gridDim.x, ranges, chunk_base, work_tile, T, point_list, rec, ext, v, partial, out);
}

// Short tile lists (fewer than VFWD_MIN_STEP entries: 71 % of the non-empty tiles of a 256^3 query, 3 % of the instances).
// They get no work item of the kernel above (launch_build_work(min_len)); here ONE WAVE renders a whole tile with no
// accumulators and no LDS: lane j gathers entry j's record, the wave then takes the entries in list order, pulls an
// entry's 16 scalars out of lane j with v_readlane (-> SGPRs), tests each of the 8 x-slabs (wave-uniform) and evaluates
// the live ones voxel-parallel (lane = voxel y*8+z).  ~45 VGPRs, so 8 waves/SIMD cover the range -> ids -> records
// round trips that dominated when these tiles went through the item kernel (they cost 150 of its 680 us).
__device__ __forceinline__ float lane_bcast(float x, int j)
{
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(x), j));
}
__device__ __forceinline__ void vfwd_short_body(
    const uint32_t bid,
    const uint2 *__restrict__ ranges, uint32_t T, const uint32_t *__restrict__ point_list, const float4 *__restrict__ rec,
    const float4 *__restrict__ ext, VoxelGrid v, float *__restrict__ out)
{
    const uint32_t tile = bid * 4u + (threadIdx.x >> 6);
    if (tile >= T) return;
    const uint2 rg = ranges[tile];
    const int n = (int)(rg.y - rg.x);
    if (n == 0 || n >= VFWD_MIN_STEP) return;   // empty: the combine kernel writes zeros; long: the item kernel
    const int lane = threadIdx.x & 63;
    const int tx = tile % v.gx, ty = (tile / v.gx) % v.gy, tz = tile / (v.gx * v.gy);
    const float y0 = (float)(ty * TILE3D), z0 = (float)(tz * TILE3D);
    float4 ep = make_float4(0.f, 0.f, 0.f, 0.f), eq = ep, er = ep, eh = ep;
    if (lane < n) {
        const uint32_t id = point_list[rg.x + (uint32_t)lane];
        ep = rec[3 * id]; eq = rec[3 * id + 1]; er = rec[3 * id + 2]; eh = ext[id];
    }
    const float vy = y0 + (float)(lane >> 3) + 0.5f, vz = z0 + (float)(lane & 7) + 0.5f;
    // Round 4: every lane first tests ITS entry against the 8 slabs (one pass of 8 tests for the whole list); the walk over
    // the entries then broadcasts only the entries that touch a slab at all and evaluates only the slabs of their masks.
    // (Until round 4 each of the n x 8 tests ran wave-uniformly, 64 lanes computing the same thing: 36 M of the kernel's
    // 229 M VALU instructions at 256^3 for 3 % of the instances.)
    uint32_t mask8 = 0;
    if (lane < n) {
#pragma unroll
        for (int sl = 0; sl < TILE3D; ++sl)
            if (slab_live(ep.x, ep.y, ep.z, eh, er.w, (float)(tx * TILE3D + v.ox + sl) + 0.5f, y0, z0)) mask8 |= 1u << sl;
    }
    float sum[TILE3D];
#pragma unroll
    for (int sl = 0; sl < TILE3D; ++sl) sum[sl] = 0.f;
    unsigned long long todo = __ballot(mask8 != 0);
    while (todo) {
        const int j = __ffsll((long long)todo) - 1;   // list order: the summation order of the reference
        todo &= todo - 1;
        const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)mask8, j);
        const float4 p = make_float4(lane_bcast(ep.x, j), lane_bcast(ep.y, j), lane_bcast(ep.z, j), 0.f);
        const float4 q = make_float4(lane_bcast(eq.x, j), lane_bcast(eq.y, j), lane_bcast(eq.z, j), lane_bcast(eq.w, j));
        const float4 r = make_float4(lane_bcast(er.x, j), lane_bcast(er.y, j), lane_bcast(er.z, j), 0.f);
        const float dy = p.y - vy, dz = p.z - vz;
        const float kyz = dy * (q.w * dy + r.x * dz) + ((r.y * dz) * dz + r.z);
        const float kx = q.y * dy + q.z * dz;
#pragma unroll
        for (int sl = 0; sl < TILE3D; ++sl) {
            if (!((m >> sl) & 1u)) continue;   // wave-uniform (scalar)
            const float xs = (float)(tx * TILE3D + v.ox + sl) + 0.5f;
            const float dx = p.x - xs;
            // same expression tree as the item kernel's voxel-parallel paths
            const float pl = dx * (q.x * dx + kx) + kyz;
            const float al = __builtin_amdgcn_exp2f(pl);
            sum[sl] += ((pl <= r.z) && (al >= ALPHA_MIN_3D)) ? al : 0.f;
        }
    }
    const int oy = ty * TILE3D + (lane >> 3), oz = tz * TILE3D + (lane & 7);
#pragma unroll
    for (int sl = 0; sl < TILE3D; ++sl) {
        const int ox = tx * TILE3D + sl;
        if (ox < v.nx && oy < v.ny && oz < v.nz) out[((size_t)ox * v.ny + oy) * v.nz + oz] = sum[sl];
    }
}

__global__ void __launch_bounds__(256) voxel_render_short_kernel(
    const uint2 *__restrict__ ranges, uint32_t T, const uint32_t *__restrict__ point_list, const float4 *__restrict__ rec,
    const float4 *__restrict__ ext, VoxelGrid v, float *__restrict__ out)
{
    vfwd_short_body(blockIdx.x, ranges, T, point_list, rec, ext, v, out);
}

// Both in ONE launch: the item workgroups first (the long ones), the short-list workgroups behind them start while the
// item tail drains (0.668 -> 0.660 ms at 256^3).  (Measured and left out: short groups interleaved evenly among the item
// groups, so that latency-bound and arithmetic waves mix for the whole kernel: 0.685 ms -- the item workgroups lose the
// neighbourhood of their list order in time.)
__global__ void __launch_bounds__(256, R2_VFWD_WGS) voxel_render_forward_both_kernel(
    const uint32_t item_blocks, const uint2 *__restrict__ ranges, const uint32_t *__restrict__ chunk_base,
    const uint4 *__restrict__ work_tile, uint32_t T, const uint32_t *__restrict__ point_list, const float4 *__restrict__ rec,
    const float4 *__restrict__ ext, VoxelGrid v, float *__restrict__ partial, float *__restrict__ out)
{
    if (blockIdx.x < item_blocks) vfwd_items(blockIdx.x, item_blocks, ranges, chunk_base, work_tile, T, point_list, rec, ext, v, partial, out);
    else vfwd_short_body(blockIdx.x - item_blocks, ranges, T, point_list, rec, ext, v, out);
}

// Debug-mode kernel (voxel-parallel): also tracks n_contrib, which only `debug` callers read back.
__global__ void __launch_bounds__(512) voxel_render_forward_debug_kernel(
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ chunk_base, const uint4 *__restrict__ work_tile,
    uint32_t T, const uint32_t *__restrict__ point_list, const float4 *__restrict__ rec, VoxelGrid v,
    float *__restrict__ partial, uint32_t *__restrict__ partial_last)
{
    const uint32_t w = blockIdx.x;
    if (w >= chunk_base[T]) return;
    const uint4 wd = work_tile[w];
    const uint32_t tile = wd.x, beg = wd.y, end = wd.z;
    const uint2 range = ranges[tile];
    const int tx = tile % v.gx, ty = (tile / v.gx) % v.gy, tz = tile / (v.gx * v.gy);
    const int tid = threadIdx.x;
    // lane = y*8+z, wave = x: z-fastest, see the header comment
    const float fx = (float)(tx * TILE3D + v.ox + (tid >> 6)) + 0.5f, fy = (float)(ty * TILE3D + ((tid >> 3) & 7)) + 0.5f,
                fz = (float)(tz * TILE3D + (tid & 7)) + 0.5f;

    __shared__ float4 s0[512];
    __shared__ float4 s1[512];
    __shared__ float2 s2[512];

    float C = 0.f;
    uint32_t last = 0;
    for (uint32_t base = beg; base < end; base += 512) {
        __syncthreads();
        const uint32_t k = base + tid;
        if (k < end) {
            const uint32_t id = point_list[k];
            s0[tid] = rec[3 * id];
            s1[tid] = rec[3 * id + 1];
            s2[tid] = *reinterpret_cast<const float2 *>(&rec[3 * id + 2]);
        }
        __syncthreads();
        const int n = min(512u, end - base);
#pragma unroll 2
        for (int j = 0; j < n; ++j) {
            const float4 p = s0[j];    // x y z opacity
            const float4 q = s1[j];    // a2 b2 c2 d2
            const float2 r = s2[j];    // e2 f2
            const float dx = p.x - fx, dy = p.y - fy, dz = p.z - fz;
            const float p2 = dx * (q.x * dx + q.y * dy + q.z * dz) + dy * (q.w * dy + r.x * dz) + (r.y * dz) * dz;
            const float alpha = p.w * __builtin_amdgcn_exp2f(p2);
            const bool ok = (p2 <= 0.0f) && (alpha >= ALPHA_MIN_3D);
            C += ok ? alpha : 0.f;
            last = ok ? (base - range.x) + (uint32_t)j + 1u : last;
        }
    }
    partial[(size_t)w * 512 + tid] = C;
    partial_last[(size_t)w * 512 + tid] = last;
}

// adds a tile's partial sums in list order and writes the volume (zeros for empty tiles)
template <bool NCONTRIB>
__global__ void __launch_bounds__(512) voxel_combine_kernel(
    const uint32_t *__restrict__ chunk_base, const float *__restrict__ partial,
    const uint32_t *__restrict__ partial_last, VoxelGrid v, float *__restrict__ out, uint32_t *__restrict__ n_contrib,
    const uint2 *__restrict__ ranges, uint32_t short_min, VoxelPublish pub)
{
    const uint32_t tile = blockIdx.x;
    // side job (small-grid path): the lists move into the binning / image state for the backward; every workgroup takes a share
    for (uint32_t i = blockIdx.x * 512u + threadIdx.x; i < pub.n; i += gridDim.x * 512u) {
        if (i < pub.R) { pub.dst_plist[i] = pub.src_plist[i]; pub.dst_tiles[i] = pub.src_tiles[i]; }
        if (i < pub.T) pub.dst_ranges[i] = pub.src_ranges[i];
        if (i < pub.T + 1u) pub.dst_chunk_base[i] = pub.src_chunk_base[i];
        if (i < pub.NW) pub.dst_work[i] = pub.src_work[i];
    }
    if (!NCONTRIB && short_min) {   // tiles with a short list were rendered by voxel_render_short_kernel
        const uint2 rg = ranges[tile];
        if (rg.y != rg.x && rg.y - rg.x < short_min) return;
    }
    const int tx = tile % v.gx, ty = (tile / v.gx) % v.gy, tz = tile / (v.gx * v.gy);
    const int tid = threadIdx.x;
    const int vx = tx * TILE3D + (tid >> 6), vy = ty * TILE3D + ((tid >> 3) & 7), vz = tz * TILE3D + (tid & 7);
    const uint32_t w0 = chunk_base[tile], w1 = chunk_base[tile + 1];
    if (!NCONTRIB && w1 - w0 == 1u) return;   // written directly by the (production) forward kernel
    float C = 0.f;
    uint32_t last = 0;
    for (uint32_t w = w0; w < w1; ++w) {
        C += partial[(size_t)w * 512 + tid];
        if (NCONTRIB) {
            const uint32_t l = partial_last[(size_t)w * 512 + tid];
            last = l ? l : last;
        }
    }
    if (vx < v.nx && vy < v.ny && vz < v.nz) {
        const size_t vid = ((size_t)vx * v.ny + vy) * v.nz + vz;
        out[vid] = C;
        if (NCONTRIB) n_contrib[vid] = last;
    }
}

// Production form of the combine pass (round 4): ONE WAVE per tile.  Almost every tile of a query has been written by the
// forward kernels already (one work item, or a short list), so the pass is a test per tile; with a 512-thread workgroup per
// tile it was 262144 waves at 256^3 (23 us) to find the 53 empty tiles.  A wave that has work walks the tile's 8 slabs.
__global__ void __launch_bounds__(512) voxel_combine_tiles_kernel(
    const uint32_t *__restrict__ chunk_base, const float *__restrict__ partial, VoxelGrid v, float *__restrict__ out,
    const uint2 *__restrict__ ranges, uint32_t T, uint32_t short_min, VoxelPublish pub)
{
    for (uint32_t i = blockIdx.x * 512u + threadIdx.x; i < pub.n; i += gridDim.x * 512u) {   // side job: see voxel_combine_kernel
        if (i < pub.R) { pub.dst_plist[i] = pub.src_plist[i]; pub.dst_tiles[i] = pub.src_tiles[i]; }
        if (i < pub.T) pub.dst_ranges[i] = pub.src_ranges[i];
        if (i < pub.T + 1u) pub.dst_chunk_base[i] = pub.src_chunk_base[i];
        if (i < pub.NW) pub.dst_work[i] = pub.src_work[i];
    }
    const uint32_t tile = blockIdx.x * 8u + (threadIdx.x >> 6);
    if (tile >= T) return;
    if (short_min) {   // tiles with a short list were rendered by the short-list kernel
        const uint2 rg = ranges[tile];
        if (rg.y != rg.x && rg.y - rg.x < short_min) return;
    }
    const uint32_t w0 = chunk_base[tile], w1 = chunk_base[tile + 1];
    if (w1 - w0 == 1u) return;   // written directly by the forward kernel
    const int lane = threadIdx.x & 63;
    const int tx = tile % v.gx, ty = (tile / v.gx) % v.gy, tz = tile / (v.gx * v.gy);
    const int vy = ty * TILE3D + (lane >> 3), vz = tz * TILE3D + (lane & 7);
    for (int sl = 0; sl < TILE3D; ++sl) {
        float C = 0.f;
        for (uint32_t w = w0; w < w1; ++w) C += partial[(size_t)w * 512 + sl * 64 + lane];   // list order
        const int vx = tx * TILE3D + sl;
        if (vx < v.nx && vy < v.ny && vz < v.nz) out[((size_t)vx * v.ny + vy) * v.nz + vz] = C;
    }
}

__device__ __forceinline__ void voxel_moments(const float4 p, float ry2, float dz, float k0, float k1, float g,
                                              float &r0, float &r1, float &r2)
{
    const float p2 = dz * (k1 + ry2 * dz) + k0;
    const float G = __builtin_amdgcn_exp2f(p2);
    const bool ok = (p2 <= 0.0f) && (p.w * G >= ALPHA_MIN_3D);
    const float w = ok ? G * g : 0.f;
    const float wdz = w * dz;
    r0 += w;
    r1 += wdz;
    r2 += wdz * dz;
}

__device__ __forceinline__ void row_to_moments(float dx, float dy, float r0, float r1, float r2, float *S)
{
    const float wx = dx * r0, wy = dy * r0;
    S[0] += r0;
    S[1] += wx;         // sum w dx
    S[2] += wy;         // sum w dy
    S[3] += r1;         // sum w dz
    S[4] += dx * wx;    // xx
    S[5] += dx * wy;    // xy
    S[6] += dx * r1;    // xz
    S[7] += dy * wy;    // yy
    S[8] += dy * r1;    // yz
    S[9] += r2;         // zz
}

// ------------------------------------------------------------------------------------------------ backward
// Item-parallel like the rasterizer's (raster_render.hip): one LANE owns one ITEM = one x-slab (1 x 8 x 8 voxels) of one
// (tile, Gaussian) instance; only slabs that the Gaussian's alpha >= 1e-6 bounding box touches become items (3-4 of 8 on
// the 32^3 TV patch).  A one-wave workgroup takes 64 consecutive instances of the sorted list, stages dL/dvol of their
// tiles in LDS (three tiles per pass), expands the instances into items through an LDS queue, evaluates 64 items at a
// time, and every instance then adds the moment rows of its own items in queue order -- no atomics, bit-reproducible.
// (The first version let one lane walk all 512 voxels of its tile: twice the arithmetic, and 64 instances per wave meant
// ~700 long-running waves on the TV patch -- 48 us for 45 k instances.)
constexpr int VB_TILES = 3;                 // tiles staged per pass
constexpr int VB_GT = 8 * 8 * 8 + 8;        // floats per staged tile (+ pad: tile slots on different banks)

// moments of w = G * dL/dvol over one x-slab whose dL/dvol rows sit in LDS at gs (row y at gs[y * 8], 8 z values)
__device__ __forceinline__ void slab_moments(const float4 p, const float4 q, const float4 r, const float *__restrict__ gs,
                                             float xc, float y0, float z0, float *M)
{
    const float dx = p.x - xc;
    const float dz0 = p.z - (z0 + 0.5f);
#pragma unroll 2
    for (int iy = 0; iy < TILE3D; ++iy) {
        const float dy = p.y - (y0 + (float)iy + 0.5f);
        const float k0 = dx * (q.x * dx + q.y * dy) + (q.w * dy) * dy;   // a2 dx^2 + b2 dx dy + d2 dy^2
        const float k1 = q.z * dx + r.x * dy;                            // c2 dx + e2 dy
        const float4 g0 = *reinterpret_cast<const float4 *>(gs + iy * 8), g1 = *reinterpret_cast<const float4 *>(gs + iy * 8 + 4);
        const float g[8] = { g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w };
        float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
        for (int iz = 0; iz < TILE3D; ++iz) voxel_moments(p, r.y, dz0 - (float)iz, k0, k1, g[iz], r0, r1, r2);
        row_to_moments(dx, dy, r0, r1, r2, M);
    }
}

__global__ void __launch_bounds__(64) voxel_render_backward_kernel(
    const uint32_t *__restrict__ tiles, const uint32_t *__restrict__ point_list,
    const float4 *__restrict__ rec, const float4 *__restrict__ ext, const uint4 *__restrict__ cube, uint32_t R, VoxelGrid v,
    uint32_t nchunks, uint32_t ipw, const float *__restrict__ dL_dvol, float4 *__restrict__ part, uint32_t work_grid,
    ZeroArrays zero)
{
    if (blockIdx.x >= work_grid) {
        // ---- zero-fill workgroups (patches only): slice (blockIdx - work_grid) of every gradient array, 16-byte stores
        const size_t t = (size_t)(blockIdx.x - work_grid) * 64u + threadIdx.x, nt = (size_t)(gridDim.x - work_grid) * 64u;
#pragma unroll
        for (int a = 0; a < 7; ++a) {
            float *__restrict__ p = zero.p[a];
            if (p == nullptr) continue;
            const size_t n = zero.n[a];
            const size_t head = min(n, (size_t)((16u - ((uintptr_t)p & 15u)) & 15u) / 4u);   // floats before the first 16-byte boundary
            float4 *__restrict__ p4 = reinterpret_cast<float4 *>(p + head);
            const size_t n4 = (n - head) / 4;
            for (size_t i = t; i < n4; i += nt) p4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < head) p[t] = 0.f;
            const size_t tail0 = head + 4 * n4;
            if (t < n - tail0) p[tail0 + t] = 0.f;
        }
        return;
    }
    __shared__ float s_gt[VB_TILES * VB_GT];      // dL/dvol of the pass's tiles: [tile][x][y][z]
    __shared__ float4 s_p[64], s_q[64], s_r[64];  // the wave's 64 instance records
    __shared__ uint16_t s_queue[64 * TILE3D];     // items: (owner lane << 5) | (tile slot << 3) | slab
    __shared__ float s_m[64][11];                 // moment rows of the current round of 64 items (+1 pad)
    const uint32_t chunk = xcd_remap(blockIdx.x, nchunks);
    if (chunk >= nchunks) return;
    const int lane = threadIdx.x;
    // ipw instances per wave (64, or fewer on small problems so that every SIMD gets a few waves: the items of 16
    // instances still fill the 64 lanes of a round)
    const uint32_t k = chunk * ipw + (uint32_t)lane;
    const bool live = (uint32_t)lane < ipw && k < R;
    uint32_t tile = 0xffffffffu, id = 0;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f), q = p, r = p, h = p;
    uint4 cb = make_uint4(0u, 0u, 0u, 0u);
    if (live) {
        tile = tiles[k];
        id = point_list[k];
        p = rec[3 * id];
        q = rec[3 * id + 1];
        r = rec[3 * id + 2];
        h = ext[id];
        cb = cube[id];   // only needed for the final store's address: requested with the other gathers
    }
    float S[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) S[i] = 0.f;
    const uint32_t gxy = (uint32_t)(v.gx * v.gy);

    const uint32_t prev_tile = __shfl_up(tile, 1);
    const unsigned long long heads = __ballot(live && (lane == 0 || tile != prev_tile));
    const int ntiles = __popcll(heads);
    const int my_slot = __popcll(heads & ((2ull << lane) - 1ull)) - 1;   // rank of this lane's tile among the heads
    unsigned long long hh = heads;
    for (int slot0 = 0; slot0 < ntiles; slot0 += VB_TILES) {
        const int npass = min(VB_TILES, ntiles - slot0);
        const bool mine = live && my_slot >= slot0 && my_slot < slot0 + npass;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // the previous pass is done with the LDS buffers
        // ---- stage dL/dvol of this pass's tiles: lane -> (x = lane/8, y = lane%8), its 8 z values as two float4
        for (int slot = 0; slot < npass; ++slot) {
            const int leader = __ffsll((long long)hh) - 1;
            hh &= hh - 1;
            const uint32_t t = __builtin_amdgcn_readfirstlane(__shfl(tile, leader));
            const int x0 = (int)(t % v.gx) * TILE3D, y0 = (int)((t / v.gx) % v.gy) * TILE3D, z0 = (int)(t / gxy) * TILE3D;
            const int vx = x0 + (lane >> 3), vy = y0 + (lane & 7);
            float g[8];
#pragma unroll
            for (int iz = 0; iz < 8; ++iz) g[iz] = 0.f;
            if (vx < v.nx && vy < v.ny) {
                const float *__restrict__ src = dL_dvol + ((size_t)vx * v.ny + vy) * v.nz + z0;
                if (z0 + 7 < v.nz && (v.nz & 3) == 0) {
                    const float4 a = *reinterpret_cast<const float4 *>(src);
                    const float4 b = *reinterpret_cast<const float4 *>(src + 4);
                    g[0] = a.x; g[1] = a.y; g[2] = a.z; g[3] = a.w; g[4] = b.x; g[5] = b.y; g[6] = b.z; g[7] = b.w;
                } else {
#pragma unroll
                    for (int iz = 0; iz < 8; ++iz)
                        if (z0 + iz < v.nz) g[iz] = src[iz];
                }
            }
            float *dst = s_gt + slot * VB_GT + lane * 8;
            *reinterpret_cast<float4 *>(dst) = make_float4(g[0], g[1], g[2], g[3]);
            *reinterpret_cast<float4 *>(dst + 4) = make_float4(g[4], g[5], g[6], g[7]);
        }
        // ---- expand instances into slab items
        const int slot = my_slot - slot0;
        const float tx0 = (float)((int)(tile % v.gx) * TILE3D + v.ox), ty0 = (float)((int)((tile / v.gx) % v.gy) * TILE3D),
                    tz0 = (float)((int)(tile / gxy) * TILE3D);
        uint32_t mask = 0;
        if (mine) {
#pragma unroll
            for (int sl = 0; sl < TILE3D; ++sl)
                if (slab_live(p.x, p.y, p.z, h, r.w, tx0 + (float)sl + 0.5f, ty0, tz0)) mask |= 1u << sl;
        }
        const int cnt = __popc(mask);
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        const int off = incl - cnt;
        const int total = __shfl(incl, 63);
        s_p[lane] = p;
        s_q[lane] = q;
        s_r[lane] = r;
        {
            int o = off;
#pragma unroll
            for (int sl = 0; sl < TILE3D; ++sl)
                if (mask & (1u << sl)) s_queue[o++] = (uint16_t)((lane << 5) | (slot << 3) | sl);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- rounds of 64 items
        for (int base = 0; base < total; base += 64) {
            const int e = base + lane;
            float M[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) M[i] = 0.f;
            const uint32_t item = e < total ? (uint32_t)s_queue[e] : 0u;
            const int owner = (int)(item >> 5), sl_t = (int)((item >> 3) & 3u), sl = (int)(item & 7u);
            const uint32_t ot = __shfl(tile, owner);   // the owner's tile (all lanes take part in the shuffle)
            if (e < total) {
                const float4 op = s_p[owner], oq = s_q[owner], orr = s_r[owner];
                const float ox0 = (float)((int)(ot % v.gx) * TILE3D + v.ox), oy0 = (float)((int)((ot / v.gx) % v.gy) * TILE3D),
                            oz0 = (float)((int)(ot / gxy) * TILE3D);
                slab_moments(op, oq, orr, s_gt + sl_t * VB_GT + sl * 64, ox0 + (float)sl + 0.5f, oy0, oz0, M);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 10; ++i) s_m[lane][i] = M[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // every instance adds the rows of its own items that were processed in this round, in item order
            for (int i = 0; i < cnt; ++i) {
                const int e2 = off + i - base;
                if (e2 >= 0 && e2 < 64) {
#pragma unroll
                    for (int j = 0; j < 10; ++j) S[j] += s_m[e2][j];
                }
            }
        }
    }
    if (live) {
        // scratch row = the instance's EMISSION index, recomputed from the Gaussian's tile cube (the duplicate kernel emits
        // a Gaussian's tiles z-major / y / x-minor from `first`): a Gaussian's rows end up contiguous, the geometry backward
        // streams them, and the tile sort does not have to carry a permutation
        const int ttx = (int)(tile % (uint32_t)v.gx), tty = (int)((tile / (uint32_t)v.gx) % (uint32_t)v.gy),
                  ttz = (int)(tile / ((uint32_t)v.gx * (uint32_t)v.gy));
        const int lox = (int)(cb.y & 0xffffu), loy = (int)(cb.y >> 16), loz = (int)(cb.z & 0xffffu);
        const int cnx = (int)(cb.z >> 16), cny = (int)cb.w;
        const size_t u = (size_t)cb.x + (size_t)(((ttz - loz) * cny + (tty - loy)) * cnx + (ttx - lox));
        part[3 * u] = make_float4(S[0], S[1], S[2], S[3]);
        part[3 * u + 1] = make_float4(S[4], S[5], S[6], S[7]);
        part[3 * u + 2] = make_float4(S[8], S[9], 0.f, 0.f);
    }
}

int launch_voxel_render_forward(const VoxelGeom &g, const VoxelBinning &b, const VoxelImage &im, const VoxelGrid &v,
                                float *out_volume, bool write_ncontrib, hipStream_t s, bool no_short_kernel,
                                const VoxelPublish *publish)
{
    const uint32_t T = (uint32_t)v.gx * v.gy * v.gz;
    const VoxelPublish pub = publish ? *publish : VoxelPublish{};
    if (no_short_kernel) {
        // small grids (voxel_small.hip): the work list covers every non-empty tile (no short-list exemption), so the item kernel
        // and the combine pass do it all; the combine launch also publishes the lists
        if (im.NW > 0)
            voxel_render_forward_kernel<<<dim3((unsigned)(((2 * im.NW + 1023) / 1024) * 1024)), dim3(256), 0, s>>>(
                im.ranges, im.chunk_base, im.work_tile, T, b.point_list, g.rec, g.ext, v, im.partial, out_volume);
        // (a workgroup per tile here: on a small grid most tiles hold several work items, and the publish job wants the lanes)
        voxel_combine_kernel<false><<<dim3(T), dim3(512), 0, s>>>(im.chunk_base, im.partial, im.partial_last, v, out_volume,
                                                                  im.n_contrib, im.ranges, 0u, pub);
        return 0;
    }
    if (write_ncontrib) {
        if (im.NW > 0)
            voxel_render_forward_debug_kernel<<<dim3((unsigned)im.NW), dim3(512), 0, s>>>(
                im.ranges, im.chunk_base, im.work_tile, T, b.point_list, g.rec, v, im.partial, im.partial_last);
        voxel_combine_kernel<true><<<dim3(T), dim3(512), 0, s>>>(im.chunk_base, im.partial, im.partial_last, v, out_volume,
                                                                 im.n_contrib, im.ranges, 0u, pub);
        return 0;
    }
    if (im.NW > 0) {
        // short lists: one wave per tile (the work list holds no item for them, see voxel_short_list_min())
        // grid rounded up to whole 1024-block XCD interleave groups (the in-kernel block -> work item map)
        // item workgroups: an estimate of the work list (two per ~512 instances; im.NW is the upper bound), see vfwd_items
        unsigned item_blocks = (unsigned)(((std::min<size_t>(2 * im.NW, std::max<size_t>(1024, im.R / 256)) + 1023) / 1024) * 1024);
#ifdef R2_EXP_EXACT_GRID   // experiment: what do the workgroups beyond the work list cost?  (host read-back: not a product path)
        {
            uint32_t nitems = 0;
            (void)hipMemcpyAsync(&nitems, im.chunk_base + T, 4, hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            item_blocks = (unsigned)(((2 * (size_t)nitems + 1023) / 1024) * 1024);
        }
#endif
        static const bool split = [] { const char *e = getenv("R2_VOXEL_SPLIT_SHORT"); return e && e[0] == '1'; }();
        if (split) {
            voxel_render_short_kernel<<<dim3((T + 3) / 4), dim3(256), 0, s>>>(im.ranges, T, b.point_list, g.rec, g.ext, v, out_volume);
            voxel_render_forward_kernel<<<dim3(item_blocks), dim3(256), 0, s>>>(
                im.ranges, im.chunk_base, im.work_tile, T, b.point_list, g.rec, g.ext, v, im.partial, out_volume);
        } else {
            voxel_render_forward_both_kernel<<<dim3(item_blocks + (T + 3) / 4), dim3(256), 0, s>>>(
                item_blocks, im.ranges, im.chunk_base, im.work_tile, T, b.point_list, g.rec, g.ext, v, im.partial, out_volume);
        }
    }
    voxel_combine_tiles_kernel<<<dim3((T + 7) / 8), dim3(512), 0, s>>>(im.chunk_base, im.partial, v, out_volume, im.ranges, T,
                                                                        im.NW > 0 ? (uint32_t)VFWD_MIN_STEP : 0u, pub);
    return 0;
}

uint32_t voxel_short_list_min(bool debug) { return debug ? 0u : (uint32_t)VFWD_MIN_STEP; }

int launch_voxel_render_backward(const VoxelGeom &g, const VoxelBinning &b, const VoxelGrid &v, size_t R,
                                 const float *dL_dvol, hipStream_t s, const ZeroArrays *zero)
{
    if (R == 0) return 0;   // (the caller then zero-fills on its own)
    // instances per wave: 64, fewer when that would leave most SIMDs without a wave (the 32^3 TV patch has 45 k instances)
    const uint32_t ipw = R >= ((size_t)1 << 19) ? 64u : (R >= ((size_t)1 << 18) ? 32u : 16u);
    const uint32_t nchunks = (uint32_t)((R + ipw - 1) / ipw);
    const uint32_t grid = ((nchunks + 7u) >> 3) << 3;
    ZeroArrays z{};
    uint32_t zgrid = 0;
    if (zero) {
        z = *zero;
        zgrid = 8192;   // ~30 MB at 300k Gaussians: 3.7 KB per one-wave workgroup
    }
    voxel_render_backward_kernel<<<dim3(grid + zgrid), dim3(64), 0, s>>>(b.tiles, b.point_list, g.rec, g.ext, g.cube, (uint32_t)R, v,
                                                                         nchunks, ipw, dL_dvol, reinterpret_cast<float4 *>(b.part), grid, z);
    return 0;
}

}  // namespace r2
