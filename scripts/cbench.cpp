// cbench.cpp -- the bench.py workload driven straight through the C ABI (no Python, no torch): a host program in the
// style of the reference's own C++ boundary (SUB/rasterize_points.cu) for quick kernel experiments on the GPU box.
//   python scripts/dump_scene.py && hipcc -O2 scripts/cbench.cpp -Iinclude -ldl -o scripts/cbench
//   scripts/cbench [steps] [lib]        (default 200 steps, r2_gaussian_amd/libr2hip.so)
// Prints views/s of forward+backward, the per-stage HIP-event breakdown, and the voxelizer's 256^3 query time.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "r2hip.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Slot { char *p = nullptr; size_t cap = 0; };
static char *grow(size_t bytes, void *user)
{
    Slot *s = static_cast<Slot *>(user);
    if (bytes > s->cap) {
        if (s->p) (void)hipFree(s->p);
        s->cap = bytes + bytes / 4 + (1u << 20);
        if (hipMalloc(reinterpret_cast<void **>(&s->p), s->cap) != hipSuccess) return nullptr;
    }
    return s->p;
}

template <typename T> static T sym(void *h, const char *n)
{
    void *p = dlsym(h, n);
    if (!p) { fprintf(stderr, "missing symbol %s\n", n); exit(1); }
    return reinterpret_cast<T>(p);
}

struct ViewH { float vm[16], pm[16], cam[3], tanx, tany; int mode; };

int main(int argc, char **argv)
{
    const int steps = argc > 1 ? atoi(argv[1]) : 200;
    const std::string libpath = argc > 2 ? argv[2] : "r2_gaussian_amd/libr2hip.so";
    // sections to run (argv[3], comma separated; default all): single,streams,stages,batch,sbatch,voxel -- a profiler pass over
    // "single" alone attributes every kernel row to the single-view step
    const std::string sections = argc > 3 ? std::string(",") + argv[3] + "," : "";
    auto want = [&](const char *name) { return sections.empty() || sections.find(std::string(",") + name + ",") != std::string::npos; };
    void *h = dlopen(libpath.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
    auto fwd = sym<decltype(&r2_raster_forward)>(h, "r2_raster_forward");
    auto bwd = sym<decltype(&r2_raster_backward)>(h, "r2_raster_backward");
    auto vfwd = sym<decltype(&r2_voxel_forward)>(h, "r2_voxel_forward");
    auto vbwd = sym<decltype(&r2_voxel_backward)>(h, "r2_voxel_backward");
    auto prof_enable = sym<decltype(&r2_profile_enable)>(h, "r2_profile_enable");
    auto prof_count = sym<decltype(&r2_profile_stage_count)>(h, "r2_profile_stage_count");
    auto prof_name = sym<decltype(&r2_profile_stage_name)>(h, "r2_profile_stage_name");
    auto prof_read = sym<decltype(&r2_profile_read)>(h, "r2_profile_read");
    auto wait_stats = sym<decltype(&r2_sync_wait_stats)>(h, "r2_sync_wait_stats");
    auto last_error = sym<decltype(&r2_last_error)>(h, "r2_last_error");

    const char *scene_path = getenv("R2_SCENE") ? getenv("R2_SCENE") : "scripts/_scene/scene.bin";   // scripts/dump_scene.py writes them
    FILE *f = fopen(scene_path, "rb");
    if (!f) { fprintf(stderr, "%s missing: run scripts/dump_scene.py\n", scene_path); return 1; }
    int hdr[4];
    if (fread(hdr, 4, 4, f) != 4) return 1;
    const int P = hdr[0], V = hdr[1], H = hdr[2], W = hdr[3];
    std::vector<float> host((size_t)P * 11);
    if (fread(host.data(), 4, host.size(), f) != host.size()) return 1;
    std::vector<ViewH> views(V);
    for (auto &v : views)
        if (fread(&v, sizeof(ViewH), 1, f) != 1) return 1;
    std::vector<float> dLh((size_t)H * W);
    if (fread(dLh.data(), 4, dLh.size(), f) != dLh.size()) return 1;
    fclose(f);

    float *d_in, *d_views, *dL, *out, *grads;
    int *radii;
    CHECK(hipMalloc(reinterpret_cast<void **>(&d_in), host.size() * 4));
    CHECK(hipMemcpy(d_in, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    const float *means = d_in, *dens = d_in + (size_t)3 * P, *scal = d_in + (size_t)4 * P, *rot = d_in + (size_t)7 * P;
    CHECK(hipMalloc(reinterpret_cast<void **>(&d_views), (size_t)V * 64 * 4));   // 64 floats per view: vm, pm, cam
    for (int i = 0; i < V; ++i) {
        CHECK(hipMemcpy(d_views + (size_t)i * 64, views[i].vm, 16 * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_views + (size_t)i * 64 + 16, views[i].pm, 16 * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_views + (size_t)i * 64 + 32, views[i].cam, 3 * 4, hipMemcpyHostToDevice));
    }
    CHECK(hipMalloc(reinterpret_cast<void **>(&dL), dLh.size() * 4));
    CHECK(hipMemcpy(dL, dLh.data(), dLh.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(reinterpret_cast<void **>(&out), (size_t)256 * 256 * 256 * 4));
    CHECK(hipMalloc(reinterpret_cast<void **>(&radii), (size_t)3 * P * 4));
    CHECK(hipMalloc(reinterpret_cast<void **>(&grads), (size_t)32 * P * 4));   // 25 floats/Gaussian (raster), 26 (voxel)
    float *g2d = grads, *gcon = grads + (size_t)3 * P, *gop = grads + (size_t)7 * P, *gmu = grads + (size_t)8 * P,
          *g3d = grads + (size_t)9 * P, *gcov = grads + (size_t)12 * P, *gsc = grads + (size_t)18 * P, *grot = grads + (size_t)21 * P;
    Slot slots[3];
    hipStream_t s;
    CHECK(hipStreamCreate(&s));

    long long Rsum = 0;
    auto step = [&](int k) {
        const ViewH &v = views[k % V];
        const float *dv = d_views + (size_t)(k % V) * 64;
        const int R = fwd(grow, &slots[0], grow, &slots[1], grow, &slots[2], P, W, H, means, dens, scal, 1.f, rot, nullptr, dv,
                          dv + 16, dv + 32, v.tanx, v.tany, 0, v.mode, out, radii, 0, s);
        if (R < 0) { fprintf(stderr, "forward: %d %s\n", R, last_error()); exit(1); }
        Rsum += R;
        const int rc = bwd(P, R, W, H, means, scal, 1.f, rot, nullptr, dv, dv + 16, dv + 32, v.tanx, v.tany, radii, slots[0].p,
                           slots[1].p, slots[2].p, dL, g2d, gcon, gop, gmu, g3d, gcov, gsc, grot, v.mode, 0, s);
        if (rc < 0) { fprintf(stderr, "backward: %d %s\n", rc, last_error()); exit(1); }
    };
    for (int k = 0; k < 20; ++k) step(k);
    CHECK(hipStreamSynchronize(s));
    double best = 1e30;
    for (int rep = 0; want("single") && rep < 3; ++rep) {
        Rsum = 0;
        wait_stats(nullptr, nullptr, 1);
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < steps; ++k) step(20 + k);
        CHECK(hipStreamSynchronize(s));
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        double wus = 0;
        long long wn = 0;
        wait_stats(&wus, &wn, 1);
        printf("rep %d: %.1f views/s  %.1f us/step  (R avg %lld, host wait %.1f us/step)\n", rep, steps / dt, 1e6 * dt / steps,
               Rsum / steps, wus / steps);
        if (dt < best) best = dt;
    }
    if (want("single")) printf("BEST %.1f views/s  %.2f us/step\n", steps / best, 1e6 * best / steps);

    // independent views in flight on several streams, one host thread each (the library's state is per host thread): the
    // launch- and latency-bound binning chain of one view overlaps the render kernels of another.  What a trainer that
    // accumulates the gradients of several views per optimiser step can do without the batched entry point.
    for (int nth : {2, 3}) {
        if (!want("streams")) break;
        struct Ctx { float *out, *grads; int *radii; Slot slots[3]; hipStream_t s; };
        std::vector<Ctx> ctx(nth);
        for (auto &c : ctx) {
            CHECK(hipMalloc(reinterpret_cast<void **>(&c.out), (size_t)H * W * 4));
            CHECK(hipMalloc(reinterpret_cast<void **>(&c.radii), (size_t)P * 4));
            CHECK(hipMalloc(reinterpret_cast<void **>(&c.grads), (size_t)32 * P * 4));
            CHECK(hipStreamCreate(&c.s));
        }
        auto worker = [&](int t, int n, int first) {
            Ctx &c = ctx[t];
            float *a = c.grads;
            for (int k = 0; k < n; ++k) {
                const int vi = (first + k * nth + t) % V;
                const ViewH &v = views[vi];
                const float *dv = d_views + (size_t)vi * 64;
                const int R = fwd(grow, &c.slots[0], grow, &c.slots[1], grow, &c.slots[2], P, W, H, means, dens, scal, 1.f, rot, nullptr,
                                  dv, dv + 16, dv + 32, v.tanx, v.tany, 0, v.mode, c.out, c.radii, 0, c.s);
                if (R < 0) { fprintf(stderr, "forward (thread %d): %d %s\n", t, R, last_error()); exit(1); }
                const int rc = bwd(P, R, W, H, means, scal, 1.f, rot, nullptr, dv, dv + 16, dv + 32, v.tanx, v.tany, c.radii,
                                   c.slots[0].p, c.slots[1].p, c.slots[2].p, dL, a, a + (size_t)3 * P, a + (size_t)7 * P,
                                   a + (size_t)8 * P, a + (size_t)9 * P, a + (size_t)12 * P, a + (size_t)18 * P, a + (size_t)21 * P,
                                   v.mode, 0, c.s);
                if (rc < 0) { fprintf(stderr, "backward (thread %d): %d %s\n", t, rc, last_error()); exit(1); }
            }
            CHECK(hipStreamSynchronize(c.s));
        };
        // persistent threads (the library keeps its depth-range history and pinned mailbox per host thread); a spin barrier
        // separates warm-up and the timed repetitions
        std::atomic<int> arrived{0};
        auto barrier = [&](int round) {
            arrived.fetch_add(1);
            while (arrived.load() < round * (nth + 1)) std::this_thread::yield();
        };
        std::vector<std::thread> th;
        for (int t = 0; t < nth; ++t)
            th.emplace_back([&, t]() {
                worker(t, 20, 0);
                for (int rep = 0; rep < 3; ++rep) {
                    barrier(2 * rep + 1);
                    worker(t, steps, 20);
                    barrier(2 * rep + 2);
                }
            });
        double bestn = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
            barrier(2 * rep + 1);
            const auto t0 = std::chrono::steady_clock::now();
            barrier(2 * rep + 2);
            bestn = std::min(bestn, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        }
        for (auto &x : th) x.join();
        printf("STREAMS %d: %.1f views/s  %.2f us/view  (%d host threads x %d views, one stream each)\n", nth, nth * steps / bestn,
               1e6 * bestn / (nth * steps), nth, steps);
        for (auto &c : ctx) {
            (void)hipFree(c.out); (void)hipFree(c.radii); (void)hipFree(c.grads); (void)hipStreamDestroy(c.s);
            for (auto &sl : c.slots) if (sl.p) (void)hipFree(sl.p);
        }
    }

    const int ns = prof_count();
    std::vector<double> ms(ns);
    std::vector<long long> cnt(ns);
    if (want("stages")) {
    prof_enable(~0ull);
    for (int k = 0; k < 50; ++k) step(20 + k);
    CHECK(hipStreamSynchronize(s));
    prof_read(ms.data(), cnt.data(), 1);
    double sum = 0;
    for (int i = 0; i < ns; ++i)
        if (cnt[i] && !strncmp(prof_name(i), "raster.", 7)) {
            printf("  %-20s %8.2f us\n", prof_name(i), 1e3 * ms[i] / cnt[i]);
            sum += 1e3 * ms[i] / cnt[i];
        }
    printf("  %-20s %8.2f us\n", "raster stage sum", sum);
    prof_enable(0);
    }

    // experiment builds (-DR2_EXP_TS) stamp s_memrealtime at phase boundaries inside selected kernels: one [phase][block]
    // table per translation unit; printed as one timeline of the last step (us since the earliest stamp)
    {
        const char *units[] = {"order", "geom", "sort", "tilefirst", "render"};
        std::vector<std::vector<unsigned long long>> tabs;
        std::vector<std::string> names;
        bool any = false;
        for (const char *u : units)
            if (dlsym(h, (std::string("r2_debug_ts_") + u).c_str())) any = true;
        if (any) {
            step(7);
            CHECK(hipStreamSynchronize(s));
            unsigned long long t0 = ~0ull;
            for (const char *u : units) {
                void *pt = dlsym(h, (std::string("r2_debug_ts_") + u).c_str());
                if (!pt) continue;
                std::vector<unsigned long long> ts(16 * 2048);
                reinterpret_cast<int (*)(unsigned long long *)>(pt)(ts.data());
                for (unsigned long long v : ts)
                    if (v && v < t0) t0 = v;
                tabs.push_back(ts);
                names.push_back(u);
            }
            // only the stamps of the step just run: the tables also hold what kernels of earlier calls (other chains, larger
            // grids) left in slots this step did not overwrite.  A step is < 1 ms; the newest stamp ends it.
            unsigned long long tmax = 0;
            for (const auto &tb : tabs)
                for (unsigned long long v : tb) tmax = std::max(tmax, v);
            const unsigned long long tlo = tmax > 100000ull ? tmax - 100000ull : 0ull;   // 1 ms at 100 MHz
            t0 = ~0ull;
            for (const auto &tb : tabs)
                for (unsigned long long v : tb)
                    if (v >= tlo && v < t0) t0 = v;
            for (size_t k = 0; k < tabs.size(); ++k)
                for (int ph = 0; ph < 16; ++ph) {
                    std::vector<double> v;
                    for (int b = 0; b < 2048; ++b)
                        if (tabs[k][ph * 2048 + b] >= tlo && tabs[k][ph * 2048 + b]) v.push_back((double)(tabs[k][ph * 2048 + b] - t0) * 0.01);   // 100 MHz -> us
                    if (v.empty()) continue;
                    std::sort(v.begin(), v.end());
                    printf("  TS %-6s %2d: n %4zu  min %7.2f  med %7.2f  max %7.2f us\n", names[k].c_str(), ph, v.size(), v.front(),
                           v[v.size() / 2], v.back());
                }
            // per-workgroup phase durations: for every unit, the difference between consecutive stamped phases of the SAME slot
            for (size_t k = 0; k < tabs.size(); ++k) {
                int prev = -1;
                for (int ph = 0; ph < 16; ++ph) {
                    bool has = false;
                    for (int b = 0; b < 2047 && !has; ++b) has = tabs[k][ph * 2048 + b] >= tlo;
                    if (!has) continue;
                    if (prev >= 0) {
                        std::vector<double> d;
                        double sum = 0;
                        for (int b = 0; b < 2047; ++b) {
                            const unsigned long long x = tabs[k][prev * 2048 + b], y = tabs[k][ph * 2048 + b];
                            if (x >= tlo && y >= x) { d.push_back((double)(y - x) * 0.01); sum += d.back(); }
                        }
                        if (!d.empty()) {
                            std::sort(d.begin(), d.end());
                            printf("  TSD %-6s %2d->%2d: n %4zu  mean %6.2f  med %6.2f  p90 %6.2f  max %6.2f us\n", names[k].c_str(), prev, ph,
                                   d.size(), sum / d.size(), d[d.size() / 2], d[d.size() * 9 / 10], d.back());
                        }
                    }
                    prev = ph;
                }
            }
        }
    }

    // batched views (r2_raster_forward_batch / _backward_batch; absent from older builds of the library): BV views per call
    if (void *pf = want("batch") ? dlsym(h, "r2_raster_forward_batch") : nullptr) {
        auto bfwd = reinterpret_cast<decltype(&r2_raster_forward_batch)>(pf);
        auto bbwd = sym<decltype(&r2_raster_backward_batch)>(h, "r2_raster_backward_batch");
        for (int BV : {2, 4, 8}) {
            if (BV > V) break;
            float *bvm, *bpm, *bout, *bdL, *bg;
            int *bradii;
            CHECK(hipMalloc(reinterpret_cast<void **>(&bvm), (size_t)V * 16 * 4));
            CHECK(hipMalloc(reinterpret_cast<void **>(&bpm), (size_t)V * 16 * 4));
            for (int i = 0; i < V; ++i) {
                CHECK(hipMemcpy(bvm + (size_t)i * 16, views[i].vm, 64, hipMemcpyHostToDevice));
                CHECK(hipMemcpy(bpm + (size_t)i * 16, views[i].pm, 64, hipMemcpyHostToDevice));
            }
            CHECK(hipMalloc(reinterpret_cast<void **>(&bout), (size_t)BV * H * W * 4));
            CHECK(hipMalloc(reinterpret_cast<void **>(&bdL), (size_t)BV * H * W * 4));
            for (int i = 0; i < BV; ++i) CHECK(hipMemcpy(bdL + (size_t)i * H * W, dLh.data(), dLh.size() * 4, hipMemcpyHostToDevice));
            CHECK(hipMalloc(reinterpret_cast<void **>(&bradii), (size_t)BV * P * 4));
            CHECK(hipMalloc(reinterpret_cast<void **>(&bg), ((size_t)8 * BV + 17) * P * 4));
            float *q2d = bg, *qcon = bg + (size_t)3 * BV * P, *qmu = bg + (size_t)7 * BV * P, *qop = bg + (size_t)8 * BV * P,
                  *q3d = qop + P, *qcov = q3d + (size_t)3 * P, *qsc = qcov + (size_t)6 * P, *qrot = qsc + (size_t)3 * P;
            const int nb = V / BV;   // batches of consecutive views
            long long Rb = 0;
            auto bstep = [&](int k) {
                const int v0 = (k % nb) * BV;
                const ViewH &v = views[v0];
                const int R = bfwd(grow, &slots[0], grow, &slots[1], grow, &slots[2], P, BV, W, H, means, dens, scal, 1.f, rot, nullptr,
                                   bvm + (size_t)v0 * 16, bpm + (size_t)v0 * 16, v.tanx, v.tany, v.mode, bout, bradii, 0, s);
                if (R < 0) { fprintf(stderr, "batch forward: %d %s\n", R, last_error()); exit(1); }
                Rb += R;
                const int rc = bbwd(P, BV, R, W, H, means, scal, 1.f, rot, nullptr, bvm + (size_t)v0 * 16, bpm + (size_t)v0 * 16, v.tanx,
                                    v.tany, bradii, slots[0].p, slots[1].p, slots[2].p, bdL, q2d, qcon, qop, qmu, q3d, qcov, qsc, qrot,
                                    v.mode, 0, s);
                if (rc < 0) { fprintf(stderr, "batch backward: %d %s\n", rc, last_error()); exit(1); }
            };
            for (int k = 0; k < 2 * nb + 4; ++k) bstep(k);
            CHECK(hipStreamSynchronize(s));
            double bbest = 1e30;
            const int bsteps = std::max(20, steps / BV);
            for (int rep = 0; rep < 3; ++rep) {
                Rb = 0;
                const auto tb = std::chrono::steady_clock::now();
                for (int k = 0; k < bsteps; ++k) bstep(k);
                CHECK(hipStreamSynchronize(s));
                bbest = std::min(bbest, std::chrono::duration<double>(std::chrono::steady_clock::now() - tb).count());
            }
            printf("BATCH V=%d: %.1f views/s  %.2f us/view  (%.1f us per call, R avg per view %lld)\n", BV, bsteps * BV / bbest,
                   1e6 * bbest / (bsteps * BV), 1e6 * bbest / bsteps, Rb / ((long long)bsteps * BV));
            prof_enable(~0ull);
            for (int k = 0; k < 20; ++k) bstep(k);
            CHECK(hipStreamSynchronize(s));
            prof_read(ms.data(), cnt.data(), 1);
            for (int i = 0; i < ns; ++i)
                if (cnt[i] && !strncmp(prof_name(i), "raster.", 7))
                    printf("  V=%d %-18s %8.2f us/view\n", BV, prof_name(i), 1e3 * ms[i] / cnt[i] / BV);
            prof_enable(0);
            (void)hipFree(bvm); (void)hipFree(bpm); (void)hipFree(bout); (void)hipFree(bdL); (void)hipFree(bradii); (void)hipFree(bg);
        }
    }

    // both: two host threads / streams, each pushing BATCHES of BV views through the batched entry points
    if (void *pf2 = want("sbatch") ? dlsym(h, "r2_raster_forward_batch") : nullptr) {
        auto bfwd = reinterpret_cast<decltype(&r2_raster_forward_batch)>(pf2);
        auto bbwd = sym<decltype(&r2_raster_backward_batch)>(h, "r2_raster_backward_batch");
        float *bvm, *bpm;
        CHECK(hipMalloc(reinterpret_cast<void **>(&bvm), (size_t)V * 16 * 4));
        CHECK(hipMalloc(reinterpret_cast<void **>(&bpm), (size_t)V * 16 * 4));
        for (int i = 0; i < V; ++i) {
            CHECK(hipMemcpy(bvm + (size_t)i * 16, views[i].vm, 64, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(bpm + (size_t)i * 16, views[i].pm, 64, hipMemcpyHostToDevice));
        }
        for (int BV : {2, 4}) {
            const int nth = 2, nb = V / BV, bsteps = std::max(20, steps / BV);
            struct BCtx { float *out, *dL, *g; int *radii; Slot slots[3]; hipStream_t s; };
            std::vector<BCtx> ctx(nth);
            for (auto &c : ctx) {
                CHECK(hipMalloc(reinterpret_cast<void **>(&c.out), (size_t)BV * H * W * 4));
                CHECK(hipMalloc(reinterpret_cast<void **>(&c.dL), (size_t)BV * H * W * 4));
                for (int i = 0; i < BV; ++i) CHECK(hipMemcpy(c.dL + (size_t)i * H * W, dLh.data(), dLh.size() * 4, hipMemcpyHostToDevice));
                CHECK(hipMalloc(reinterpret_cast<void **>(&c.radii), (size_t)BV * P * 4));
                CHECK(hipMalloc(reinterpret_cast<void **>(&c.g), ((size_t)8 * BV + 17) * P * 4));
                CHECK(hipStreamCreate(&c.s));
            }
            auto worker = [&](int t, int n) {
                BCtx &c = ctx[t];
                float *q2d = c.g, *qcon = c.g + (size_t)3 * BV * P, *qmu = c.g + (size_t)7 * BV * P, *qop = c.g + (size_t)8 * BV * P,
                      *q3d = qop + P, *qcov = q3d + (size_t)3 * P, *qsc = qcov + (size_t)6 * P, *qrot = qsc + (size_t)3 * P;
                for (int k = 0; k < n; ++k) {
                    const int v0 = ((k * nth + t) % nb) * BV;
                    const ViewH &v = views[v0];
                    const int R = bfwd(grow, &c.slots[0], grow, &c.slots[1], grow, &c.slots[2], P, BV, W, H, means, dens, scal, 1.f, rot,
                                       nullptr, bvm + (size_t)v0 * 16, bpm + (size_t)v0 * 16, v.tanx, v.tany, v.mode, c.out, c.radii, 0, c.s);
                    if (R < 0) { fprintf(stderr, "batch forward (thread %d): %d %s\n", t, R, last_error()); exit(1); }
                    const int rc = bbwd(P, BV, R, W, H, means, scal, 1.f, rot, nullptr, bvm + (size_t)v0 * 16, bpm + (size_t)v0 * 16,
                                        v.tanx, v.tany, c.radii, c.slots[0].p, c.slots[1].p, c.slots[2].p, c.dL, q2d, qcon, qop, qmu, q3d,
                                        qcov, qsc, qrot, v.mode, 0, c.s);
                    if (rc < 0) { fprintf(stderr, "batch backward (thread %d): %d %s\n", t, rc, last_error()); exit(1); }
                }
                CHECK(hipStreamSynchronize(c.s));
            };
            std::atomic<int> arrived{0};
            auto barrier = [&](int round) {
                arrived.fetch_add(1);
                while (arrived.load() < round * (nth + 1)) std::this_thread::yield();
            };
            std::vector<std::thread> th;
            for (int t = 0; t < nth; ++t)
                th.emplace_back([&, t]() {
                    worker(t, 2 * nb + 4);
                    for (int rep = 0; rep < 3; ++rep) {
                        barrier(2 * rep + 1);
                        worker(t, bsteps);
                        barrier(2 * rep + 2);
                    }
                });
            double bestn = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                barrier(2 * rep + 1);
                const auto t0 = std::chrono::steady_clock::now();
                barrier(2 * rep + 2);
                bestn = std::min(bestn, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            }
            for (auto &x : th) x.join();
            printf("STREAMS 2 x BATCH V=%d: %.1f views/s  %.2f us/view\n", BV, (double)nth * bsteps * BV / bestn,
                   1e6 * bestn / ((double)nth * bsteps * BV));
            for (auto &c : ctx) {
                (void)hipFree(c.out); (void)hipFree(c.dL); (void)hipFree(c.radii); (void)hipFree(c.g); (void)hipStreamDestroy(c.s);
                for (auto &sl : c.slots) if (sl.p) (void)hipFree(sl.p);
            }
        }
        (void)hipFree(bvm); (void)hipFree(bpm);
    }

    if (!want("voxel")) return 0;
    // voxelizer: the full 256^3 query
    auto vox = [&]() {
        const int R3 = vfwd(grow, &slots[0], grow, &slots[1], grow, &slots[2], P, 256, 256, 256, 2.f, 2.f, 2.f, 0.f, 0.f, 0.f, means,
                            dens, scal, 1.f, rot, nullptr, 0, out, radii, radii + P, radii + 2 * (size_t)P, 0, s);
        if (R3 < 0) { fprintf(stderr, "voxel forward: %d %s\n", R3, last_error()); exit(1); }
        return R3;
    };
    int R3 = 0;
    for (int k = 0; k < 3; ++k) R3 = vox();
    CHECK(hipStreamSynchronize(s));
    const auto t1 = std::chrono::steady_clock::now();
    for (int k = 0; k < 10; ++k) vox();
    CHECK(hipStreamSynchronize(s));
    const double tv = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() / 10;
    printf("voxel 256^3: %.3f ms  %.2f GVoxel/s  (R3 %d)\n", tv * 1e3, 256.0 * 256 * 256 / tv / 1e9, R3);
    // experiment builds: the stamp table of the stick-first chain's kernels (voxel_sticks.hip) for ONE more call, raw, for
    // scripts/sticks_timeline.py ([16 phases][2048 workgroups] of 100 MHz ticks)
    if (void *pt = dlsym(h, "r2_debug_ts_sticks")) {
        CHECK(hipStreamSynchronize(s));
        vox();
        CHECK(hipStreamSynchronize(s));
        std::vector<unsigned long long> ts(16 * 2048);
        reinterpret_cast<int (*)(unsigned long long *)>(pt)(ts.data());
        const char *path = getenv("R2_TS_DUMP") ? getenv("R2_TS_DUMP") : "gpurun_out/ts_sticks.bin";
        if (FILE *f = fopen(path, "wb")) { fwrite(ts.data(), 8, ts.size(), f); fclose(f); printf("stamps of the stick chain -> %s\n", path); }
    }
    // experiment builds: stamps inside the item workgroups of the voxel render kernel (voxel_render.hip, VR_TS): phase durations of
    // the first 2047 work items (their first half-item), by item length
    if (void *pt = dlsym(h, "r2_debug_ts_vrender")) {
        CHECK(hipStreamSynchronize(s));
        vox();
        CHECK(hipStreamSynchronize(s));
        std::vector<unsigned long long> ts(16 * 2048);
        reinterpret_cast<int (*)(unsigned long long *)>(pt)(ts.data());
        auto at = [&](int ph, int b) { return ts[(size_t)ph * 2048 + b]; };
        const int edges[] = {0, 256, 512, 768, 1025};
        const int pairs[][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 4}, {4, 8}, {8, 9}, {9, 10}, {4, 5}, {10, 5}, {5, 6}, {0, 6}};
        for (int e = 0; e + 1 < 5; ++e) {
            int n = 0;
            for (int b = 0; b < 2047; ++b)
                if (at(0, b) && (int)at(15, b) >= edges[e] && (int)at(15, b) < edges[e + 1]) ++n;
            printf("  vrender items of %d..%d entries: %d\n", edges[e], edges[e + 1] - 1, n);
            for (const auto &pr : pairs) {
                std::vector<double> d;
                double sum = 0;
                for (int b = 0; b < 2047; ++b) {
                    if (!at(0, b) || (int)at(15, b) < edges[e] || (int)at(15, b) >= edges[e + 1]) continue;
                    const unsigned long long x = at(pr[0], b), y = at(pr[1], b);
                    if (x >= at(0, b) && y >= x && y - x < 100000ull) { d.push_back((double)(y - x) * 0.01); sum += d.back(); }
                }
                if (d.empty()) continue;
                std::sort(d.begin(), d.end());
                printf("    VR %2d->%2d: n %4zu  mean %6.2f  med %6.2f  p90 %6.2f us\n", pr[0], pr[1], d.size(), sum / d.size(), d[d.size() / 2],
                       d[d.size() * 9 / 10]);
            }
        }
    }
    // the training loop's TV regulariser: forward + backward on a 32^3 sub-volume (train.py, tv_vol_size = 32)
    {
        float *dLv;
        CHECK(hipMalloc(reinterpret_cast<void **>(&dLv), (size_t)32 * 32 * 32 * 4));
        CHECK(hipMemset(dLv, 0x3c, (size_t)32 * 32 * 32 * 4));   // some small positive floats
        float *gn = grads, *gc3 = grads + (size_t)3 * P, *go = grads + (size_t)9 * P, *gm = grads + (size_t)10 * P,
              *gcv = grads + (size_t)13 * P, *gs = grads + (size_t)19 * P, *gr = grads + (size_t)22 * P;
        auto tv = [&](int k) {
            const float cx = -0.3f + 0.1f * (float)(k % 7), cy = 0.1f * (float)(k % 5) - 0.2f, cz = 0.05f * (float)(k % 9) - 0.2f;
            const int R3s = vfwd(grow, &slots[0], grow, &slots[1], grow, &slots[2], P, 32, 32, 32, 0.25f, 0.25f, 0.25f, cx, cy, cz, means,
                                 dens, scal, 1.f, rot, nullptr, 0, out, radii, radii + P, radii + 2 * (size_t)P, 0, s);
            if (R3s < 0) { fprintf(stderr, "voxel forward (tv): %d %s\n", R3s, last_error()); exit(1); }
            const int rc = vbwd(P, R3s, 32, 32, 32, 0.25f, 0.25f, 0.25f, cx, cy, cz, means, scal, 1.f, rot, nullptr, radii, radii + P,
                                radii + 2 * (size_t)P, slots[0].p, slots[1].p, slots[2].p, dLv, gn, gc3, go, gm, gcv, gs, gr, 0, s);
            if (rc < 0) { fprintf(stderr, "voxel backward (tv): %d %s\n", rc, last_error()); exit(1); }
            return R3s;
        };
        int R3s = 0;
        for (int k = 0; k < 10; ++k) R3s = tv(k);
        CHECK(hipStreamSynchronize(s));
        const auto t2 = std::chrono::steady_clock::now();
        for (int k = 0; k < 100; ++k) tv(k);
        CHECK(hipStreamSynchronize(s));
        const double tt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t2).count() / 100;
        printf("voxel 32^3 fwd+bwd (TV patch): %.1f us  (R3 %d)\n", tt * 1e6, R3s);
        prof_enable(~0ull);
        for (int k = 0; k < 20; ++k) tv(k);
        CHECK(hipStreamSynchronize(s));
        prof_read(ms.data(), cnt.data(), 1);
        for (int i = 0; i < ns; ++i)
            if (cnt[i] && !strncmp(prof_name(i), "voxel.", 6)) printf("  tv %-17s %8.2f us\n", prof_name(i), 1e3 * ms[i] / cnt[i]);
        prof_enable(0);
    }
    // (one call at a time: a stage's begin event is stamped when the command processor reaches it, which with calls queued back to
    // back is while the PREVIOUS call's render kernel still runs -- round 2's "voxel.preprocess 348 us" was that tail, not the
    // 24 us kernel rocprofv3 shows)
    prof_enable(~0ull);
    for (int k = 0; k < 5; ++k) { vox(); CHECK(hipStreamSynchronize(s)); }
    CHECK(hipStreamSynchronize(s));
    prof_read(ms.data(), cnt.data(), 1);
    for (int i = 0; i < ns; ++i)
        if (cnt[i] && !strncmp(prof_name(i), "voxel.", 6)) printf("  %-20s %8.2f us\n", prof_name(i), 1e3 * ms[i] / cnt[i]);
    return 0;
}
