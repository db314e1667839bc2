// torch_shim.cpp -- the torch boundary of libr2hip.so as a compiled module (r2_gaussian_amd/_r2shim.so).
//
// Plays the part of the reference's SUB/rasterize_points.cu / SUB/voxelize_points.cu (the glue between torch tensors
// and the native layer): output / gradient / state tensors are allocated here, the three allocation callbacks of the
// C ABI (include/r2hip.h) resize torch byte tensors (the reference's resizeFunctional, SUB/utility.h:7-13), and raw
// device pointers + the caller's HIP stream go down to the library.  Host C++ only -- no kernels, no HIP runtime calls;
// the same job is also implemented in Python over ctypes (r2_gaussian_amd/_C.py), which costs ~70 us more interpreter
// time per training view; _C.py uses this module when it has been built.
#include <torch/extension.h>
#include <c10/core/DeviceGuard.h>
#include <stdexcept>
#include <string>
#include <tuple>
#include "../../include/r2hip.h"

namespace {

using torch::Tensor;

[[noreturn]] void fail(const char *what, int rc)
{
    const char *msg = r2_last_error();
    throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + (msg ? msg : ""));
}

void require_gpu(const Tensor &t, const char *name)
{
    if (!t.is_cuda())
        throw std::runtime_error(std::string(name) + " must be a GPU tensor: the MI355X kernels have no CPU fallback");
}

// contiguous float32 tensor on the kernels' device; undefined for the reference's empty placeholders
Tensor dev_f32(const Tensor &t, const c10::Device &dev)
{
    if (!t.defined() || t.numel() == 0) return Tensor();
    Tensor r = t;
    if (r.scalar_type() != torch::kFloat) r = r.to(torch::kFloat);
    if (r.device() != dev) r = r.to(dev);
    r = r.contiguous();
    // the kernels read rotations / dL_dpix rows 16 bytes at a time (include/r2hip.h, "Alignment"): a view carved out of a
    // flat parameter buffer at an odd offset is contiguous but not 16-byte aligned -> take an (allocator-aligned) copy
    if (reinterpret_cast<uintptr_t>(r.data_ptr()) & 15u) r = r.clone();
    return r;
}
const float *fptr(const Tensor &t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
char *bptr(const Tensor &t) { return (t.defined() && t.numel() > 0) ? reinterpret_cast<char *>(t.data_ptr()) : nullptr; }

// The state sizes follow num_rendered, which changes from view to view: large requests are rounded up to a coarse grid
// so that the caching allocator sees a handful of recurring sizes instead of a new one per view (a miss is a hipMalloc,
// i.e. a device synchronisation in the middle of the forward pass).
size_t round_up(size_t n)
{
    if (n > (size_t(1) << 20)) {
        int bits = 0;
        for (size_t v = n; v; v >>= 1) ++bits;
        const size_t g = size_t(1) << std::max(20, bits - 4);   // 1/16 .. 1/8 of the size
        n = (n + g - 1) / g * g;
    }
    return n;
}
char *resize_state(size_t bytes, void *user)
{
    try {
        Tensor *t = static_cast<Tensor *>(user);
        t->resize_({(int64_t)round_up(bytes)});
        return reinterpret_cast<char *>(t->data_ptr());
    } catch (...) {
        return nullptr;   // out of memory etc.: the library turns NULL into R2_ERR_ALLOC
    }
}

// (num_rendered, out_color[1,H,W], radii[P] i32, geomBuffer, binningBuffer, imgBuffer)   <- SUB/rasterize_points.cu:28-97
std::tuple<int64_t, Tensor, Tensor, Tensor, Tensor, Tensor> rasterize_gaussians(
    const Tensor &means3D, const Tensor &opacity, const Tensor &scales, const Tensor &rotations, double scale_modifier,
    const Tensor &cov3D_precomp, const Tensor &viewmatrix, const Tensor &projmatrix, double tan_fovx, double tan_fovy,
    int64_t image_height, int64_t image_width, const Tensor &campos, bool prefiltered, int64_t mode, bool debug,
    int64_t stream)
{
    if (means3D.dim() != 2 || means3D.size(1) != 3) throw std::runtime_error("means3D must have dimensions (num_points, 3)");
    require_gpu(means3D, "means3D");
    const c10::Device dev = means3D.device();
    const int64_t P = means3D.size(0), H = image_height, W = image_width;
    const auto bytes = torch::TensorOptions().dtype(torch::kUInt8).device(dev);
    Tensor geom = torch::empty({0}, bytes), binning = torch::empty({0}, bytes), img = torch::empty({0}, bytes);
    if (P == 0)   // SUB/rasterize_points.cu:58-70: zero image, no state
        return {0, torch::zeros({1, H, W}, means3D.options().dtype(torch::kFloat)),
                torch::zeros({0}, means3D.options().dtype(torch::kInt)), geom, binning, img};
    // both outputs are written in full by the kernels (every pixel, every Gaussian): no zero-fill needed
    Tensor out_color = torch::empty({1, H, W}, means3D.options().dtype(torch::kFloat));
    Tensor radii = torch::empty({P}, means3D.options().dtype(torch::kInt));
    const Tensor m3 = dev_f32(means3D, dev), op = dev_f32(opacity, dev), sc = dev_f32(scales, dev), ro = dev_f32(rotations, dev),
                 cp = dev_f32(cov3D_precomp, dev), vm = dev_f32(viewmatrix, dev), pm = dev_f32(projmatrix, dev),
                 cam = dev_f32(campos, dev);
    int rc;
    {
        c10::DeviceGuard guard(dev);
        py::gil_scoped_release nogil;
        rc = r2_raster_forward(resize_state, &geom, resize_state, &binning, resize_state, &img, (int)P, (int)W, (int)H, fptr(m3),
                               fptr(op), fptr(sc), (float)scale_modifier, fptr(ro), fptr(cp), fptr(vm), fptr(pm), fptr(cam),
                               (float)tan_fovx, (float)tan_fovy, prefiltered ? 1 : 0, (int)mode, out_color.data_ptr<float>(),
                               radii.data_ptr<int>(), debug ? 1 : 0, reinterpret_cast<void *>(stream));
    }
    if (rc < 0) fail("r2_raster_forward", rc);
    return {rc, out_color, radii, geom, binning, img};
}

// (dL_dmeans2D[P,3], dL_dopacity[P,1], dL_dmu[P,1], dL_dmeans3D[P,3], dL_dcov3D[P,6], dL_dscales[P,3], dL_drotations[P,4])
// <- SUB/rasterize_points.cu:99-164
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> rasterize_gaussians_backward(
    const Tensor &means3D, const Tensor &radii, const Tensor &scales, const Tensor &rotations, double scale_modifier,
    const Tensor &cov3D_precomp, const Tensor &viewmatrix, const Tensor &projmatrix, double tan_fovx, double tan_fovy,
    const Tensor &dL_dout_color, const Tensor &campos, const Tensor &geomBuffer, int64_t R, const Tensor &binningBuffer,
    const Tensor &imageBuffer, int64_t mode, bool debug, int64_t stream)
{
    require_gpu(means3D, "means3D");
    const c10::Device dev = means3D.device();
    const int64_t P = means3D.size(0);
    const int64_t H = dL_dout_color.size(1), W = dL_dout_color.size(2);
    // one allocation for all eight gradient arrays (25 floats per Gaussian), 16-byte rows first; the kernels write every
    // row (zeros for culled Gaussians), so no fill
    Tensor flat = torch::empty({25 * P}, means3D.options().dtype(torch::kFloat));
    int64_t o = 0;
    auto carve = [&](int64_t k, at::IntArrayRef shape) {
        Tensor v = flat.narrow(0, o, k * P).view(shape);
        o += k * P;
        return v;
    };
    // conic and rot first (16-byte rows), then the four parameter gradients a trainer exchanges between GPUs ADJACENT to
    // each other: rot | means3D | scales | opacity = one contiguous [11 P] block (dist.grad_block: all-reduce without a copy)
    Tensor dL_dconic = carve(4, {P, 2, 2}), dL_drot = carve(4, {P, 4}), dL_dmeans3D = carve(3, {P, 3}), dL_dscales = carve(3, {P, 3}),
           dL_dopacity = carve(1, {P, 1}), dL_dmeans2D = carve(3, {P, 3}), dL_dmu = carve(1, {P, 1}), dL_dcov3D = carve(6, {P, 6});
    if (P != 0) {
        const Tensor m3 = dev_f32(means3D, dev), sc = dev_f32(scales, dev), ro = dev_f32(rotations, dev),
                     cp = dev_f32(cov3D_precomp, dev), vm = dev_f32(viewmatrix, dev), pm = dev_f32(projmatrix, dev),
                     cam = dev_f32(campos, dev), g = dev_f32(dL_dout_color, dev);
        const Tensor rad = radii.contiguous();
        int rc;
        {
            c10::DeviceGuard guard(dev);
            py::gil_scoped_release nogil;
            rc = r2_raster_backward((int)P, (int)R, (int)W, (int)H, fptr(m3), fptr(sc), (float)scale_modifier, fptr(ro), fptr(cp),
                                    fptr(vm), fptr(pm), fptr(cam), (float)tan_fovx, (float)tan_fovy, rad.data_ptr<int>(),
                                    bptr(geomBuffer), bptr(binningBuffer), bptr(imageBuffer), fptr(g),
                                    dL_dmeans2D.data_ptr<float>(), dL_dconic.data_ptr<float>(), dL_dopacity.data_ptr<float>(),
                                    dL_dmu.data_ptr<float>(), dL_dmeans3D.data_ptr<float>(), dL_dcov3D.data_ptr<float>(),
                                    dL_dscales.data_ptr<float>(), dL_drot.data_ptr<float>(), (int)mode, debug ? 1 : 0,
                                    reinterpret_cast<void *>(stream));
        }
        if (rc < 0) fail("r2_raster_backward", rc);
    }
    return {dL_dmeans2D, dL_dopacity, dL_dmu, dL_dmeans3D, dL_dcov3D, dL_dscales, dL_drot};
}

// (num_rendered, out_volume[nx,ny,nz], radii_x, radii_y, radii_z, geomBuffer, binningBuffer, imgBuffer)
// <- SUB/voxelize_points.cu:29-100
std::tuple<int64_t, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> voxelize_gaussians(
    const Tensor &means3D, const Tensor &opacity, const Tensor &scales, const Tensor &rotations, double scale_modifier,
    const Tensor &cov3D_precomp, int64_t nx, int64_t ny, int64_t nz, double sx, double sy, double sz, double cx, double cy,
    double cz, bool prefiltered, bool debug, int64_t stream, int64_t tile_x0, int64_t tile_x1)
{
    // tile layers [tile_x0, tile_x1) along x of the (nx, ny, nz) grid: the whole grid for the reference's call, one x-slab for the
    // sharded query (r2_voxel_forward_slab); the volume returned is the slab's block
    if (means3D.dim() != 2 || means3D.size(1) != 3) throw std::runtime_error("means3D must have dimensions (num_points, 3)");
    require_gpu(means3D, "means3D");
    const int64_t nxs = std::min<int64_t>(tile_x1 * 8, nx) - tile_x0 * 8;
    if (tile_x0 < 0 || nxs <= 0) throw std::runtime_error("voxelize_gaussians: empty or invalid x-slab");
    const c10::Device dev = means3D.device();
    const int64_t P = means3D.size(0);
    const auto bytes = torch::TensorOptions().dtype(torch::kUInt8).device(dev);
    Tensor geom = torch::empty({0}, bytes), binning = torch::empty({0}, bytes), img = torch::empty({0}, bytes);
    if (P == 0) {
        const auto io = means3D.options().dtype(torch::kInt);
        return {0, torch::zeros({nxs, ny, nz}, means3D.options().dtype(torch::kFloat)), torch::zeros({0}, io), torch::zeros({0}, io),
                torch::zeros({0}, io), geom, binning, img};
    }
    Tensor out = torch::empty({nxs, ny, nz}, means3D.options().dtype(torch::kFloat));   // written in full by the combine kernel
    Tensor radii = torch::empty({3, P}, means3D.options().dtype(torch::kInt));         // written in full by the preprocess kernel
    const Tensor m3 = dev_f32(means3D, dev), op = dev_f32(opacity, dev), sc = dev_f32(scales, dev), ro = dev_f32(rotations, dev),
                 cp = dev_f32(cov3D_precomp, dev);
    int *r0 = radii.data_ptr<int>();
    int rc;
    {
        c10::DeviceGuard guard(dev);
        py::gil_scoped_release nogil;
        rc = r2_voxel_forward_slab(resize_state, &geom, resize_state, &binning, resize_state, &img, (int)P, (int)nx, (int)ny, (int)nz,
                                   (float)sx, (float)sy, (float)sz, (float)cx, (float)cy, (float)cz, (int)tile_x0, (int)tile_x1, fptr(m3),
                                   fptr(op), fptr(sc), (float)scale_modifier, fptr(ro), fptr(cp), prefiltered ? 1 : 0,
                                   out.data_ptr<float>(), r0, r0 + P, r0 + 2 * P, debug ? 1 : 0, reinterpret_cast<void *>(stream));
    }
    if (rc < 0) fail("r2_voxel_forward", rc);
    return {rc, out, radii[0], radii[1], radii[2], geom, binning, img};
}

// (dL_dopacity[P,1], dL_dmeans3D[P,3], dL_dcov3D[P,6], dL_dscales[P,3], dL_drotations[P,4])   <- SUB/voxelize_points.cu:102-167
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> voxelize_gaussians_backward(
    const Tensor &means3D, const Tensor &radii_x, const Tensor &radii_y, const Tensor &radii_z, const Tensor &scales,
    const Tensor &rotations, double scale_modifier, const Tensor &cov3D_precomp, const Tensor &dL_dout, const Tensor &geomBuffer,
    int64_t R, const Tensor &binningBuffer, const Tensor &imageBuffer, int64_t nx, int64_t ny, int64_t nz, double sx, double sy,
    double sz, double cx, double cy, double cz, bool debug, int64_t stream, int64_t tile_x0, int64_t tile_x1)
{
    require_gpu(means3D, "means3D");
    const c10::Device dev = means3D.device();
    const int64_t P = means3D.size(0);
    Tensor flat = torch::empty({26 * P}, means3D.options().dtype(torch::kFloat));   // every row is written by the kernels
    int64_t o = 0;
    auto carve = [&](int64_t k) {
        Tensor v = flat.narrow(0, o, k * P).view({P, k});
        o += k * P;
        return v;
    };
    Tensor dL_drot = carve(4), dL_dmeans3D = carve(3), dL_dnorm = carve(3), dL_dconic3D = carve(6), dL_dopacity = carve(1),
           dL_dcov3D = carve(6), dL_dscales = carve(3);
    if (P != 0) {
        const Tensor m3 = dev_f32(means3D, dev), sc = dev_f32(scales, dev), ro = dev_f32(rotations, dev),
                     cp = dev_f32(cov3D_precomp, dev), g = dev_f32(dL_dout, dev);
        const Tensor rx = radii_x.contiguous(), ry = radii_y.contiguous(), rz = radii_z.contiguous();
        int rc;
        {
            c10::DeviceGuard guard(dev);
            py::gil_scoped_release nogil;
            rc = r2_voxel_backward_slab((int)P, (int)R, (int)nx, (int)ny, (int)nz, (float)sx, (float)sy, (float)sz, (float)cx, (float)cy,
                                        (float)cz, (int)tile_x0, (int)tile_x1, fptr(m3), fptr(sc), (float)scale_modifier, fptr(ro),
                                        fptr(cp), rx.data_ptr<int>(), ry.data_ptr<int>(), rz.data_ptr<int>(), bptr(geomBuffer),
                                        bptr(binningBuffer), bptr(imageBuffer), fptr(g), dL_dnorm.data_ptr<float>(),
                                        dL_dconic3D.data_ptr<float>(), dL_dopacity.data_ptr<float>(), dL_dmeans3D.data_ptr<float>(),
                                        dL_dcov3D.data_ptr<float>(), dL_dscales.data_ptr<float>(), dL_drot.data_ptr<float>(),
                                        debug ? 1 : 0, reinterpret_cast<void *>(stream));
        }
        if (rc < 0) fail("r2_voxel_backward", rc);
    }
    return {dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dscales, dL_drot};
}

}  // namespace

PYBIND11_MODULE(_r2shim, m)
{
    m.doc() = "torch boundary of libr2hip.so (see r2_gaussian_amd/_C.py)";
    m.def("abi_version", []() { return r2_abi_version(); });
    m.def("rasterize_gaussians", &rasterize_gaussians);
    m.def("rasterize_gaussians_backward", &rasterize_gaussians_backward);
    m.def("voxelize_gaussians", &voxelize_gaussians);
    m.def("voxelize_gaussians_backward", &voxelize_gaussians_backward);
}
