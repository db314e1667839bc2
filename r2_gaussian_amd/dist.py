"""View-sharded data parallelism for R2-Gaussian training on one MI355X node (new functionality: the
reference is single-GPU).

Every rank holds a full replica of the Gaussians and renders its own training view(s); one exchange step per
optimiser step sums the per-Gaussian gradients over ranks.  The four parameter gradients
(xyz[P,3], density[P,1], scaling[P,3], rotation[P,4]) travel as ONE flat [P,11] float32 buffer = 44 B per
Gaussian (13 MB at 300k) so that a step costs a single RCCL all-reduce over xGMI; the densification
statistics (train.py:151-154, gaussian_model.py:552-556) are reduced with the matching semantics
(sum of the 2D-gradient norms and visibility counts, max of the screen radii) so that densify/prune takes
bit-identical decisions on every rank.

Works with any torch.distributed backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" on CPU (tests).
"""
import torch
import torch.distributed as dist

GRAD_WIDTH = 11   # 3 + 1 + 3 + 4


def pin_to_gpu_numa_node(device_index=0, max_cpus=16, rank_slot=None):
    """Restrict this process to (a compact slice of) the CPUs of the NUMA node the GPU hangs off -- one process per GPU,
    each on its own socket's cores.  The host side of a training view is ~20 kernel launches and one busy-wait on a device->host read; on the
    2-socket host of an MI355X node a process that lands on the remote socket runs ~20 % slower (measured).  Returns the
    CPU set it pinned to, or None when the topology cannot be read (then nothing is changed)."""
    import os
    try:
        props = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
        path = "/sys/bus/pci/devices/%s/local_cpulist" % bdf
        with open(path) as f:
            txt = f.read().strip()
        cpus = set()
        for part in txt.split(","):
            if "-" in part:
                a, b = part.split("-")
                cpus.update(range(int(a), int(b) + 1))
            elif part:
                cpus.add(int(part))
        allowed = os.sched_getaffinity(0)
        cpus = sorted(cpus & allowed)
        if not cpus:
            return None
        # a compact slice of the node: the main thread, the autograd engine thread and the runtime's helpers then
        # share caches instead of migrating over 128 hardware threads (measured: +10 % views/s over node-wide pinning);
        # ranks of one node take disjoint slices
        slot = device_index if rank_slot is None else rank_slot
        nslots = max(1, len(cpus) // max_cpus)
        lo = (slot % nslots) * max_cpus
        pick = set(cpus[lo:lo + max_cpus]) or set(cpus)
        os.sched_setaffinity(0, pick)
        return pick
    except Exception:
        return None


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def view_for(step, n_views, perm=None, rank_=None, world_=None):
    """Index of the training view rank r renders at optimiser step k: perm[(k*world + r) mod n_views]
    (the ragged tail wraps around, like the reference's refill-when-empty view stack, train.py:104-106)."""
    r = rank() if rank_ is None else rank_
    w = world() if world_ is None else world_
    i = (step * w + r) % n_views
    return int(perm[i]) if perm is not None else i


def pack_grads(g_xyz, g_density, g_scaling, g_rotation, out=None):
    """-> flat [P,11] buffer (xyz | density | scaling | rotation)."""
    P = g_xyz.shape[0]
    if out is None:
        out = torch.empty((P, GRAD_WIDTH), dtype=g_xyz.dtype, device=g_xyz.device)
    out[:, 0:3] = g_xyz
    out[:, 3:4] = g_density.reshape(P, 1)
    out[:, 4:7] = g_scaling
    out[:, 7:11] = g_rotation
    return out


def grad_block(g_xyz, g_density, g_scaling, g_rotation):
    """The drop-in backward carves its gradients out of one buffer with rotation | xyz | scaling | density adjacent: if
    these four tensors are that block, return it as ONE flat [11 P] view (all-reduce it in place: no packing copy, the
    .grad tensors are reduced where they are); otherwise None (use pack_grads)."""
    P = g_xyz.shape[0]
    ts = (g_rotation, g_xyz, g_scaling, g_density)
    if P == 0 or any(t is None or not t.is_contiguous() or t.dtype != torch.float32 for t in ts):
        return None
    try:
        if len({t.untyped_storage().data_ptr() for t in ts}) != 1:
            return None
    except Exception:
        return None
    o = g_rotation.storage_offset()
    if (g_xyz.storage_offset(), g_scaling.storage_offset(), g_density.storage_offset()) != (o + 4 * P, o + 7 * P, o + 10 * P):
        return None
    if g_rotation.numel() != 4 * P or g_xyz.numel() != 3 * P or g_scaling.numel() != 3 * P or g_density.numel() != P:
        return None
    return g_rotation.as_strided((GRAD_WIDTH * P,), (1,), o)


def unpack_grads(flat):
    return flat[:, 0:3], flat[:, 3:4], flat[:, 4:7], flat[:, 7:11]


def allreduce_grads(flat, average=True, async_op=False):
    """Sum (or mean) the packed gradients over ranks, in place.  One collective per optimiser step.
    async_op=True returns the Work handle (call .wait() before touching ``flat``); the sum is then left un-averaged."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    if world() == 1 and not async_op:
        return None
    h = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op)
    if average and not async_op:
        flat.div_(world())
    return h


def allreduce_param_grads(params, average=True):
    """params = (xyz, density, scaling, rotation) leaf tensors with .grad set: exchange + write back."""
    flat = pack_grads(*(p.grad for p in params))
    allreduce_grads(flat, average=average)
    for p, g in zip(params, unpack_grads(flat)):
        p.grad.copy_(g.reshape(p.grad.shape))
    return flat


def allreduce_densify_stats(grad_norm_inc, denom_inc, radii):
    """Densification statistics of one step, reduced over the views rendered by all ranks:
    grad_norm_inc[P] (||d L/d means2D[:, :2]|| where visible, else 0) and denom_inc[P] (visibility count) are
    summed; radii[P] (screen radius, 0 where culled) is max-reduced for max_radii2D."""
    if world() == 1:
        return grad_norm_inc, denom_inc, radii
    both = torch.stack([grad_norm_inc.float(), denom_inc.float()], 0)
    dist.all_reduce(both, op=dist.ReduceOp.SUM)
    rmax = radii.clone()
    dist.all_reduce(rmax, op=dist.ReduceOp.MAX)
    return both[0], both[1], rmax


def assert_replicas_equal(t, what="tensor"):
    """Cheap consistency check after densify/prune: every rank must hold the same P and the same bytes."""
    if world() == 1:
        return
    sig = torch.tensor([float(t.numel()), float(t.double().sum()), float(t.double().abs().sum())],
                       dtype=torch.float64, device=t.device)
    lo, hi = sig.clone(), sig.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if not torch.equal(lo, hi):
        raise RuntimeError("replica divergence detected in %s: %s vs %s" % (what, lo.tolist(), hi.tolist()))


# ---- full-volume query sharded by x-slab (independent units, no exchange, optional gather to assemble) ----------
# The volume array is [nx, ny, nz] with x slowest, so a slab of whole 8-voxel tiles along x is a contiguous block of
# memory.  A rank's call is the FULL grid's call restricted to its tile layers (r2_voxel_forward_slab): same centre, same voxel
# size, same voxel coordinates and radii on every rank, so the concatenated slabs are bit-identical to the unsharded volume.
# (Rounds 1-5 re-centred a sub-volume per rank: voxel coordinates were then rounded differently and a 1e-4 share of the voxels
# differed by more than 1e-4 relative.)
TILE3D = 8


def slab_bounds(n_voxel_x, rank_, world_):
    """[x0, x1) of rank_'s slab: whole tiles, the first ``tiles % world`` ranks get one tile more."""
    tiles = (int(n_voxel_x) + TILE3D - 1) // TILE3D
    per, extra = divmod(tiles, int(world_))
    t0 = rank_ * per + min(rank_, extra)
    t1 = t0 + per + (1 if rank_ < extra else 0)
    return min(t0 * TILE3D, int(n_voxel_x)), min(t1 * TILE3D, int(n_voxel_x))


def slab_settings(settings, rank_=None, world_=None):
    """``GaussianVoxelizationSlabSettings`` of this rank's x-slab of the volume described by ``settings``: the full volume's
    settings + the rank's range of tile layers (None if the slab is empty: more ranks than tile layers)."""
    from .voxelization import GaussianVoxelizationSlabSettings
    r = rank() if rank_ is None else rank_
    w = world() if world_ is None else world_
    x0, x1 = slab_bounds(settings.nVoxel_x, r, w)
    if x1 <= x0:
        return None, (x0, x1)
    base = tuple(settings)[:12]
    return GaussianVoxelizationSlabSettings(*base, tile_x0=x0 // TILE3D, tile_x1=(x1 + TILE3D - 1) // TILE3D), (x0, x1)


def query_sharded(voxelizer_cls, settings, means3D, opacities, scales, rotations, gather=True):
    """Every rank voxelizes its x-slab (no exchange between the slabs); with ``gather`` the slabs are all-gathered into
    the full [nx, ny, nz] volume on every rank, otherwise ``(slab, (x0, x1))`` is returned."""
    sub, (x0, x1) = slab_settings(settings)
    if sub is not None:
        vol, _radii = voxelizer_cls(sub)(means3D=means3D, opacities=opacities, scales=scales, rotations=rotations)
    else:
        vol = means3D.new_zeros((0, settings.nVoxel_y, settings.nVoxel_z))
    if not gather or world() == 1:
        return (vol, (x0, x1)) if not gather else vol
    import torch
    import torch.distributed as dist
    full = means3D.new_empty((settings.nVoxel_x, settings.nVoxel_y, settings.nVoxel_z))
    # slabs differ in size by at most one tile layer: gather into per-rank views of the full volume
    parts = []
    for r in range(world()):
        a, b = slab_bounds(settings.nVoxel_x, r, world())
        parts.append(full[a:b])
    if all(p.shape == parts[0].shape for p in parts):
        dist.all_gather(parts, vol.contiguous())
    else:   # ragged: pad to the largest slab
        m = max(p.shape[0] for p in parts)
        pad = vol.new_zeros((m, settings.nVoxel_y, settings.nVoxel_z))
        pad[:vol.shape[0]] = vol
        bufs = [torch.empty_like(pad) for _ in parts]
        dist.all_gather(bufs, pad)
        for p, bsrc in zip(parts, bufs):
            p.copy_(bsrc[:p.shape[0]])
    return full
