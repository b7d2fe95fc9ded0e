"""Object -> rank sharding and the final-render gather (the only collective on the path).

The reference assigns object k to device k mod nGPU inside one process (CORE/src/nerf.cu:27-33) and has no
inter-GPU communication at all; here it is one process per GPU (torch.distributed, backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests) with the same round-robin map.  Training needs no collective:
object NeRFs are independent units.  Rendered crops (rgb[3hw] + depth[hw] + mask[hw] float32, variable h x w)
are gathered to the ROOT rank, which composites / writes the images: sizes first, then one point-to-point message of the true size per peer."""
import numpy as np


def objects_of_rank(n_objects, world, rank):
    """Round-robin object -> rank map (nerf.cu:27-33: curGPUid = (curGPUid + 1) % GPUnum)."""
    return [k for k in range(n_objects) if k % world == rank]


def owner_of_object(k, world):
    return k % world


def pack_crop(rgb, depth, mask):
    h, w = depth.shape
    return np.concatenate([np.array([h, w], np.float32), rgb.reshape(-1), depth.reshape(-1), mask.reshape(-1)]).astype(np.float32)


def unpack_crop(buf):
    h, w = int(buf[0]), int(buf[1]); n = h * w
    rgb = buf[2:2 + 3 * n].reshape(h, w, 3); depth = buf[2 + 3 * n:2 + 4 * n].reshape(h, w); mask = buf[2 + 4 * n:2 + 5 * n].reshape(h, w)
    return rgb, depth, mask


def render_packed(obj, box, pose16, torch, device, pose_is_Toc=False):
    """Renders one crop of `obj` straight into a packed float32 tensor [h, w, rgb(3hw), depth(hw), mask(hw)] that lives on the
    collective's device: HBM for RCCL (mon_object_render with dst_on_device=1 -- the crop never visits the host before the
    gather), host memory for gloo."""
    h, w = int(box[3]), int(box[4]); n = h * w
    on_dev = str(device) != "cpu"
    buf = torch.empty(2 + 5 * n, dtype=torch.float32, device=device)
    buf[:2] = torch.tensor([float(h), float(w)], dtype=torch.float32)
    if on_dev:
        torch.cuda.synchronize()             # the header write runs on torch's stream, the render on the object's own
    base = buf.data_ptr() + 8
    obj.render_into(box, pose16, base, base + 12 * n, base + 16 * n, on_dev, pose_is_Toc)
    return buf


def _pack_rank(torch, crops, device):
    """One float32 message per rank: [n_crops, size_0, crop_0..., size_1, crop_1...] -- a GPU's objects / views travel as one message per frame."""
    f32 = lambda a: torch.tensor(a, dtype=torch.float32, device=device)
    parts = [f32([float(len(crops))])]
    for c in crops:
        t = c.to(device) if torch.is_tensor(c) else torch.from_numpy(np.ascontiguousarray(c, np.float32)).to(device)
        parts += [f32([float(t.numel())]), t]
    return torch.cat(parts)


def _unpack_rank(msg):
    a = msg.cpu().numpy(); k = int(a[0]); p = 1; items = []
    for _ in range(k):
        sz = int(a[p]); items.append(unpack_crop(a[p + 1:p + 1 + sz])); p += 1 + sz
    return items


def gather_crops(dist, torch, crops, device, root=0):
    """Gather-to-root of the final render: crops = packed float32 crops rendered by this rank -- numpy arrays (pack_crop) or tensors already
    on `device` (render_packed: HBM for RCCL).  On `root` returns the list (over ranks) of lists of (rgb, depth, mask); on every other rank None.
    Two steps: the message sizes are gathered to the root (one int64 per rank), then every peer SENDS its message of its true size straight to the root
    and the root posts all receives together (one grouped batch of point-to-point operations: on the 8-GPU xGMI mesh every peer has a direct link to
    the root, so the 7 transfers run side by side, ~153 GB/s each, no ring and no padding).  Only the root unpacks (one device-to-host copy per peer)."""
    world = dist.get_world_size(); rank = dist.get_rank()
    mine = _pack_rank(torch, crops, device)
    n = torch.tensor([mine.numel()], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)] if rank == root else None
    dist.gather(n, sizes, dst=root)
    if rank != root:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, mine, root)]):
            w.wait()
        return None
    bufs = [mine if r == root else torch.empty(int(sizes[r].item()), dtype=torch.float32, device=device) for r in range(world)]
    ops = [dist.P2POp(dist.irecv, bufs[r], r) for r in range(world) if r != root]
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return [_unpack_rank(b) for b in bufs]


def loopback_crop(dist, torch, packed):
    """A packed crop sent from this rank to itself through the backend's point-to-point path (one grouped send + receive): what a peer's message to the
    root goes through, runnable with a single rank -- the 1-GPU check that the RCCL transport moves a device-resident crop unchanged."""
    me = dist.get_rank(); out = torch.empty_like(packed)
    for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, packed, me), dist.P2POp(dist.irecv, out, me)]):
        w.wait()
    return out


def max_over_ranks(dist, torch, value, device):
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
