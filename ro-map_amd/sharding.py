"""Object -> rank sharding and the final-render gather (the only collective on the path).

The reference assigns object k to device k mod nGPU inside one process (CORE/src/nerf.cu:27-33) and has no
inter-GPU communication at all; here it is one process per GPU (torch.distributed, backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests) with the same round-robin map.  Training needs no collective:
object NeRFs are independent units.  Rendered crops (rgb[3hw] + depth[hw] + mask[hw] float32, variable h x w)
are gathered to every rank with one padded all_gather per call."""
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


def gather_crops(dist, torch, crops, device):
    """crops: list of packed float32 crops rendered by this rank -- numpy arrays (pack_crop) or tensors already on `device`
    (render_packed).  Returns, on every rank, the list (over ranks) of lists of (rgb, depth, mask).  One size all_gather + one
    padded payload all_gather; tensors given on the device stay there until the gathered payload is unpacked."""
    world = dist.get_world_size()
    f32 = lambda a: torch.tensor(a, dtype=torch.float32, device=device)
    parts = [f32([float(len(crops))])]
    for c in crops:
        t = c.to(device) if torch.is_tensor(c) else torch.from_numpy(np.ascontiguousarray(c, np.float32)).to(device)
        parts += [f32([float(t.numel())]), t]
    mine = torch.cat(parts)
    n = torch.tensor([mine.numel()], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    cap = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros(cap, dtype=torch.float32, device=device); pad[: mine.numel()] = mine
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    out = []
    for b in bufs:
        a = b.cpu().numpy(); k = int(a[0]); p = 1; items = []
        for _ in range(k):
            sz = int(a[p]); items.append(unpack_crop(a[p + 1:p + 1 + sz])); p += 1 + sz
        out.append(items)
    return out


def max_over_ranks(dist, torch, value, device):
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
