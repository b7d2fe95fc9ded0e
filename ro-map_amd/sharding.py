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


def gather_crops(dist, torch, crops, device):
    """crops: list of packed float32 arrays rendered by this rank.  Returns, on every rank, the list (over ranks)
    of lists of (rgb, depth, mask).  One size all_gather + one padded payload all_gather."""
    world = dist.get_world_size()
    parts = [np.array([len(crops)], np.float32)]
    for c in crops:
        parts += [np.array([c.size], np.float32), c.astype(np.float32)]
    mine = np.concatenate(parts)
    n = torch.tensor([mine.size], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    cap = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros(cap, dtype=torch.float32, device=device); pad[: mine.size] = torch.from_numpy(mine).to(device)
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
