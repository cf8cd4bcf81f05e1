"""Multi-GPU sharding of the hot path, one process per GPU over torch.distributed (nccl = RCCL on ROCm).

Two independent partitions, as in the reference's own parallel structure (one extractor per camera thread,
src/cMultiFrame.cpp:128-164; keyframes are independent in the brute-force database searches):

  * extraction   camera c of every multi-frame  -> rank  c % world           (camera_shard)
  * matching     stored keyframe k              -> rank  k % world           (keyframe_shard)

and exactly ONE exchange step between them: every rank needs the descriptors (+ masks, + per-camera counts) of ALL cameras
of the current multi-frames before it can match them against its keyframe shard.  That is an all-gather of fixed-size
blocks (cap rows per camera, zero-padded; ranks owning fewer cameras pad to ceil(ncam/world) blocks) — a latency-bound
message of tens of KiB per GPU, done with one all_gather_into_tensor per tensor.  Match results stay sharded by keyframe.
torch is plumbing here (buffers + process group); the compute is libmcs_hip.so.
"""
import torch
import torch.distributed as dist


def camera_shard(ncam, rank, world):
    return [c for c in range(ncam) if c % world == rank]


def keyframe_shard(nkf, rank, world):
    return [k for k in range(nkf) if k % world == rank]


def cams_per_rank(ncam, world):
    return (ncam + world - 1) // world


def allgather_rig(desc, dmask, nkp, ncam, rank, world, group=None):
    """desc/dmask: [F, local_cams, cap, ds] uint8, nkp: [F, local_cams] int32 for the cameras camera_shard(ncam, rank, world).
    Returns (desc_all [F, ncam, cap, ds], dmask_all, nkp_all [F, ncam]) in global camera order on every rank."""
    F, lc, cap, ds = desc.shape
    cpr = cams_per_rank(ncam, world)
    if world == 1:
        return desc, dmask, nkp
    dev = desc.device

    def pad(t, fill=0):
        if lc == cpr:
            return t.contiguous()
        shape = list(t.shape)
        shape[1] = cpr
        out = torch.full(shape, fill, dtype=t.dtype, device=dev)
        out[:, :lc] = t
        return out

    send_d, send_m, send_n = pad(desc), pad(dmask), pad(nkp)
    host_bounce = send_d.is_cuda and dist.get_backend(group) == "gloo"   # functional tests of the N>1 path on one GPU

    def gather(send):   # concatenated-along-dim-0 output form (accepted by both nccl/RCCL and gloo), viewed as [world, ...]
        if host_bounce:
            send = send.cpu()
        out = torch.empty((world * send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        dist.all_gather_into_tensor(out, send, group=group)
        return out.view((world,) + tuple(send.shape)).to(dev)

    out_d, out_m, out_n = gather(send_d), gather(send_m), gather(send_n)
    # rank r, slot j  ->  camera r + j*world   (camera_shard order)
    desc_all = torch.zeros((F, ncam, cap, ds), dtype=desc.dtype, device=dev)
    mask_all = torch.zeros((F, ncam, cap, ds), dtype=dmask.dtype, device=dev)
    nkp_all = torch.zeros((F, ncam), dtype=nkp.dtype, device=dev)
    for r in range(world):
        for j, c in enumerate(camera_shard(ncam, r, world)):
            desc_all[:, c] = out_d[r, :, j]
            mask_all[:, c] = out_m[r, :, j]
            nkp_all[:, c] = out_n[r, :, j]
    return desc_all, mask_all, nkp_all


def reduce_timing(elapsed_s, units, device, world):
    """max-over-ranks time and summed units (bench contract)."""
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    u = torch.tensor([units], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())
