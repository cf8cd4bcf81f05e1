"""Multi-GPU split of the hot path (BASELINE north_star): one process per GPU over torch.distributed (backend nccl = RCCL over xGMI).

  extraction   the (camera, frame) images of a step are sharded over the ranks, camera-major: image x = camera * frames_total + frame, rank r owns the
               contiguous slab x in [r*L, (r+1)*L), L = ncam * frames_total / world.  The cameras of one multi-frame therefore sit on different
               GPUs as soon as world > 1 — the reference's own parallel axis (one extractor per camera thread, src/cMultiFrame.cpp:128-164) — and
               the slabs are equal for every camera count and world size.
  exchange     ONE all-gather per step.  A rank's send buffer is L image blocks, each (cap + 1) rows of 2*descSize bytes: row k = descriptor | mask of
               keypoint k (the extractor writes them interleaved, mcs_extract_batch_strided), row cap = header holding the image's keypoint count
               (mcs_rig_pack_headers).  Because the slabs are contiguous in x, the gathered buffer IS the global [camera][frame][row] array — nothing
               is permuted or copied afterwards.
  matching     a multi-frame is a block-structured descriptor set inside that array (mcs_desc_set.block_rows = cap, block_pitch_rows =
               frames_total * (cap + 1); consecutive frames lie cap + 1 rows apart), consumed in place by the searches.  The (frame, keyframe) pairs
               are sharded: stored keyframe k -> rank k % world (database sweeps, BASELINE configs[2]/[4]), or frame f -> rank f // F when every frame
               is matched against its predecessor (configs[1]).  Match results stay on the rank that produced them.

RigLayout is pure index arithmetic (shared by bench.py and the world-size-2 gloo test, which runs it with the oracle as compute); torch is plumbing
(buffers + the process group), the compute is libmcs_hip.so.
"""
import numpy as np


class RigLayout:
    def __init__(self, ncam, frames_total, world, cap, desc_size=32):
        if (ncam * frames_total) % world:
            raise ValueError("ncam * frames_total must be a multiple of the world size")
        if frames_total % world:
            raise ValueError("frames_total must be a multiple of the world size (frame_pairs / the ring call give every rank frames_total / world frames)")
        self.ncam, self.frames_total, self.world, self.cap, self.desc_size = ncam, frames_total, world, cap, desc_size
        self.images_total = ncam * frames_total
        self.L = self.images_total // world            # images per rank
        self.rows_img = cap + 1                        # descriptor rows + the header row
        self.row_stride = 2 * desc_size                # descriptor | mask
        self.block_bytes = self.rows_img * self.row_stride
        self.send_bytes = self.L * self.block_bytes
        self.rows_frame = ncam * cap                   # logical rows of one multi-frame (what match indices refer to)

    # ---- extraction shard
    def slab(self, rank):
        """(camera, frame) of the images rank `rank` extracts, in send order"""
        return [(x // self.frames_total, x % self.frames_total) for x in range(rank * self.L, (rank + 1) * self.L)]

    def image_index(self, cam, frame):
        return cam * self.frames_total + frame

    # ---- a multi-frame inside the gathered array
    def frame_desc_set(self, frame):
        """byte offsets and geometry of multi-frame `frame` in the gathered array / the valid array:
        (desc_off, mask_off, valid_off, n, stride, block_rows, block_pitch_rows, set_pitch_rows)"""
        return (frame * self.block_bytes, frame * self.block_bytes + self.desc_size, frame * self.rows_img, self.rows_frame, self.row_stride,
                self.cap, self.frames_total * self.rows_img, self.rows_img)

    def frame_rows(self, frame):
        """physical row (of rows_img-row image blocks) of every logical row of multi-frame `frame`: the same mapping mcs_desc_set's block fields express"""
        i = np.arange(self.rows_frame)
        return frame * self.rows_img + (i // self.cap) * (self.frames_total * self.rows_img) + i % self.cap

    # ---- matching shards
    def keyframe_shard(self, nkf, rank):
        return [k for k in range(nkf) if k % self.world == rank]

    def frame_pairs(self, rank):
        """(frame, predecessor) pairs rank `rank` matches when every multi-frame meets the one before it (cyclic within the step)"""
        F = self.frames_total // self.world
        return [(f, (f - 1) % self.frames_total) for f in range(rank * F, (rank + 1) * F)]


class RingExchange:
    """What a rank RECEIVES when every multi-frame only meets its predecessor (BASELINE configs[1]): rank r matches frames [r*F, (r+1)*F), so it needs the
    camera blocks of frames r*F - 1 .. (r+1)*F - 1 and nothing else — 1/world of what the all-gather delivers.  The receive buffer is the local array
    [camera][F + 1 frames][cap + 1 rows] (local frame 0 = the predecessor of the rank's first frame, cyclic), again consumed in place: `view` is the
    RigLayout of that array (world 1, frames_total F + 1), so frame_desc_set / frame_rows / unpack_frame apply unchanged with LOCAL frame numbers and
    mcs_search_kf_kf_ring runs on it with first = 1, count = F.  The transfers are runs of consecutive image blocks (consecutive frames of one camera lie
    next to each other both in the owner's send buffer and in the receiver's array): `recvs(r)` / `sends(r)` list them as
    (peer, offset in MY buffer in blocks, number of blocks), identically ordered on both sides — point-to-point sends grouped into one exchange step
    (ncclSend / ncclRecv inside ncclGroupStart / End; torch: batch_isend_irecv), runs a rank owns itself are plain copies.
    Blocks are sent whole (cap rows, not the image's keypoint count): a count-sized transfer would need the counts on the host, i.e. a device-to-host round
    trip per step; with cap = nfeatures + 3 * nlevels the padding is 2 % here."""

    def __init__(self, layout):
        lay = layout
        self.lay = lay
        self.F = lay.frames_total // lay.world
        self.view = RigLayout(lay.ncam, self.F + 1, 1, lay.cap, lay.desc_size)

    def global_frame(self, rank, j):
        """local frame j of rank `rank` shows this global frame"""
        return (rank * self.F - 1 + j) % self.lay.frames_total

    def _runs(self, rank):
        """(owner, first block in the owner's send buffer, first block in rank's local array, blocks) for everything rank needs"""
        lay, out = self.lay, []
        for c in range(lay.ncam):
            j = 0
            while j <= self.F:
                x = lay.image_index(c, self.global_frame(rank, j))
                owner = x // lay.L
                n = 1   # extend the run while the global image index stays consecutive and inside the owner's slab
                while j + n <= self.F and lay.image_index(c, self.global_frame(rank, j + n)) == x + n and (x + n) // lay.L == owner:
                    n += 1
                out.append((owner, x - owner * lay.L, c * (self.F + 1) + j, n))
                j += n
        return out

    def recvs(self, rank):
        return [(owner, dst, n) for owner, _, dst, n in self._runs(rank)]

    def sends(self, rank):
        return [(dest, src, n) for dest in range(self.lay.world) for owner, src, _, n in self._runs(dest) if owner == rank]

    def bytes_received(self, rank):
        return sum(n for owner, _, n in self.recvs(rank) if owner != rank) * self.lay.block_bytes


def plan_check(ncam, frames_per_rank, world, cap, keyframes, desc_size=32, topk=32):
    """Everything a multi-GPU run of a workload relies on, checked WITHOUT a GPU (bench.py --dry-run, tests/test_rig_gloo.py): the slabs partition the images;
    every transfer of the exchange stays inside the sender's send buffer and the receiver's array, arrives exactly once and pairs up with a send of the same
    size in the same order; every (frame, keyframe) / (frame, predecessor) pair has exactly one owner.  Returns the plan's sizes (bytes per rank); raises
    ValueError naming the first violated property."""
    FT = frames_per_rank * world
    lay = RigLayout(ncam, FT, world, cap, desc_size)

    def need(cond, what):
        if not cond:
            raise ValueError("rig plan (ncam %d, frames/rank %d, world %d, keyframes %d): %s" % (ncam, frames_per_rank, world, keyframes, what))
    slabs = [lay.slab(r) for r in range(world)]
    need(all(len(sl) == lay.L for sl in slabs), "unequal slabs")
    need(sum(slabs, []) == [(c, f) for c in range(ncam) for f in range(FT)], "the slabs are not the contiguous camera-major partition of the images")
    out = {"world": world, "images_per_rank": lay.L, "send_bytes_per_rank": lay.send_bytes, "block_bytes": lay.block_bytes}
    if keyframes == 0:
        ex = RingExchange(lay)
        view_blocks = ncam * (ex.F + 1)
        for r in range(world):
            seen = {}
            for owner, src, dst, n in ex._runs(r):
                need(0 <= owner < world and n >= 1, "bad run")
                need(0 <= src and src + n <= lay.L, "a send run leaves rank %d's send buffer" % owner)
                need(0 <= dst and dst + n <= view_blocks, "a receive run leaves rank %d's local array" % r)
                for i in range(n):
                    need(dst + i not in seen, "block %d of rank %d's array is written twice" % (dst + i, r))
                    seen[dst + i] = owner * lay.L + src + i
            need(sorted(seen) == list(range(view_blocks)), "rank %d's array is not covered" % r)
            need(all(seen[c * (ex.F + 1) + j] == lay.image_index(c, ex.global_frame(r, j)) for c in range(ncam) for j in range(ex.F + 1)), "rank %d receives the wrong image somewhere" % r)
        for a in range(world):
            for b in range(world):
                need([n for d, _, n in ex.sends(a) if d == b] == [n for o, _, n in ex.recvs(b) if o == a], "sends of rank %d and receives of rank %d do not pair up" % (a, b))
        pairs = sum((lay.frame_pairs(r) for r in range(world)), [])
        need(sorted(f for f, _ in pairs) == list(range(FT)) and all(p == (f - 1) % FT for f, p in pairs), "a (frame, predecessor) pair is owned twice or by nobody")
        for r in range(world):   # the pairs a rank owns only read frames its local array holds
            have = {ex.global_frame(r, j) for j in range(ex.F + 1)}
            need(all(f in have and pp in have for f, pp in lay.frame_pairs(r)), "rank %d matches a frame it did not receive" % r)
        out.update(exchange="point-to-point frame ring", recv_bytes_per_rank=[ex.bytes_received(r) for r in range(world)], array_bytes_per_rank=view_blocks * lay.block_bytes,
                   pairs_per_rank=[len(lay.frame_pairs(r)) for r in range(world)], match_rows_per_rank=ex.F * lay.rows_frame,
                   topk_list_bytes_per_rank=ex.F * lay.rows_frame * topk * 4)
    else:
        shards = [lay.keyframe_shard(keyframes, r) for r in range(world)]
        need(sorted(sum(shards, [])) == list(range(keyframes)), "a stored keyframe is owned twice or by nobody")
        owned = {}
        for r in range(world):
            for k in shards[r]:
                for f in range(FT):
                    need((f, k) not in owned, "pair (frame %d, keyframe %d) is owned twice" % (f, k))
                    owned[(f, k)] = r
        need(len(owned) == FT * keyframes, "a (frame, keyframe) pair has no owner")
        need(world * lay.send_bytes == lay.images_total * lay.block_bytes, "the all-gather's output is not the global array")
        out.update(exchange="all-gather", recv_bytes_per_rank=[(world - 1) * lay.send_bytes] * world, array_bytes_per_rank=lay.images_total * lay.block_bytes,
                   pairs_per_rank=[FT * len(sh) for sh in shards], match_rows_per_rank=max(FT * len(sh) for sh in shards) * lay.rows_frame,
                   topk_list_bytes_per_rank=max(FT * len(sh) for sh in shards) * lay.rows_frame * topk * 4)
    need(out["array_bytes_per_rank"] < 2 ** 40 and out["topk_list_bytes_per_rank"] < 200 * 2 ** 30, "a buffer exceeds what a 288 GB GPU can hold")
    return out


_FNV_BASES = (0xCBF29CE484222325, 0x84222325CBF29CE4, 0x9E3779B97F4A7C15, 0xC2B2AE3D27D4EB4F)


def plan_digest(ncam, frames_per_rank, world, cap, keyframes, desc_size=32, topk=32):
    """32 bytes that name the WHOLE plan of a multi-GPU run — the arguments, every rank's slab, every transfer run of the ring exchange (owner, source block,
    destination block, length) or every rank's keyframe shard, every (frame, predecessor) pair and its owner — as four 64-bit FNV-1a hashes (different offset
    bases) over the plan's integers as little-endian int64.  Every rank computes it from ITS OWN arguments before the first exchange and the ranks compare
    (check_plan_digests): a rank started with another frame count, world size or library build says so before any buffer is wrong.  host/rig_host.cpp computes the
    same bytes from its C++ restatement of the layout (tests/test_rig_plan_digest.py compares the two), `bench.py --dry-run` prints them per workload."""
    lay = RigLayout(ncam, frames_per_rank * world, world, cap, desc_size)
    ints = [ncam, frames_per_rank, world, cap, keyframes, desc_size, topk, lay.L, lay.rows_img, lay.row_stride, lay.block_bytes, lay.send_bytes]
    for r in range(world):
        for c, f in lay.slab(r):
            ints += [c, f]
    if keyframes == 0:
        ex = RingExchange(lay)
        for r in range(world):
            for run in ex._runs(r):
                ints += list(run)
            for f, p in lay.frame_pairs(r):
                ints += [f, p]
    else:
        for r in range(world):
            sh = lay.keyframe_shard(keyframes, r)
            ints += [len(sh)] + sh
    data = np.asarray(ints, "<i8").tobytes()
    out = b""
    for h in _FNV_BASES:
        for byte in data:
            h = ((h ^ byte) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
        out += h.to_bytes(8, "little")
    return out


def check_plan_digests(mine, rank, world, all_gather):
    """compare the plan digests of all ranks (`all_gather`: 32 bytes -> world * 32 bytes, rank-major); returns the hex digest, raises ValueError naming the ranks whose
    digest differs from the one most ranks hold"""
    got = bytes(all_gather(mine))
    if len(got) != 32 * world:
        raise ValueError("plan digest exchange returned %d bytes, expected %d" % (len(got), 32 * world))
    per = [got[32 * r:32 * (r + 1)] for r in range(world)]
    major = max(set(per), key=per.count)
    odd = [r for r in range(world) if per[r] != major]
    if odd:
        raise ValueError("rig plan digest mismatch: rank(s) %s hold %s, the others %s (rank %d holds %s) — different arguments, world size or build on those ranks"
                         % (odd, sorted({per[r].hex()[:16] for r in odd}), major.hex()[:16], rank, mine.hex()[:16]))
    return major.hex()


def ring_exchange_begin(ex, rank, send, recv, group=None, self_via_p2p=False):
    """start the ring exchange of one step: send / recv are flat uint8 torch tensors (the rank's send blocks, its local [camera][F + 1] array).  Local runs
    are copied at once (self_via_p2p: sent through the backend like the others — a one-rank RCCL run then exercises the transport); returns the list of
    pending point-to-point requests (wait() on each: ring_exchange_end)."""
    import torch.distributed as dist
    bb = ex.lay.block_bytes
    ops = []
    for (peer, off, n) in ex.recvs(rank):
        if peer != rank or self_via_p2p:
            ops.append(dist.P2POp(dist.irecv, recv[off * bb:(off + n) * bb], peer, group))
    for (peer, off, n) in ex.sends(rank):
        if peer != rank or self_via_p2p:
            ops.append(dist.P2POp(dist.isend, send[off * bb:(off + n) * bb], peer, group))
    if not self_via_p2p:
        for owner, src, dst, n in ex._runs(rank):
            if owner == rank:
                recv[dst * bb:(dst + n) * bb].copy_(send[src * bb:(src + n) * bb])
    return dist.batch_isend_irecv(ops) if ops else []


def ring_exchange_end(reqs):
    for r in reqs:
        r.wait()


def pack_blocks(layout, desc, mask, nkp):
    """numpy restatement of the send buffer (what mcs_extract_batch_strided + mcs_rig_pack_headers produce on the device):
    desc / mask [L][cap][ds] uint8, nkp [L] -> [L][cap + 1][2*ds] uint8"""
    L, cap, ds = desc.shape
    out = np.zeros((L, cap + 1, 2 * ds), np.uint8)
    out[:, :cap, :ds] = desc
    out[:, :cap, ds:] = mask
    out[:, cap, :4] = np.ascontiguousarray(nkp, "<i4").view(np.uint8).reshape(L, 4)
    return out


def unpack_frame(layout, gathered, frame):
    """numpy view of multi-frame `frame` of a gathered [images_total][cap + 1][2*ds] array: (desc [rows_frame][ds], mask, valid [rows_frame])"""
    ds, cap = layout.desc_size, layout.cap
    rows = gathered.reshape(-1, layout.row_stride)
    phys = layout.frame_rows(frame)
    d, m = rows[phys, :ds], rows[phys, ds:]
    counts = np.array([int(gathered[layout.image_index(c, frame), cap, :4].view("<i4")[0]) for c in range(layout.ncam)])
    valid = (np.arange(layout.rows_frame) % cap) < np.repeat(counts, cap)
    return np.ascontiguousarray(d), np.ascontiguousarray(m), valid.astype(np.uint8)


def all_gather_blocks(send, world, group=None):
    """ONE all-gather of the ranks' send buffers (torch uint8 tensors of identical size) -> [world * len(send)] on every rank.
    A CUDA tensor under the gloo backend (functional runs of the N > 1 path on a single GPU) bounces through host memory."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return send
    flat = send.reshape(-1)
    bounce = flat.is_cuda and dist.get_backend(group) == "gloo"
    src = flat.cpu() if bounce else flat
    out = torch.empty(world * src.numel(), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src, group=group)
    return out.to(flat.device) if bounce else out


def reduce_timing(elapsed_s, units, device, world):
    """max-over-ranks time and summed units (bench contract)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    u = torch.tensor([units], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())
