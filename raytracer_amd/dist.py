"""Multi-GPU frame assembly: screen tiles are dealt round-robin to ranks
(tile t -> rank t % world_size, scene replicated, seeds use the global pixel
index so the partition never changes a pixel's value) and the finished tiles are
gathered to rank 0 with one torch.distributed gather (RCCL over xGMI on GPUs).

No reference counterpart: the reference renders on one device.
The only collective on the path is this gather; rendering itself needs none.
"""
import numpy as np


def tile_origins(width, height, tile_w=64, tile_h=32):
    tx = (width + tile_w - 1) // tile_w
    ty = (height + tile_h - 1) // tile_h
    return [((t % tx) * tile_w, (t // tx) * tile_h) for t in range(tx * ty)]


def tiles_of_rank(width, height, rank, world_size, tile_w=64, tile_h=32):
    """Same rule as rtx_create (raytracer_amd/csrc/rtx_api.hip)."""
    return [o for t, o in enumerate(tile_origins(width, height, tile_w, tile_h)) if t % world_size == rank]


def pack_tiles_host(frame, tiles, tile_w=64, tile_h=32):
    """Host mirror of rtx_pack_tiles: frame (h, w, 4) -> (len(tiles), tile_h, tile_w, 4)."""
    h, w, _ = frame.shape
    out = np.zeros((len(tiles), tile_h, tile_w, 4), dtype=frame.dtype)
    for k, (x0, y0) in enumerate(tiles):
        ww, hh = min(tile_w, w - x0), min(tile_h, h - y0)
        out[k, :hh, :ww] = frame[y0:y0 + hh, x0:x0 + ww]
    return out


def unpack_tiles_host(frame, packed, tiles, tile_w=64, tile_h=32):
    """Host mirror of rtx_unpack_tiles."""
    h, w, _ = frame.shape
    for k, (x0, y0) in enumerate(tiles):
        ww, hh = min(tile_w, w - x0), min(tile_h, h - y0)
        frame[y0:y0 + hh, x0:x0 + ww] = packed[k, :hh, :ww]
    return frame


def max_packed_floats(width, height, world_size, tile_w=64, tile_h=32):
    n = len(tile_origins(width, height, tile_w, tile_h))
    return ((n + world_size - 1) // world_size) * tile_w * tile_h * 4


def gather_frame_host(frame, rank, world_size, tile=(64, 32), group=None):
    """CPU / gloo variant (used by the tests): every rank passes its full-size
    frame array in which only its own tiles are meaningful; rank 0 returns the
    assembled frame, the others None."""
    import torch
    import torch.distributed as dist
    h, w, _ = frame.shape
    n = max_packed_floats(w, h, world_size, *tile)
    mine = tiles_of_rank(w, h, rank, world_size, *tile)
    send = torch.zeros(n, dtype=torch.float32)
    p = pack_tiles_host(frame, mine, *tile).reshape(-1)
    send[:p.size] = torch.from_numpy(np.ascontiguousarray(p))
    bufs = [torch.zeros(n, dtype=torch.float32) for _ in range(world_size)] if rank == 0 else None
    dist.gather(send, bufs, dst=0, group=group)
    if rank != 0:
        return None
    out = np.array(frame, copy=True)
    for src in range(world_size):
        tiles = tiles_of_rank(w, h, src, world_size, *tile)
        k = len(tiles) * tile[0] * tile[1] * 4
        unpack_tiles_host(out, bufs[src][:k].numpy().reshape(len(tiles), tile[1], tile[0], 4), tiles, *tile)
    return out


class DeviceGather:
    """GPU variant: packs this rank's tiles with rtx_pack_tiles into a torch CUDA
    tensor, gathers over the process group (backend "nccl" = RCCL), and on rank 0
    scatters every rank's tiles into the resident framebuffer with
    rtx_unpack_tiles. Buffers are allocated once.

    host_staging=True moves the packed tiles through host memory around the
    collective, for process groups that cannot gather device tensors (gloo): the
    way to rehearse the multi-rank path with several ranks on one GPU."""

    def __init__(self, shader, rank, world_size, device, group=None, host_staging=False):
        import torch
        self.shader, self.rank, self.world, self.group = shader, rank, world_size, group
        self.host_staging = host_staging
        n = max(shader.packed_floats(r) for r in range(world_size))
        # torch.empty: nothing is queued on torch's stream for these buffers, so the renderer's own (non-blocking)
        # stream, which rtx_pack_tiles writes `send` on, cannot be overtaken by a late zero-fill
        self.send = torch.empty(n, dtype=torch.float32, device=device)
        recv_device = "cpu" if host_staging else device
        self.recv = [torch.empty(n, dtype=torch.float32, device=recv_device) for _ in range(world_size)] if rank == 0 else None
        self.stage = torch.empty(n, dtype=torch.float32, device=device) if (host_staging and rank == 0) else None
        self.send_host = torch.empty(n, dtype=torch.float32, pin_memory=True) if host_staging else None
        torch.cuda.synchronize(device)

    def gather(self):
        """One exchange: every rank's tiles to rank 0's resident framebuffer. The device-tensor (nccl = RCCL) path has
        only ever run with one rank on this pool (no multi-GPU box was available to the build); the two-rank tests use
        host staging over gloo."""
        import torch
        import torch.distributed as dist
        self.shader.pack_tiles(self.send.data_ptr())          # synchronises the renderer's stream
        if self.host_staging:
            self.send_host.copy_(self.send)
            torch.cuda.current_stream().synchronize()
            dist.gather(self.send_host, self.recv, dst=0, group=self.group)
        else:
            dist.gather(self.send, self.recv, dst=0, group=self.group)
        if self.rank == 0:
            torch.cuda.current_stream().synchronize()
            for src in range(1, self.world):
                buf = self.recv[src]
                if self.host_staging:
                    self.stage.copy_(buf)
                    torch.cuda.current_stream().synchronize()
                    buf = self.stage
                self.shader.unpack_tiles(buf.data_ptr(), src)
