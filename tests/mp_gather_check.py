"""Run by tests/test_gpu_multirank.py under torch.distributed.run with 2 ranks sharing one GPU (gloo, tiles staged
through host memory): every rank renders its own tiles, the tiles are gathered to rank 0, and the assembled
frame must equal the frame one rank renders alone — bit for bit."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from raytracer_amd.dist import DeviceGather
    from raytracer_amd.renderer import HIPShader
    from raytracer_amd.scene import HostScene
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    scene = HostScene.load_blob(os.path.join(ROOT, "tests", "golden", "scene2.blob"))
    shader = HIPShader(scene, device=0, rank=rank, world_size=world)
    for fc in range(3):
        shader.frame_counter = fc
        shader.Render(None)
    DeviceGather(shader, rank, world, torch.device("cuda", 0), host_staging=True).gather()
    ok = 1
    if rank == 0:
        got = shader.download()
        alone = HIPShader(scene, device=0)
        for fc in range(3):
            alone.frame_counter = fc
            alone.Render(None)
        want = alone.download()
        ok = int((got == want).all())
        print("assembled frame equals the single-rank frame:", bool(ok), flush=True)
    flag = torch.tensor([ok])
    dist.broadcast(flag, src=0)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
