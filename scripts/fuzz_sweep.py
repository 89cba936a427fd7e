"""One-off randomised parity sweep (not a test): N random scenes / cameras / rank splits against the oracle.
   python scripts/fuzz_sweep.py [count=40] [first_seed=0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from raytracer_amd.renderer import HIPShader
from raytracer_amd.scene import orbit, synthetic_scene

count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(5000 + seed)
    ntri = int(rng.integers(200, 200000))
    w, h = int(rng.integers(2, 9)) * 32, int(rng.integers(2, 7)) * 32
    s = synthetic_scene(ntri, w, h, seed=7000 + seed, tex_size=32, nthreads=4)
    d, cam = s.desc(), s.camera
    orbit(cam, float(rng.uniform(0, 360)), float(rng.uniform(-40, 40)), radius=float(rng.uniform(900, 5000)))
    world = int(rng.choice([1, 1, 2, 3, 8]))
    want = np.zeros((d.height, d.width, 4), np.float32)
    frames = [0, 1, 2, 3, 4, 5, 6, 7, 8, 0, 1][: int(rng.integers(2, 12))]
    for fc in frames:
        oracle.render(d, cam, fc, want, 8)
    ids_want = oracle.primary_hits(d, cam, frames[-1])
    got = np.zeros_like(want)
    ok_ids = True
    for rank in range(world):
        r = HIPShader(s, rank=rank, world_size=world, tile=(32, 16) if seed % 2 else (64, 32))
        for fc in frames:
            r.frame_counter = fc
            r.Render(None, cam)
        part = np.full_like(want, np.nan)
        r.frame_counter = frames[-1]
        # deliver this rank's tiles: download() returns the resident mean (zeros outside the rank's tiles)
        img = r.download()
        mask = np.zeros((d.height, d.width), bool)
        tw, th = (32, 16) if seed % 2 else (64, 32)
        tx = (d.width + tw - 1) // tw
        t = 0
        for y0 in range(0, d.height, th):
            for x0 in range(0, d.width, tw):
                if t % world == rank:
                    mask[y0:y0 + th, x0:x0 + tw] = True
                t += 1
        got[mask] = img[mask]
        if world == 1:
            ok_ids = bool((r.primary_hits(frames[-1], cam) == ids_want).all())
        r.close()
    nan = np.isnan(want)
    same_nan = bool((np.isnan(got) == nan).all())
    a, b = np.where(nan, 0.0, got), np.where(nan, 0.0, want)
    err = float(np.abs(a - b).max())
    exact = bool((a == b).all())
    flag = "" if (same_nan and err <= 2e-3 and ok_ids and exact) else "   <-- MISMATCH"
    bad += bool(flag)
    print(f"seed {seed:3d} tris {ntri:6d} {w}x{h} world {world} frames {len(frames):2d} max|d| {err:.2e} bit-identical {exact} ids {ok_ids}{flag}", flush=True)
print(f"{count} cases, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
