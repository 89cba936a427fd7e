"""One-off randomised parity sweep over primitive and material space (not a test): random spheres, cubes, planes,
disks and small meshes with random materials (transparent, metallic, emissive, textured), random cameras, BVH on or off,
against the oracle.    python scripts/fuzz_prims.py [count=60] [first_seed=0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from raytracer_amd.renderer import HIPShader
from raytracer_amd.scene import HostScene, orbit, procedural_texture

count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(9000 + seed)
    w, h = int(rng.integers(2, 7)) * 32, int(rng.integers(2, 6)) * 32
    s = HostScene(w, h)
    texs = [s.add_texture(procedural_texture(int(rng.choice([1, 2, 16, 64])), 10 + i)) for i in range(int(rng.integers(1, 4)))]
    mats = []
    for i in range(int(rng.integers(2, 9))):
        f = lambda: float(rng.random())
        mats.append(s.add_material(
            f"M{i}", albedo=(f(), f(), f(), 1.0), tint=(f(), f(), f(), 1.0), specular=(f(), f(), f(), 1.0),
            transmissionFilter=(f(), f(), f(), 1.0), specularIntensity=f(),
            transparency=float(rng.choice([0.0, 0.0, f(), 1.0])), indexOfRefraction=1.0 + 1.5 * f(),
            roughness=float(rng.choice([0.0, f(), 1.0])), metallic=float(rng.choice([0.0, 1.0, f()])), sheen=f(),
            tintRoughness=f(), clearcoatThickness=float(rng.choice([0.0, f()])), clearcoatRoughness=f(),
            emmissionIntensity=float(rng.choice([0.0, 0.0, 0.0, f()])), anisotropy=f(), anisotropyRotation=f(),
            textureID=int(rng.choice([0] + texs))))
    pick = lambda: int(rng.choice(mats + [0]))
    pos = lambda: tuple(float(v) for v in rng.uniform([0, 100, 0], [1000, 900, 900]))
    if rng.random() < 0.8:
        s.add_plane((500, 100, 500), (0, 1, 0), (3000, 3000, 3000), pick())
    if rng.random() < 0.8:
        s.add_disk((500, float(rng.uniform(1200, 3000)), 500), (0, -1, 0), float(rng.uniform(200, 900)), pick())
    for _ in range(int(rng.integers(0, 25))):
        s.add_sphere(pos(), float(rng.uniform(5, 200)), pick())
    for _ in range(int(rng.integers(0, 12))):
        s.add_cube(pos(), tuple(float(v) for v in rng.uniform(10, 250, 3)), pick())
    if len(s.objects) == 0:
        s.add_sphere(pos(), 100.0, pick())
    use_bvh = bool(rng.random() < 0.8)
    if use_bvh:
        s.build_bvh(4)
    d, cam = s.desc(), s.camera
    orbit(cam, float(rng.uniform(0, 360)), float(rng.uniform(-40, 60)), radius=float(rng.uniform(600, 4000)))
    frames = [0, 1, 2, 3, 4, 0, 1][: int(rng.integers(1, 8))]
    want = np.zeros((d.height, d.width, 4), np.float32)
    for fc in frames:
        oracle.render(d, cam, fc, want, 8)
    r = HIPShader(s)
    for fc in frames:
        r.frame_counter = fc
        r.Render(None, cam)
    got = r.download()
    ok_ids = bool((r.primary_hits(frames[-1], cam) == oracle.primary_hits(d, cam, frames[-1])).all())
    r.close()
    nan = np.isnan(want)
    same_nan = bool((np.isnan(got) == nan).all())
    a, b = np.where(nan, 0.0, got), np.where(nan, 0.0, want)
    fin = np.isfinite(a) & np.isfinite(b)
    err = float(np.abs(np.where(fin, a - b, 0.0)).max())
    same_inf = bool((np.where(fin, 0.0, a) == np.where(fin, 0.0, b)).all())
    exact = bool((a.view(np.uint32) == b.view(np.uint32)).all())
    flag = "" if (same_nan and same_inf and err <= 2e-3 and ok_ids and exact) else "   <-- MISMATCH"
    bad += bool(flag)
    print(f"seed {seed:3d} objs {len(s.objects):4d} {w}x{h} bvh {use_bvh} frames {len(frames)} nan px {int(nan.any(axis=2).sum()):5d} max|d| {err:.2e} bit-identical {exact} ids {ok_ids}{flag}", flush=True)
print(f"{count} cases, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
