"""One-off randomised stress of the look-ahead and of cutting dropped sets short (not a test):
random sequences of synchronous / queued renders, present / download deliveries, jumps of the frame counter, camera changes,
uploads, clears, probes and counter reads, at a size where the sets started ahead are still running when they are dropped.
Every sequence is run three ways — never looking ahead, looking ahead with dropped sets running out (RTX_NO_ABORT=1), looking
ahead with dropped sets cut short (the default) — as a single renderer and as a device group of two members on one GPU; all
images, id maps and counters must agree bit for bit.
    python scripts/fuzz_ahead.py [sequences=6] [ops=120] [first_seed=0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd.renderer import HIPShader
from raytracer_amd.scene import orbit, synthetic_scene

nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 6
nops = int(sys.argv[2]) if len(sys.argv) > 2 else 120
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
bad = 0
t0 = time.time()
for seed in range(first, first + nseq):
    rng = np.random.default_rng(31000 + seed)
    w, h = int(rng.choice([640, 1280, 1920])), int(rng.choice([352, 704, 1088]))
    s = synthetic_scene(int(rng.integers(20_000, 300_000)), w, h, seed=100 + seed, tex_size=64, nthreads=8)
    d = s.desc()
    cams = [s.camera]
    for k in range(2):
        c = type(s.camera).from_buffer_copy(s.camera); orbit(c, float(rng.uniform(0, 360)), float(rng.uniform(-30, 30)), radius=float(rng.uniform(1500, 4000))); cams.append(c)
    # the script: mostly the loop (same camera, frame counter + 1, one delivery per frame), left in every way the API allows
    script, cam, fc = [], 0, 0
    for _ in range(nops):
        u = rng.random()
        if u < 0.62: script.append((str(rng.choice(["r", "r", "r", "p", "d"])), cam, fc)); fc += 1
        elif u < 0.70: fc = int(rng.integers(0, 900)); script.append(("r", cam, fc)); fc += 1          # a jump
        elif u < 0.76: cam = int(rng.integers(0, 3)); script.append(("r", cam, fc)); fc += 1            # another camera
        elif u < 0.84:
            for _ in range(int(rng.integers(1, 12))): script.append(("q", cam, fc)); fc += 1            # queued frames in between
        elif u < 0.88: script.append(("upload", 0, int(rng.integers(0, 1 << 30))))
        elif u < 0.91: script.append(("clear", 0, 0)); fc = 0
        elif u < 0.95: script.append(("ids", int(rng.integers(0, 3)), int(rng.integers(0, 50))))
        else: script.append(("counters", 0, 0))
    def run(look, no_abort, group):
        print(f"  seed {seed} run look={look} no_abort={int(no_abort)} group={int(group)} ...", flush=True)
        if no_abort: os.environ["RTX_NO_ABORT"] = "1"
        else: os.environ.pop("RTX_NO_ABORT", None)
        r = HIPShader(s, devices=[0, 0], look_ahead=look) if group else HIPShader(s, look_ahead=look)
        px = np.zeros((d.height, d.width, 4), np.float32)
        import zlib
        log = []
        keep = lambda a: log.append(zlib.crc32(np.ascontiguousarray(a).view(np.uint8).reshape(-1)))
        for op, a, b in script:
            if op in ("r", "q", "p", "d"):
                r.frame_counter = b
                r.Render(px if op == "r" else None, cams[a])
                if op == "r": keep(px)
                if op == "p": keep(r.present(2.2))
                if op == "d": keep(r.download())
            elif op == "upload": r.upload(np.random.default_rng(b).random((d.height, d.width, 4), dtype=np.float32))
            elif op == "clear": r.clear()
            elif op == "ids": keep(r.primary_hits(b, cams[a]))
            elif op == "counters":
                c = r.counters(); log.append(tuple([c.frames] + list(c.raysPerBounce) + list(c.hitsPerBounce)))
        c = r.counters(); log.append(tuple([c.frames] + list(c.raysPerBounce) + list(c.hitsPerBounce)))
        cut = r.info().setsCutShort
        r.close()
        return log, cut
    for group in (False, True):
        base, _ = run(-1, False, group)
        out, cut0 = run(0, True, group)
        cutlog, cut = run(0, False, group)
        ok = base == out == cutlog and cut0 == 0
        bad += not ok
        print(f"seed {seed} {w}x{h} ops {len(script)} group {int(group)}: sets cut short {cut}, identical {ok}" + ("" if ok else "   <-- MISMATCH"), flush=True)
os.environ.pop("RTX_NO_ABORT", None)
print(f"{2 * nseq} cases, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
