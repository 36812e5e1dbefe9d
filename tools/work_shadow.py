# executed-work counters of the SHADOW kernels (counting build): python tools/work_shadow.py [dodge|wavy|cube] W H grid depth
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, rtpkg
pkg = rtpkg.load()
scene = sys.argv[1] if len(sys.argv) > 1 else "dodge"
W, H, G, D = (int(x) for x in (sys.argv[2:6] if len(sys.argv) > 5 else (1920, 1080, 8, 4)))
name, path = bench.scene_of(scene)
hs = pkg.HostScene(path, 1000, 15)
w = bench.work_counters(pkg, hs, W, H, G, D)
for k, v in w["shadow"].items():
    print(f"  {k:36s} {v:14d}")
