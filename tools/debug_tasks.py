# prints the leaf-task counts of one frame (RT_DEBUG output of the library): python tools/debug_tasks.py [wavy|file.obj] W H grid depth
import os, sys
os.environ["RT_DEBUG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtpkg
pkg = rtpkg.load()
scene = sys.argv[1] if len(sys.argv) > 1 else "wavy"
w, h, u, d = (int(x) for x in (sys.argv[2:6] if len(sys.argv) > 5 else (3840, 2160, 16, 8)))
if scene == "wavy":
    import scenes_gen
    path = scenes_gen.wavy_grid("/tmp/rt_wavy_dbg", n=708)
else:
    path = os.path.join(ROOT, "tests/golden/scenes", scene)
fs = pkg.Flyscene(scene_path=path)
fs.initialize(w, h, True, False)
fs.usteps = fs.vsteps = u
fs.max_depth = d
for _ in range(3):
    fs.raytraceScene(w, h, write_ppm=False)
print("rays", fs.stats.total_rays(), "items", fs.stats.shaded_hits)
