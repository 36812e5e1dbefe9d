# diagnostic: render one frame with the RT_PROFILE build and print executed-work counters
import ctypes as C, os, sys
os.environ["RT_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracer-in-cpp_amd", "lib", "librt_mi355x_prof.so")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtpkg
pkg = rtpkg.load()
scene = sys.argv[1] if len(sys.argv) > 1 else "dodgeColorTest.obj"
w, h, u, d = (int(x) for x in (sys.argv[2:6] if len(sys.argv) > 5 else (1920, 1080, 8, 4)))
if scene == "wavy":
    import scenes_gen
    path = scenes_gen.wavy_grid("/tmp/rt_wavy_prof", n=708)
else:
    path = os.path.join(ROOT, "tests/golden/scenes", scene)
fs = pkg.Flyscene(scene_path=path)
fs.initialize(w, h, True, False)
fs.usteps = fs.vsteps = u
fs.max_depth = d
for rep in range(3):          # the last frame is the warm one: read ITS lines
    sys.stderr.write("RT_PROFILE ---- frame %d\n" % rep)
    sys.stderr.flush()
    fs.raytraceScene(w, h, write_ppm=False)
st = fs.stats
print("rays", st.total_rays(), "items", st.shaded_hits, "ms", st.ms_trace, st.ms_shadow, st.ms_shade)
