# diagnostic: render one frame with the RT_PROFILE build and print executed-work counters per kernel split
import ctypes as C, os, sys
os.environ["RT_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "raytracer-in-cpp_amd", "lib", "librt_mi355x_prof.so")
import rtpkg
pkg = rtpkg.load()
scene = sys.argv[1] if len(sys.argv) > 1 else "dodgeColorTest.obj"
fs = pkg.Flyscene(scene_path=os.path.join("tests/golden/scenes", scene))
fs.initialize(1920, 1080, True, False)
fs.usteps = fs.vsteps = 8
fs.max_depth = 4
fs.raytraceScene(1920, 1080, write_ppm=False)
st = fs.stats
print("rays", st.total_rays(), "ms", st.ms_trace, st.ms_shadow, st.ms_shade)
