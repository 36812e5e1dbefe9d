# Assembles profiles/traffic.json and profiles/valu.json from the outputs of tools/traffic.sh and tools/valu.sh
# (gpurun_out/traffic_<scene>/traffic_<scene>.json, gpurun_out/valu_<scene>/valu_<scene>.json).
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the PMC constants are only quoted by bench.py while they belong to the kernels source that is being timed
SHA = hashlib.sha256(open(os.path.join(ROOT, "raytracer-in-cpp_amd", "csrc", "rt_kernels.hip"), "rb").read()).hexdigest()
cfg = {"cube": "1920x1080 depth 4 64 samples", "dodge": "1920x1080 depth 4 64 samples", "wavy": "3840x2160 depth 8 256 samples"}
traffic_path, valu_path = os.path.join(ROOT, "profiles", "traffic.json"), os.path.join(ROOT, "profiles", "valu.json")
traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
valu = json.load(open(valu_path)) if os.path.exists(valu_path) else {}


def shadow_kernel(d, cont):
    if not cont:
        for k in d:
            if "k_shadow_shaft" in k:
                return k
    for k in d:
        if "k_shadow<false" in k and k.rstrip(">").endswith("true" if cont else "false"):
            return k
    return None


for sc in cfg:
    p = os.path.join(ROOT, "gpurun_out", f"traffic_{sc}", f"traffic_{sc}.json")
    if os.path.exists(p):
        d = json.load(open(p))
        k = shadow_kernel(d, False)
        f, w = d[k]["FETCH_SIZE_KiB_max_launch"], d[k]["WRITE_SIZE_KiB_max_launch"]
        # every kernel of the frame (heaviest launch): FETCH_SIZE x 2 + WRITE_SIZE; the counting-pass variants are left out
        per_kernel = {}
        for kk, vv in d.items():
            args = [a.strip() for a in kk[kk.index("<") + 1:kk.rindex(">")].split(",")] if "<" in kk else []
            count_arg = {"k_shadow": 0, "k_trace": 1, "k_stage": 1}.get(kk.split("::")[-1].split("<")[0])
            if count_arg is not None and args and args[count_arg] == "true":
                continue
            if kk.split("::")[-1].startswith("k_shade<") and args[1] == "false" and any(q.split("::")[-1].startswith("k_shade<") and q.rstrip(">").endswith("true") for q in d):
                continue
            per_kernel[kk] = {"fetch_KiB": vv["FETCH_SIZE_KiB_max_launch"], "write_KiB": vv["WRITE_SIZE_KiB_max_launch"],
                              "hbm_bytes": int((2 * vv["FETCH_SIZE_KiB_max_launch"] + vv["WRITE_SIZE_KiB_max_launch"]) * 1024)}
        traffic[sc] = {
            "kernels": per_kernel,
            "kernel": k, "fetch_KiB_per_launch": f, "write_KiB_per_launch": w, "hbm_bytes_per_launch": int((2 * f + w) * 1024),
            "how": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --scene {sc} --steps 2 --warmup 1` "
                   "(tools/traffic.sh); level-0 launch; FETCH_SIZE doubled per the gfx950 rule of MI355X_MICROARCH.md §HBM (calibrated there "
                   "for 16-B/lane streams; our mix of 64-B scalar and 16-B lane loads is uncalibrated)",
            "config": cfg[sc], "kernels_sha256": SHA}
    p = os.path.join(ROOT, "gpurun_out", f"valu_{sc}", f"valu_{sc}.json")
    if os.path.exists(p):
        d = json.load(open(p))
        ent = {"config": cfg[sc], "kernels_sha256": SHA, "how": f"rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE over `bench.py --scene {sc} --steps 2 --warmup 1` "
                                          "(tools/valu.sh); heaviest (level-0) launch of each kernel; gpu_cycles = GRBM_GUI_ACTIVE / 8 XCDs",
               "kernels": {}}
        for k, v in d.items():
            args = [a.strip() for a in k[k.index("<") + 1:k.rindex(">")].split(",")] if "<" in k else []
            count_arg = {"k_shadow": 0, "k_trace": 1, "k_stage": 1}.get(k.split("::")[-1].split("<")[0])
            if count_arg is not None and args and args[count_arg] == "true":
                continue                                    # counting-pass variants
            if k.split("::")[-1].startswith("k_shade<") and args[1] == "false" and any(kk.split("::")[-1].startswith("k_shade<") and kk.rstrip(">").endswith("true") for kk in d):
                continue                                    # flat scenes: k_shade<., false> is the counting pass's launch (the timed frames run k_shade<., true>)
            ent["kernels"][k] = v
        valu[sc] = ent
json.dump(traffic, open(traffic_path, "w"), indent=1)
json.dump(valu, open(valu_path, "w"), indent=1)
print("traffic:", {k: v["hbm_bytes_per_launch"] for k, v in traffic.items()})
print("valu:", {k: {kk: int(vv["valu_wave_instructions"]) for kk, vv in v["kernels"].items() if "k_shadow" in kk} for k, v in valu.items()})
