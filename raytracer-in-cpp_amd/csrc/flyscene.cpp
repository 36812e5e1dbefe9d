// flyscene.cpp -- see flyscene.hpp.  Everything here goes through the C ABI of include/rt_mi355x.h.
#include "flyscene.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

namespace rtamd {

Flyscene::~Flyscene() {
    if (ctx_) rt_destroy(ctx_);
    if (scene_) rt_host_scene_free(scene_);
}

void Flyscene::initialize(int width, int height) {
    int area = 1, point = 0;
    std::cout << "Enter 0 if Point Lights or 1 if Area Lights : " << std::endl;
    std::cin >> area;
    std::cout << "Enter 0 if spherical or 1 if point : " << std::endl;
    std::cin >> point;
    initialize(width, height, area != 0, point != 0);
}

void Flyscene::initialize(int width, int height, bool area_light, bool point_light) {
    areaLight = area_light;
    pointLight = point_light;
    if (!pointLight && !areaLight) {
        // the spherical mode (flyscene.cpp:974-995): 25 points per light, drawn ONCE per initialize() from the seeded restatement of
        // the reference's loop (it re-draws them from an unseeded std::random_device at every shaded hit: not reproducible)
        sphere_offsets_.assign(25 * 3, 0.f);
        rt_sphere_offsets(sphere_seed_, 1.0f, 25, sphere_offsets_.data());
    }
    view_w_ = width; view_h_ = height;
    rt_default_camera(&camera_, width, height);
    lights_.clear();
    lights_.push_back({-1.0f, 1.0f, 1.0f});           // flyscene.cpp:72
    std::cout << "Seting up acceleration data structure ..." << std::endl;
    const auto t0 = std::chrono::high_resolution_clock::now();
    if (scene_) { rt_host_scene_free(scene_); scene_ = nullptr; }
    if (rt_host_scene_load(scene_path_.c_str(), 1000, 15, &scene_) != RT_OK) {
        std::cerr << "Cannot open " << scene_path_ << std::endl;
        std::exit(1);                                  // objimporter.hpp:105
    }
    const std::chrono::duration<double> el = std::chrono::high_resolution_clock::now() - t0;
    std::cout << "Seting up acceleration data structure: done!" << std::endl;
    std::cout << "ELAPSED TIME:" << el.count() << std::endl;
    if (!ctx_) {
        const rt_status s = rt_create(&ctx_, device_);
        if (s != RT_OK) {
            std::cerr << "rt_mi355x: cannot create a device context (status " << s << "); there is no CPU fallback" << std::endl;
            std::exit(1);
        }
    }
    rt_scene view;
    rt_host_scene_view(scene_, &view);
    if (rt_upload_scene(ctx_, &view) != RT_OK) {
        std::cerr << "rt_mi355x: " << rt_last_error(ctx_) << std::endl;
        std::exit(1);
    }
}

void Flyscene::fill_lights(rt_lights *l, const std::vector<Vec3f> &pts) const {
    rt_default_lights(l, (areaLight && !pointLight) ? 1 : 0);
    if (!pointLight && !areaLight) {        // createSpherePoint's third branch
        l->mode = RT_LIGHT_SPHERE;
        l->n_offsets = static_cast<int32_t>(sphere_offsets_.size() / 3);
        l->offsets = sphere_offsets_.data();
    }
    l->n_lights = static_cast<int32_t>(pts.size());
    for (size_t i = 0; i < pts.size() && i < RT_MAX_LIGHTS; ++i) std::memcpy(l->pos[i], pts[i].data(), sizeof(float) * 3);
    l->usteps = usteps_; l->vsteps = vsteps_;
}

void Flyscene::raytraceScene(int width, int height) {
    const auto t0 = std::chrono::high_resolution_clock::now();
    if (width == 0 || height == 0) { width = view_w_; height = view_h_; }
    if (width != view_w_ || height != view_h_) {
        // the reference keeps the viewport's camera; an explicit size re-targets the perspective like initialize(w,h)
        rt_camera keep = camera_;
        rt_default_camera(&camera_, width, height);
        std::memcpy(camera_.center, keep.center, sizeof keep.center);
        std::memcpy(camera_.inv_view, keep.inv_view, sizeof keep.inv_view);
    }
    rt_lights L;
    fill_lights(&L, lights_);
    rt_params p{};
    p.width = width; p.height = height; p.max_depth = max_depth_;
    p.row0 = 0; p.row1 = height; p.stripe = 1; p.rank = 0; p.nranks = 1; p.collect_stats = 0;
    image_.assign(static_cast<size_t>(width) * height * 3, 0.f);
    std::cout << "Ray tracing ..." << std::endl;
    const rt_status s = rt_render(ctx_, &camera_, &L, &p, image_.data(), nullptr, &stats_);
    last_status_ = s;
    if (s != RT_OK) {
        std::cerr << "rt_mi355x: render failed: " << rt_last_error(ctx_) << std::endl;
        return;
    }
    std::cout << "Writting to restult.ppm ... " << std::endl;
    last_status_ = rt_write_ppm(output_path_.c_str(), image_.data(), width, height);
    if (last_status_ != RT_OK) { std::cerr << "rt_mi355x: cannot write " << output_path_ << std::endl; return; }
    const std::chrono::duration<double> el = std::chrono::high_resolution_clock::now() - t0;
    std::cout << "Writting to restult.ppm done!" << std::endl << std::endl << "ray tracing done! " << std::endl;
    std::cout << "ELAPSED TIME:" << el.count() << std::endl;
}

Vec3f Flyscene::traceRay(Vec3f &origin, Vec3f &direction, int level, std::vector<Vec3f> &lights, bool) {
    rt_lights L;
    fill_lights(&L, lights);
    Vec3f out{0.f, 0.f, 0.f};
    // `level` is never tested by the reference; here it offsets the depth budget
    int budget = max_depth_ < 0 ? -1 : (max_depth_ - level < 0 ? 0 : max_depth_ - level);
    if (rt_trace_rays(ctx_, &L, budget, 1, origin.data(), direction.data(), out.data(), nullptr, nullptr) != RT_OK)
        std::cerr << "rt_mi355x: traceRay failed: " << rt_last_error(ctx_) << std::endl;
    return out;
}

bool Flyscene::lightStrikes(Vec3f &hitPoint, std::vector<Vec3f> &lights, bool visibleLights[]) {
    const int n = static_cast<int>(lights.size());
    std::vector<float> hit(static_cast<size_t>(n) * 3), src(static_cast<size_t>(n) * 3);
    std::vector<uint8_t> vis(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) {
        std::memcpy(&hit[i * 3], hitPoint.data(), sizeof(float) * 3);
        std::memcpy(&src[i * 3], lights[i].data(), sizeof(float) * 3);
    }
    bool any = false;
    if (rt_light_strikes(ctx_, n, hit.data(), src.data(), vis.data()) != RT_OK) {
        std::cerr << "rt_mi355x: lightStrikes failed: " << rt_last_error(ctx_) << std::endl;
        return false;
    }
    for (int i = 0; i < n; ++i) { visibleLights[i] = vis[i] != 0; any = any || visibleLights[i]; }
    return any;
}

std::vector<Vec3f> Flyscene::createSpherePoint(Vec3f p) {
    std::vector<Vec3f> out;
    if (pointLight) { out.push_back(p); return out; }
    if (!areaLight) {
        for (size_t i = 0; i + 2 < sphere_offsets_.size(); i += 3) out.push_back({sphere_offsets_[i] + p[0], sphere_offsets_[i + 1] + p[1], sphere_offsets_[i + 2] + p[2]});
        return out;
    }
    // createAreaLight(lightPoint, 0.3, 0.15, usteps, vsteps).getPointLights()  (flyscene.cpp:956-971, arealight.hpp:15-25)
    const float lx = static_cast<float>(0.3), ly = static_cast<float>(0.15);
    const float ux = p[0] + lx * 1.0f, uz = p[2] + lx * 0.0f, vy = p[1] + ly * 1.0f;
    for (int i = 0; i < usteps_; ++i)
        for (int j = 0; j < vsteps_; ++j)
            out.push_back({static_cast<float>(i + 0.5) * (ux / static_cast<float>(usteps_)),
                           static_cast<float>(j + 0.5) * (vy / static_cast<float>(vsteps_)), uz});
    return out;
}

void Flyscene::addLight() {
    if (lights_.size() < RT_MAX_LIGHTS) lights_.push_back({camera_.center[0], camera_.center[1], camera_.center[2]});
}

}  // namespace rtamd
