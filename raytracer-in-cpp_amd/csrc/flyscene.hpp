// flyscene.hpp -- host facade with the reference's scene-controller interface (src/flyscene.hpp:28-200) for the one
// path this library replaces: initialize() -> raytraceScene() -> result.ppm, plus traceRay/lightStrikes/
// createSpherePoint for single-ray use (the reference's debug ray calls traceRay, flyscene.cpp:286).
// Same member names, argument meaning and outputs; Eigen/Tucano types are replaced by a plain Vec3f because
// neither library exists outside the reference tree.  The OpenGL preview, GUI and debug-ray drawing are out of scope.
#pragma once

#include <array>
#include <string>
#include <vector>

#include "rt_mi355x.h"

namespace rtamd {

using Vec3f = std::array<float, 3>;

class Flyscene {
public:
    Flyscene() = default;
    ~Flyscene();
    Flyscene(const Flyscene &) = delete;
    Flyscene &operator=(const Flyscene &) = delete;

    // reference: prompts "Enter 0 if Point Lights or 1 if Area Lights" / "Enter 0 if spherical or 1 if point" on stdin,
    // loads resources/models/cube.obj, normalises it, builds the octree (capacity 1000) -- flyscene.cpp:29-126
    void initialize(int width, int height);
    // same without stdin: the two switches given directly
    void initialize(int width, int height, bool area_light, bool point_light);

    // reference: flyscene.cpp:519-648.  0 => viewport size.  Writes ./result.ppm (ASCII P3) and prints ELAPSED TIME.
    void raytraceScene(int width = 0, int height = 0);

    // reference: flyscene.cpp:651-771 (countRay only drives the progress bar there; accepted and ignored here)
    Vec3f traceRay(Vec3f &origin, Vec3f &direction, int level, std::vector<Vec3f> &lights, bool countRay);
    // reference: flyscene.cpp:912-954
    bool lightStrikes(Vec3f &hitPoint, std::vector<Vec3f> &lights, bool visibleLights[]);
    // reference: flyscene.cpp:962-972 (point and area modes)
    std::vector<Vec3f> createSpherePoint(Vec3f lightPoint);
    // reference: flyscene.hpp:58-62 (adds a light at the camera centre)
    void addLight();

    // knobs the reference hard-codes (flyscene.cpp:51,86,971; boxTree.cpp:3) or does not have
    void setScenePath(const std::string &obj) { scene_path_ = obj; }
    void setAreaGrid(int usteps, int vsteps) { usteps_ = usteps; vsteps_ = vsteps; }
    void setMaxDepth(int d) { max_depth_ = d; }       // <0: the library's maximum (the reference is unbounded)
    void setOutputPath(const std::string &p) { output_path_ = p; }
    void setDevice(int d) { device_ = d; }
    const rt_stats &lastStats() const { return stats_; }
    // status of the last raytraceScene() (the reference's member is void; a headless caller needs to know): RT_OK or a negative rt_status
    rt_status lastStatus() const { return last_status_; }
    const std::vector<float> &lastImage() const { return image_; }
    rt_camera *getCamera() { return &camera_; }
    std::vector<Vec3f> &getLights() { return lights_; }
    bool ok() const { return ctx_ != nullptr; }

private:
    void fill_lights(rt_lights *l, const std::vector<Vec3f> &pts) const;
    std::string scene_path_ = "resources/models/cube.obj";
    std::string output_path_ = "result.ppm";
    rt_ctx *ctx_ = nullptr;
    rt_host_scene *scene_ = nullptr;
    rt_camera camera_{};
    std::vector<Vec3f> lights_;
    bool areaLight = true, pointLight = false;
    int usteps_ = 5, vsteps_ = 5, max_depth_ = -1, device_ = 0;
    int view_w_ = 0, view_h_ = 0;
    rt_stats stats_{};
    rt_status last_status_ = RT_OK;
    std::vector<float> image_;
    std::vector<float> sphere_offsets_;          // spherical light mode: Vector3f(x, y, z) / 5 per sample (rt_sphere_offsets)
    uint32_t sphere_seed_ = 65u;                 // "CG Raytracing Group 65" (README.MD:1)
};

}  // namespace rtamd
