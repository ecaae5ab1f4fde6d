// worker.h -- render worker above the C ABI.  The public half is upstream's interface (include/worker.h:47-65); what is
// behind it is this build's own.
//
// One shared state block under one mutex and one condition variable replaces upstream's three spun flags: requests raise
// `frame_wanted`; the tracing thread sleeps until a frame is wanted, traces it on every GPU of the worker (one host
// thread per GPU, each its partition of the samples), has the library add the buffers onto the first GPU, and publishes
// the frame; the writing thread sleeps until a published frame has not been written yet and turns it into
// <output_dir>/<scene>-output.png.  As upstream: requests that arrive while a frame is being traced are absorbed by it,
// and stop() lets both threads finish what is pending.
#ifndef RT_WORKER_H
#define RT_WORKER_H

#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "scene_loader.h"

namespace rth {

constexpr int MaxFilenameLength = 256;

class Worker {
public:
    Worker(const std::string& scene_file, const std::string& output_dir, int batch_size);
    ~Worker();

    void run();
    void stop();
    void update_camera(rt_float3 delta_pos, rt_float3 delta_rot);
    void force_render();
    void reset_camera();
    RenderConfig& config() { return scene_.config(); }
    void get_output_filename(char filename[MaxFilenameLength]);
    long frames_encoded();

private:
    struct Gpu {                            // one context per GPU, each with the whole scene and a buffer of its own
        rt_context* ctx = nullptr;
        float* accum = nullptr;             // 3 floats per pixel on that GPU
        int device = 0;
    };
    struct View { rt_float3 pos, dir, up; };
    struct Shared {                         // everything the threads hand each other
        std::mutex lock;
        std::condition_variable changed;
        bool quitting = false;
        bool frame_wanted = false;
        long traced = 0, written = 0;       // frames published by the tracing thread / put on disk by the writing thread
        std::vector<float> newest;          // the last published frame
        std::string png_path;               // where the last frame was written ("" before the first)
    };

    SceneLoader scene_;
    std::vector<Gpu> gpus_;
    rt_group* group_ = nullptr;             // only with more than one GPU
    Shared shared_;
    std::thread tracer_, writer_;
    size_t n_pixels_ = 0;
    std::string output_dir_, scene_name_;
    View home_;                             // the camera of the scene file
    uint64_t seed_ = 0;

    void trace_frames();                    // body of the tracing thread
    void write_frames();                    // body of the writing thread
    bool trace_one(const RenderConfig& config);
    bool trace_partition(size_t gpu, const RenderConfig& config, bool report);
    void request_frame();
    void release_devices();
};

}  // namespace rth
#endif
