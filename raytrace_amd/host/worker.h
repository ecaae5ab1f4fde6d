// worker.h -- render worker above the C ABI, with upstream's interface (include/worker.h:21-66).
//
// Two threads and three flags, as upstream: the render thread waits for has_new_input, traces a frame into the
// device buffer, copies it to the host and raises has_new_output; the encode thread turns the newest host
// buffer into <output_dir>/<scene>-output.png.  stop() lets both finish what is pending.
#ifndef RT_WORKER_H
#define RT_WORKER_H

#include <atomic>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "scene_loader.h"

namespace rth {

constexpr int MaxFilenameLength = 256;

class Worker {
    SceneLoader loader;
    rt_context* ctx_;
    std::atomic_bool should_exit;
    std::atomic_bool has_new_input;
    std::atomic_bool has_new_output;
    char output_filename[MaxFilenameLength];
    std::thread render_thread, encode_thread;
    float* d_buffer_;                       // 3 floats per pixel on the device
    std::vector<float> h_buffer_;           // newest finished frame
    size_t n_pixels_;
    std::mutex filename_mutex, buffer_mutex, config_mutex;
    std::string output_dir;
    std::string scene_name;
    rt_float3 initial_cam_pos, initial_cam_dir, initial_cam_up;
    uint64_t seed_;
    std::atomic<long> frames_rendered_, frames_encoded_;

    void post_process_loop_();
    void render_();
    void run_loop_();

public:
    Worker(const std::string& scene_file, const std::string& output_dir, int batch_size);
    ~Worker();

    void run();
    void stop() { should_exit.store(true); }
    void update_camera(rt_float3 delta_pos, rt_float3 delta_rot);
    void force_render() { has_new_input.store(true); }
    void reset_camera();
    RenderConfig& config() { return loader.config(); }

    void get_output_filename(char filename[MaxFilenameLength]) {
        std::lock_guard<std::mutex> lock(filename_mutex);
        std::strcpy(filename, output_filename);
    }
    long frames_encoded() const { return frames_encoded_.load(); }
};

}  // namespace rth
#endif
