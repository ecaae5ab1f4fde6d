// worker.cpp -- see worker.h.  Follows csrc/worker.cpp (cited per block); the per-batch kernel pair of upstream's
// render_ (:71-91) is one rt_raytrace call per slice of whole batches.
#include "worker.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <iostream>

#include "image_io.h"

namespace rth {

namespace {

void print_progress_bar(int current, int total, std::chrono::duration<float> elapsed) {   // csrc/worker.cpp:49-62
    const int bar_width = 32;
    const float progress = float(current) / float(total);
    const int pos = int(bar_width * progress);
    std::cout << "[";
    for (int i = 0; i < bar_width; i++) std::cout << (i < pos ? "=" : (i == pos ? ">" : " "));
    std::cout << "] " << int(progress * 100.0) << "%\t";
    std::cout << "Elapsed(s): " << elapsed.count() << ", ";
    std::cout << "ETA(s): " << (elapsed / progress - elapsed).count() << "" << std::endl;
}

void nap() { std::this_thread::sleep_for(std::chrono::milliseconds(1)); }

}  // namespace

// csrc/worker.cpp:120-160
Worker::Worker(const std::string& scene_file, const std::string& output_dir_, int batch_size)
    : ctx_(nullptr), should_exit(false), has_new_input(false), has_new_output(false), d_buffer_(nullptr), n_pixels_(0),
      output_dir(output_dir_), seed_(0), frames_rendered_(0), frames_encoded_(0) {
    output_filename[0] = '\0';
    if (loader.load_scene(scene_file) != 0) {
        std::cerr << "Failed to load scene file " << scene_file << std::endl;
        std::exit(0);
    }
    RenderConfig& config = loader.config();
    if (batch_size <= 0 || batch_size > config.n_samples) batch_size = config.n_samples;
    config.batch_size = batch_size;
    const char* dev = std::getenv("RT_DEVICE");
    if (rt_create(dev ? std::atoi(dev) : 0, &ctx_) != RT_OK || loader.upload(ctx_) != RT_OK) {
        std::cerr << "Failed to set up the device: " << rt_last_error() << std::endl;
        std::exit(0);
    }
    // upstream seeds its generator with time(0) (cuda/camera.cu:20); RT_SEED pins it for reproducible frames
    const char* seed_env = std::getenv("RT_SEED");
    seed_ = seed_env ? std::strtoull(seed_env, nullptr, 0) : uint64_t(std::time(nullptr));
    rt_init_randstate(ctx_, seed_, config.width, config.height);
    n_pixels_ = size_t(config.width) * size_t(config.height);
    if (rt_device_alloc(ctx_, n_pixels_ * 3 * sizeof(float), reinterpret_cast<void**>(&d_buffer_)) != RT_OK) {
        std::cerr << "Failed to allocate device memory for buffer, error: " << rt_last_error() << std::endl;
        std::exit(0);
    }
    // scene name = file name without directory and extension: prefix of the output file
    size_t pos = scene_file.find_last_of('/');
    pos = (pos == std::string::npos) ? 0 : pos + 1;
    scene_name = scene_file.substr(pos, scene_file.find_last_of('.') - pos);
    std::cout << "worker: loaded scene: " << scene_name << std::endl;
    h_buffer_.assign(n_pixels_ * 3, 0.0f);
    initial_cam_pos = config.camera_pos;
    initial_cam_dir = config.camera_dir;
    initial_cam_up = config.camera_up;
}

Worker::~Worker() {                                       // csrc/worker.cpp:162-170
    should_exit.store(true);
    if (render_thread.joinable()) render_thread.join();
    if (encode_thread.joinable()) encode_thread.join();
    if (ctx_) {
        rt_device_free(ctx_, d_buffer_);
        rt_destroy(ctx_);
    }
}

void Worker::run() {                                      // csrc/worker.cpp:172-175
    render_thread = std::thread(&Worker::run_loop_, this);
    encode_thread = std::thread(&Worker::post_process_loop_, this);
}

// csrc/worker.cpp:64-97
void Worker::render_() {
    RenderConfig config;
    {
        std::lock_guard<std::mutex> lock(config_mutex);
        config = loader.config();
    }
    const auto start = std::chrono::high_resolution_clock::now();
    std::cout << "---> Raytracing Start <---" << std::endl;
    const int n_pixel = config.width * config.height;
    rt_device_zero(ctx_, d_buffer_, n_pixels_ * 3 * sizeof(float), nullptr);
    rt_setup_raytrace(ctx_, config.width, config.height, &config.camera_pos.x, &config.camera_dir.x, &config.camera_up.x, config.fov);
    const float ambient[3] = {config.background.x, config.background.y, config.background.z};
    // whole batches per call; small batches are grouped so that a call carries about 1024 samples per pixel (the
    // library's pipeline drains once per call, so long calls cost less; progress is printed per call)
    const int batches_per_slice = std::max(1, 1024 / std::max(1, config.batch_size));
    const int slice = batches_per_slice * config.batch_size;
    int done = 0;
    while (done < config.n_samples) {
        const int n = std::min(slice, config.n_samples - done);
        if (rt_raytrace(ctx_, n_pixel, done, n, config.batch_size, config.max_depth, ambient, config.russian_roulette, d_buffer_, nullptr) != RT_OK ||
            rt_synchronize(ctx_) != RT_OK) {
            std::cerr << "render failed: " << rt_last_error() << std::endl;
            return;
        }
        done += n;
        print_progress_bar(done, config.n_samples, std::chrono::high_resolution_clock::now() - start);
    }
    rt_post_process(ctx_, n_pixel, config.n_samples, config.gamma, d_buffer_, nullptr);
    rt_synchronize(ctx_);
    const auto end = std::chrono::high_resolution_clock::now();
    std::cout << "---> done in " << std::chrono::duration_cast<std::chrono::milliseconds>(end - start).count() << "ms <---" << std::endl;
}

// csrc/worker.cpp:99-118
void Worker::run_loop_() {
    std::vector<float> frame(n_pixels_ * 3);
    while (!should_exit.load() || has_new_input.load()) {
        if (!has_new_input.load()) { nap(); continue; }
        render_();
        rt_copy_to_host(ctx_, frame.data(), d_buffer_, n_pixels_ * 3 * sizeof(float));
        {
            std::lock_guard<std::mutex> lock(buffer_mutex);
            std::swap(h_buffer_, frame);
        }
        frames_rendered_.fetch_add(1);
        // this order keeps the encode thread alive until the frame is out: it leaves only when all three flags are down
        has_new_output.store(true);
        has_new_input.store(false);                      // requests that arrived while this frame rendered are absorbed, as upstream
    }
    std::cout << "worker: cleaning up" << std::endl;
}

// csrc/worker.cpp:14-47
void Worker::post_process_loop_() {
    std::vector<float> frame;
    std::vector<uint8_t> pixels;
    while (!should_exit.load() || has_new_input.load() || has_new_output.load()) {
        if (!has_new_output.load()) { nap(); continue; }
        const int width = loader.config().width, height = loader.config().height;
        {
            std::lock_guard<std::mutex> lock(buffer_mutex);
            frame = h_buffer_;
        }
        frame_to_rgb8(frame.data(), width, height, pixels);     // flip, x255, round, saturate
        {
            std::lock_guard<std::mutex> lock(filename_mutex);
            const int n = std::snprintf(output_filename, MaxFilenameLength, "%s/%s-%s.png", output_dir.c_str(), scene_name.c_str(), "output");
            if (n >= MaxFilenameLength) {
                std::cerr << "Filename too long" << std::endl;
                return;
            }
            std::string error;
            if (write_png_rgb8(output_filename, width, height, pixels.data(), error)) std::cout << "Image saved to " << output_filename << std::endl;
            else std::cerr << "Cannot save image: " << error << std::endl;
        }
        frames_encoded_.fetch_add(1);
        has_new_output.store(false);
    }
    std::cout << "encoder: cleaning up" << std::endl;
}

void Worker::reset_camera() {                             // csrc/worker.cpp:177-183
    {
        std::lock_guard<std::mutex> lock(config_mutex);
        RenderConfig& config = loader.config();
        config.camera_pos = initial_cam_pos;
        config.camera_dir = initial_cam_dir;
        config.camera_up = initial_cam_up;
    }
    has_new_input.store(true);
}

// csrc/worker.cpp:185-199: the position sent by the client is discarded (the camera stays where the scene file put
// it); the view direction comes from the device's orientation angles alpha (about z) and beta (about x), in degrees.
void Worker::update_camera(rt_float3 pos, rt_float3 rot) {
    (void)pos;
    {
        std::lock_guard<std::mutex> lock(config_mutex);
        RenderConfig& config = loader.config();
        config.camera_pos = initial_cam_pos;
        const float alpha = rot.x * 0.0174532925f, beta = rot.y * 0.0174532925f;
        config.camera_dir = {std::cos(alpha), std::sin(alpha), -std::cos(beta)};
        config.camera_up = {0.0f, 0.0f, 1.0f};
    }
    has_new_input.store(true);
}

}  // namespace rth
