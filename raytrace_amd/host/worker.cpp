// worker.cpp -- see worker.h.  What upstream's Worker does per frame (csrc/worker.cpp:64-118: clear, per batch a
// primary-ray kernel and a path kernel, post-process, copy to the host, encode) is here a handful of library calls per
// GPU: rt_device_zero, rt_setup_raytrace, rt_raytrace per slice of whole batches, then rt_reduce_to_root when there is
// more than one GPU, rt_post_process with the total sample count, rt_copy_to_host.
#include "worker.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <sstream>

#include "image_io.h"

namespace rth {

namespace {

typedef std::chrono::steady_clock Clock;

double seconds_since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

// "scene_luyu" from "assets/scene/scene_luyu.txt": what lies between the last '/' and the last '.' behind it
std::string stem_of(const std::string& path) {
    const size_t slash = path.rfind('/');
    const size_t begin = slash == std::string::npos ? 0 : slash + 1;
    size_t end = path.rfind('.');
    if (end == std::string::npos || end < begin) end = path.size();
    return path.substr(begin, end - begin);
}

// One line per slice: a 32-cell gauge, the percentage, seconds so far and a linear estimate of the seconds left.
std::string progress_line(int done, int total, double elapsed) {
    const double share = total > 0 ? double(done) / double(total) : 1.0;
    std::string gauge(32, ' ');
    const size_t head = size_t(share * 32.0);
    for (size_t i = 0; i < gauge.size(); i++) gauge[i] = i < head ? '=' : (i == head ? '>' : ' ');
    std::ostringstream line;
    line << '[' << gauge << "] " << int(share * 100.0) << "%\tElapsed(s): " << float(elapsed) << ", ETA(s): "
         << float(share > 0.0 ? elapsed / share - elapsed : 0.0);
    return line.str();
}

// The devices a worker runs on: RT_GPU_LIST = "0,1,2" (an id may repeat: several contexts then share that GPU, which
// is how the multi-GPU path is rehearsed on one GPU), else RT_GPUS consecutive devices starting at RT_DEVICE (default:
// one GPU, device 0; upstream never selects a device).
std::vector<int> devices_from_environment() {
    std::vector<int> list;
    if (const char* text = std::getenv("RT_GPU_LIST")) {
        std::istringstream in(text);
        std::string item;
        while (std::getline(in, item, ','))
            if (!item.empty()) list.push_back(std::atoi(item.c_str()));
    }
    if (list.empty()) {
        const char* first = std::getenv("RT_DEVICE");
        const char* count = std::getenv("RT_GPUS");
        const int base = first ? std::atoi(first) : 0, n = count ? std::max(1, std::atoi(count)) : 1;
        for (int k = 0; k < n; k++) list.push_back(base + k);
    }
    return list;
}

}  // namespace

Worker::Worker(const std::string& scene_file, const std::string& output_dir, int batch_size) : output_dir_(output_dir) {
    if (scene_.load_scene(scene_file) != 0) {
        std::cerr << "Failed to load scene file " << scene_file << std::endl;
        std::exit(0);                                   // upstream leaves with status 0 here (csrc/worker.cpp:123-126)
    }
    RenderConfig& config = scene_.config();
    // a batch is at least one sample and at most the frame (csrc/worker.cpp:128-131)
    config.batch_size = (batch_size <= 0 || batch_size > config.n_samples) ? config.n_samples : batch_size;
    // upstream seeds its generator with time(0) (cuda/camera.cu:20); RT_SEED pins it for reproducible frames
    const char* seed_text = std::getenv("RT_SEED");
    seed_ = seed_text ? std::strtoull(seed_text, nullptr, 0) : uint64_t(std::time(nullptr));
    n_pixels_ = size_t(config.width) * size_t(config.height);
    for (int device : devices_from_environment()) {
        Gpu g;
        g.device = device;
        if (rt_create(device, &g.ctx) != RT_OK || scene_.upload(g.ctx) != RT_OK || rt_init_randstate(g.ctx, seed_, config.width, config.height) != RT_OK) {
            std::cerr << "Failed to set up the device: " << rt_last_error() << std::endl;
            std::exit(0);
        }
        // RT_SHARE_PRIMARY=0: walk every sample's primary ray on its own, as upstream executes it (same picture, slower)
        if (const char* share = std::getenv("RT_SHARE_PRIMARY")) rt_set_option(g.ctx, RT_OPT_SHARE_PRIMARY, std::atoi(share) != 0);
        if (rt_device_alloc(g.ctx, n_pixels_ * 3 * sizeof(float), reinterpret_cast<void**>(&g.accum)) != RT_OK) {
            std::cerr << "Failed to allocate device memory for buffer, error: " << rt_last_error() << std::endl;
            std::exit(0);
        }
        gpus_.push_back(g);
    }
    if (gpus_.size() > 1) {
        std::vector<rt_context*> all;
        std::vector<int> distinct;
        for (const Gpu& g : gpus_) {
            all.push_back(g.ctx);
            bool seen = false;
            for (int d : distinct) seen = seen || d == g.device;
            if (!seen) distinct.push_back(g.device);
        }
        if (rt_group_create(all.data(), int(all.size()), 0, int(distinct.size()), nullptr, &group_) != RT_OK) {
            std::cerr << "Failed to set up the multi-GPU group: " << rt_last_error() << std::endl;
            std::exit(0);
        }
        std::cout << "worker: " << gpus_.size() << " contexts on " << distinct.size() << " GPU(s), samples split between them" << std::endl;
    }
    scene_name_ = stem_of(scene_file);
    std::cout << "worker: loaded scene: " << scene_name_ << std::endl;
    shared_.newest.assign(n_pixels_ * 3, 0.0f);
    home_ = View{config.camera_pos, config.camera_dir, config.camera_up};
}

void Worker::release_devices() {
    if (group_) { rt_group_destroy(group_); group_ = nullptr; }
    for (Gpu& g : gpus_) {
        if (!g.ctx) continue;
        rt_device_free(g.ctx, g.accum);
        rt_destroy(g.ctx);
        g.ctx = nullptr;
    }
}

Worker::~Worker() {
    stop();
    if (tracer_.joinable()) tracer_.join();
    if (writer_.joinable()) writer_.join();
    release_devices();
}

void Worker::run() {
    tracer_ = std::thread(&Worker::trace_frames, this);
    writer_ = std::thread(&Worker::write_frames, this);
}

void Worker::stop() {
    std::lock_guard<std::mutex> hold(shared_.lock);
    shared_.quitting = true;
    shared_.changed.notify_all();
}

void Worker::request_frame() {
    std::lock_guard<std::mutex> hold(shared_.lock);
    shared_.frame_wanted = true;
    shared_.changed.notify_all();
}

void Worker::force_render() { request_frame(); }

void Worker::get_output_filename(char filename[MaxFilenameLength]) {
    std::lock_guard<std::mutex> hold(shared_.lock);
    std::snprintf(filename, MaxFilenameLength, "%s", shared_.png_path.c_str());
}

long Worker::frames_encoded() {
    std::lock_guard<std::mutex> hold(shared_.lock);
    return shared_.written;
}

// One GPU's share of a frame: its partition of the samples (all of them with one GPU), in slices of whole batches.
// Small batches are grouped so that a call carries about 1024 samples per pixel: the library's pipeline drains once
// per call, so long calls cost less.  The GPU that reports prints one progress line per slice.
bool Worker::trace_partition(size_t k, const RenderConfig& config, bool report) {
    const Gpu& g = gpus_[k];
    const int n_pixel = config.width * config.height;
    int first = 0, count = config.n_samples;
    if (rt_partition_samples(config.n_samples, config.batch_size, int(gpus_.size()), int(k), &first, &count) != RT_OK) return false;
    const float ambient[3] = {config.background.x, config.background.y, config.background.z};
    if (rt_device_zero(g.ctx, g.accum, n_pixels_ * 3 * sizeof(float), nullptr) != RT_OK) return false;
    if (rt_setup_raytrace(g.ctx, config.width, config.height, &config.camera_pos.x, &config.camera_dir.x, &config.camera_up.x, config.fov) != RT_OK)
        return false;
    const int slice = std::max(1, 1024 / std::max(1, config.batch_size)) * config.batch_size;
    const Clock::time_point t0 = Clock::now();
    for (int done = 0; done < count;) {
        const int n = std::min(slice, count - done);
        if (rt_raytrace(g.ctx, n_pixel, first + done, n, config.batch_size, config.max_depth, ambient, config.russian_roulette, g.accum, nullptr) != RT_OK ||
            rt_synchronize(g.ctx) != RT_OK)
            return false;
        done += n;
        if (report) std::cout << progress_line(done, count, seconds_since(t0)) << std::endl;
    }
    return true;
}

bool Worker::trace_one(const RenderConfig& config) {
    const Clock::time_point t0 = Clock::now();
    std::cout << "---> Raytracing Start <---" << std::endl;
    const int n_pixel = config.width * config.height;
    bool ok = true;
    if (gpus_.size() == 1) {
        ok = trace_partition(0, config, true);
    } else {
        // one host thread per GPU; the library keeps its error text per thread, so a failing thread reports its own
        std::vector<char> good(gpus_.size(), 1);
        std::vector<std::string> why(gpus_.size());
        std::vector<std::thread> drivers;
        for (size_t k = 1; k < gpus_.size(); k++)
            drivers.emplace_back([this, k, &config, &good, &why] {
                if (!trace_partition(k, config, false)) { good[k] = 0; why[k] = rt_last_error(); }
            });
        if (!trace_partition(0, config, true)) { good[0] = 0; why[0] = rt_last_error(); }
        for (std::thread& t : drivers) t.join();
        for (size_t k = 0; k < gpus_.size(); k++)
            if (!good[k]) { std::cerr << "render failed on GPU " << gpus_[k].device << ": " << why[k] << std::endl; ok = false; }
        if (!ok) return false;
        std::vector<float*> buffers;
        for (const Gpu& g : gpus_) buffers.push_back(g.accum);
        ok = rt_reduce_to_root(group_, buffers.data(), uint64_t(n_pixels_) * 3, nullptr) == RT_OK;
    }
    ok = ok && rt_post_process(gpus_[0].ctx, n_pixel, config.n_samples, config.gamma, gpus_[0].accum, nullptr) == RT_OK &&
         rt_synchronize(gpus_[0].ctx) == RT_OK;
    if (!ok) {
        std::cerr << "render failed: " << rt_last_error() << std::endl;
        return false;
    }
    std::cout << "---> done in " << long(seconds_since(t0) * 1000.0) << "ms <---" << std::endl;
    return true;
}

void Worker::trace_frames() {
    std::vector<float> frame(n_pixels_ * 3);
    for (;;) {
        RenderConfig config;
        {
            std::unique_lock<std::mutex> hold(shared_.lock);
            shared_.changed.wait(hold, [this] { return shared_.frame_wanted || shared_.quitting; });
            if (!shared_.frame_wanted) break;           // quitting with nothing pending; a pending frame is still traced
            config = scene_.config();                   // the camera as it is now; later changes belong to a later request
        }
        const bool ok = trace_one(config) && rt_copy_to_host(gpus_[0].ctx, frame.data(), gpus_[0].accum, n_pixels_ * 3 * sizeof(float)) == RT_OK;
        std::lock_guard<std::mutex> hold(shared_.lock);
        if (ok) {                                       // a frame that failed is not published: the last good picture stays
            shared_.newest.swap(frame);
            shared_.traced++;
        }
        shared_.frame_wanted = false;                   // requests that came in meanwhile are absorbed, as upstream (csrc/worker.cpp:116)
        shared_.changed.notify_all();
    }
    std::cout << "worker: cleaning up" << std::endl;
}

// float frame -> PNG: rows flipped, x255, rounded half to even, saturated, RGB (csrc/worker.cpp:20-42 via OpenCV there)
void Worker::write_frames() {
    std::vector<float> frame;
    std::vector<uint8_t> pixels;
    const std::string path = output_dir_ + "/" + scene_name_ + "-output.png";
    for (;;) {
        int width, height;
        long seen;                                      // the frame this turn encodes: `traced` as it stood when `newest` was copied
        {
            std::unique_lock<std::mutex> hold(shared_.lock);
            shared_.changed.wait(hold, [this] { return shared_.traced > shared_.written || (shared_.quitting && !shared_.frame_wanted); });
            if (shared_.traced == shared_.written) break;
            frame = shared_.newest;
            seen = shared_.traced;
            width = scene_.config().width;
            height = scene_.config().height;
        }
        if (path.size() >= size_t(MaxFilenameLength)) {
            std::cerr << "Filename too long" << std::endl;
            break;
        }
        frame_to_rgb8(frame.data(), width, height, pixels);
        std::string error;
        const bool saved = write_png_rgb8(path.c_str(), width, height, pixels.data(), error);
        if (saved) std::cout << "Image saved to " << path << std::endl;
        else std::cerr << "Cannot save image: " << error << std::endl;
        std::lock_guard<std::mutex> hold(shared_.lock);
        if (saved) shared_.png_path = path;
        // a frame published while this one was being encoded stays unwritten, so the wait above fires again for it
        shared_.written = seen;
    }
    std::cout << "encoder: cleaning up" << std::endl;
}

void Worker::reset_camera() {
    {
        std::lock_guard<std::mutex> hold(shared_.lock);
        RenderConfig& config = scene_.config();
        config.camera_pos = home_.pos;
        config.camera_dir = home_.dir;
        config.camera_up = home_.up;
    }
    request_frame();
}

// The client's position is dropped -- the camera stays where the scene file put it -- and the view direction comes from
// the orientation angles, in degrees: alpha about z, beta about x (csrc/worker.cpp:185-199).
void Worker::update_camera(rt_float3 delta_pos, rt_float3 delta_rot) {
    (void)delta_pos;
    const float to_radians = 0.0174532925f;
    const float alpha = delta_rot.x * to_radians, beta = delta_rot.y * to_radians;
    {
        std::lock_guard<std::mutex> hold(shared_.lock);
        RenderConfig& config = scene_.config();
        config.camera_pos = home_.pos;
        config.camera_dir = rt_float3{std::cos(alpha), std::sin(alpha), -std::cos(beta)};
        config.camera_up = rt_float3{0.0f, 0.0f, 1.0f};
    }
    request_frame();
}

}  // namespace rth
