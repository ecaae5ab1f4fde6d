// server.cpp -- bin/server <scene_file> <batch_size> <address> <port>: the interactive front end.
//
// The contract with the browser page is upstream's (csrc/server.cpp:40-107): six routes, their bodies and content
// types, and the frames also reachable as static files.  Here the routes are rows of one table, each bound to a member
// of the Viewer object that owns the worker; nothing else is shared between them.
#include <signal.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <iterator>

#include "http_server.h"
#include "worker.h"

namespace {

bool slurp(const std::string& path, std::string& into) {
    std::ifstream in(path.c_str(), std::ios::binary);
    if (!in.is_open()) return false;
    into.assign(std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>());
    return true;
}

// Six comma-separated numbers, "x,y,z,alpha,beta,gamma": what the page's gyroscope code posts.  Fields that are missing
// or not numbers stay zero.
void six_numbers(const std::string& text, float out[6]) {
    const char* p = text.c_str();
    for (int k = 0; k < 6; k++) {
        out[k] = 0.0f;
        char* end = nullptr;
        const float v = std::strtof(p, &end);
        if (end == p) break;
        out[k] = v;
        p = end;
        while (*p == ' ') p++;
        if (*p != ',') break;
        p++;
    }
}

class Viewer {
public:
    Viewer(const std::string& scene_file, const std::string& static_dir, int batch_size, rth::HttpServer& http)
        : worker_(scene_file, static_dir, batch_size), static_dir_(static_dir), http_(http) {}

    void start() {
        worker_.run();
        worker_.force_render();
        typedef rth::HttpReply (Viewer::*Action)(const rth::HttpRequest&);
        static const struct { const char* method; const char* path; Action action; } rows[] = {
            {"GET", "/", &Viewer::page},
            {"POST", "/update-camera", &Viewer::move_camera},
            {"GET", "/image", &Viewer::newest_frame},
            {"POST", "/reset-camera", &Viewer::home_camera},
            {"GET", "/camera-pos", &Viewer::where},
            {"GET", "/stop", &Viewer::quit},
        };
        for (const auto& row : rows) {
            const Action action = row.action;
            http_.route(row.method, row.path, [this, action](const rth::HttpRequest& request) { return (this->*action)(request); });
        }
        http_.files_from(static_dir_);
    }

private:
    rth::Worker worker_;
    std::string static_dir_;
    rth::HttpServer& http_;

    rth::HttpReply page(const rth::HttpRequest&) {
        std::string html;
        if (!slurp(static_dir_ + "/index.html", html)) return rth::HttpReply::text("File not found", 404);
        std::cout << "Serving index.html" << std::endl;
        return rth::HttpReply::of("text/html", html);
    }

    rth::HttpReply move_camera(const rth::HttpRequest& request) {
        float v[6];
        six_numbers(request.body, v);
        worker_.update_camera(rt_float3{v[0], v[1], v[2]}, rt_float3{v[3], v[4], v[5]});
        const rt_float3 at = worker_.config().camera_pos;
        std::cout << "Camera updated to " << at.x << "," << at.y << "," << at.z << std::endl;
        return rth::HttpReply::text("updated");
    }

    rth::HttpReply newest_frame(const rth::HttpRequest&) {
        char name[rth::MaxFilenameLength];
        worker_.get_output_filename(name);
        std::string png;
        if (!name[0] || !slurp(name, png)) return rth::HttpReply::text("File not found", 404);
        return rth::HttpReply::of("image/png", png);
    }

    rth::HttpReply home_camera(const rth::HttpRequest&) {
        worker_.reset_camera();
        return rth::HttpReply::text("Camera reset");
    }

    rth::HttpReply where(const rth::HttpRequest&) {
        const rt_float3 at = worker_.config().camera_pos;
        char text[128];
        std::snprintf(text, sizeof text, "%f,%f,%f", at.x, at.y, at.z);
        return rth::HttpReply::text(text);
    }

    rth::HttpReply quit(const rth::HttpRequest&) {
        worker_.stop();
        http_.shut_down();
        std::cout << "Server stopped" << std::endl;
        return rth::HttpReply::text("Server stopped");
    }
};

}  // namespace

int main(int argc, char** argv) {
    if (argc != 5) {
        std::cerr << "Usage: " << argv[0] << " <scene_file> <batch_size> <address> <port>" << std::endl;
        return 0;
    }
    // RT_STATIC_DIR moves the directory that holds index.html and receives the frames (upstream: fixed)
    const char* moved = std::getenv("RT_STATIC_DIR");
    const std::string static_dir = moved ? moved : "./assets/static";
    signal(SIGPIPE, SIG_IGN);           // a client that goes away must not end the process
    rth::HttpServer http;
    Viewer viewer(argv[1], static_dir, std::atoi(argv[2]), http);
    viewer.start();
    const std::string address = argv[3];
    const int port = std::atoi(argv[4]);
    std::cout << "Starting server at " << address << ":" << port << std::endl;
    if (!http.serve(address, port)) {
        std::cerr << "Failed to start server" << std::endl;
        return 1;
    }
    return 0;
}
