// server.cpp -- bin/server <scene_file> <batch_size> <address> <port>: the interactive front end
// (csrc/server.cpp:17-116).  Same six routes, same bodies, same static directory.
#include <signal.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <iterator>
#include <sstream>

#include "http_server.h"
#include "worker.h"

int main(int argc, char** argv) {
    if (argc != 5) {
        std::cerr << "Usage: " << argv[0] << " <scene_file> <batch_size> <address> <port>" << std::endl;
        return 0;
    }
    const std::string scene_file = argv[1];
    const int batch_size = std::atoi(argv[2]);
    const std::string address = argv[3];
    const int port = std::atoi(argv[4]);
    // RT_STATIC_DIR moves the directory that holds index.html and receives the frames (upstream: fixed)
    const char* dir_env = std::getenv("RT_STATIC_DIR");
    const std::string output_dir = dir_env ? dir_env : "./assets/static";
    const std::string index_file = output_dir + "/index.html";
    rth::Worker worker(scene_file, output_dir, batch_size);
    worker.run();
    signal(SIGPIPE, SIG_IGN);           // a client that goes away must not end the process
    worker.force_render();

    rth::HttpServer svr;
    svr.Get("/", [index_file](const rth::HttpRequest&, rth::HttpResponse& res) {
        std::ifstream file(index_file.c_str());
        if (!file.is_open()) {
            res.status = 404;
            res.set_content("File not found", "text/plain");
            return;
        }
        const std::string content((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());
        res.set_content(content, "text/html");
        std::cout << "Serving index.html" << std::endl;
    });
    svr.set_base_dir(output_dir);

    // body "x,y,z,alpha,beta,gamma": position (ignored by the worker) and orientation angles in degrees
    svr.Post("/update-camera", [&](const rth::HttpRequest& req, rth::HttpResponse& res) {
        rt_float3 delta_pos = {0.0f, 0.0f, 0.0f}, delta_dir = {0.0f, 0.0f, 0.0f};
        std::istringstream body(req.body);
        char comma;
        body >> delta_pos.x >> comma >> delta_pos.y >> comma >> delta_pos.z >> comma;
        body >> delta_dir.x >> comma >> delta_dir.y >> comma >> delta_dir.z;
        worker.update_camera(delta_pos, delta_dir);
        res.set_content("updated", "text/plain");
        const rth::RenderConfig& config = worker.config();
        std::cout << "Camera updated to " << config.camera_pos.x << "," << config.camera_pos.y << "," << config.camera_pos.z << std::endl;
    });

    svr.Get("/image", [&](const rth::HttpRequest&, rth::HttpResponse& res) {
        char filename[rth::MaxFilenameLength];
        worker.get_output_filename(filename);
        std::ifstream file(filename, std::ios::binary);
        if (!file.is_open()) {
            res.status = 404;
            res.set_content("File not found", "text/plain");
            return;
        }
        const std::string content((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());
        res.set_content(content, "image/png");
    });

    svr.Post("/reset-camera", [&](const rth::HttpRequest&, rth::HttpResponse& res) {
        worker.reset_camera();
        res.set_content("Camera reset", "text/plain");
    });

    svr.Get("/camera-pos", [&](const rth::HttpRequest&, rth::HttpResponse& res) {
        const rth::RenderConfig& config = worker.config();
        char buffer[128];
        std::snprintf(buffer, sizeof buffer, "%f,%f,%f", config.camera_pos.x, config.camera_pos.y, config.camera_pos.z);
        res.set_content(buffer, "text/plain");
    });

    svr.Get("/stop", [&](const rth::HttpRequest&, rth::HttpResponse& res) {
        worker.stop();
        svr.stop();
        std::cout << "Server stopped" << std::endl;
        res.set_content("Server stopped", "text/plain");
    });

    std::cout << "Starting server at " << address << ":" << port << std::endl;
    if (!svr.listen(address, port)) {
        std::cerr << "Failed to start server" << std::endl;
        return 1;
    }
    return 0;
}
