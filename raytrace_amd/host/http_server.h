// http_server.h -- the small part of HTTP/1.1 the render server needs, over POSIX sockets.
//
// Upstream uses cpp-httplib (third/httplib.h) for six routes and one static directory (csrc/server.cpp:36-113).
// This server handles one request per connection ("Connection: close"), GET and POST with Content-Length, exact-path
// handlers, and GET of plain files below a base directory.
#ifndef RT_HTTP_SERVER_H
#define RT_HTTP_SERVER_H

#include <atomic>
#include <functional>
#include <map>
#include <string>

namespace rth {

struct HttpRequest {
    std::string method, path, body;
};

struct HttpResponse {
    int status = 200;
    std::string content_type = "text/plain";
    std::string body;
    void set_content(const std::string& b, const std::string& type) { body = b; content_type = type; }
};

class HttpServer {
    typedef std::function<void(const HttpRequest&, HttpResponse&)> Handler;
    std::map<std::string, Handler> handlers_;      // key = "METHOD path"
    std::string base_dir_;
    std::atomic<int> listen_fd_;
    std::atomic_bool stopping_;

    void serve(int fd);

public:
    HttpServer() : listen_fd_(-1), stopping_(false) {}
    void Get(const std::string& path, Handler h) { handlers_["GET " + path] = h; }
    void Post(const std::string& path, Handler h) { handlers_["POST " + path] = h; }
    void set_base_dir(const std::string& dir) { base_dir_ = dir; }
    bool listen(const std::string& address, int port);   // blocks until stop(); false if the socket cannot be set up
    void stop();                                          // may be called from a handler
};

}  // namespace rth
#endif
