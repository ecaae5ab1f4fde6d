// http_server.h -- the small part of HTTP/1.1 the render server needs, over POSIX sockets.
//
// Upstream uses cpp-httplib (third/httplib.h) for six routes and one static directory (csrc/server.cpp:36-113).
// This server handles one request per connection ("Connection: close"), GET and POST with Content-Length, a table of
// exact (method, path) routes, and GET of plain files below one directory.  Connections are answered concurrently (a
// thread per connection, as cpp-httplib's pool does): routes must be safe to call from several threads at once.
#ifndef RT_HTTP_SERVER_H
#define RT_HTTP_SERVER_H

#include <atomic>
#include <functional>
#include <string>
#include <vector>

namespace rth {

struct HttpRequest {
    std::string method, path, body;
};

struct HttpReply {
    int status;
    std::string type, body;
    static HttpReply text(const std::string& b, int status = 200) { return HttpReply{status, "text/plain", b}; }
    static HttpReply of(const std::string& type, const std::string& b) { return HttpReply{200, type, b}; }
};

// A route answers a request; returning is all it does (a route that wants the server to end calls shut_down()).
typedef std::function<HttpReply(const HttpRequest&)> HttpRoute;

class HttpServer {
public:
    void route(const std::string& method, const std::string& path, HttpRoute action);
    void files_from(const std::string& directory) { directory_ = directory; }   // GET of anything else looks here
    bool serve(const std::string& address, int port);    // blocks until shut_down(); false if the socket cannot be set up
    void shut_down() { stopping_.store(true); }           // may be called from a route

private:
    struct Entry { std::string method, path; HttpRoute action; };
    std::vector<Entry> table_;
    std::string directory_;
    std::atomic_bool stopping_{false};
    std::atomic_int conversations_{0};                    // connections being answered right now
    static const int kMaxConversations = 32;

    HttpReply answer(const HttpRequest& request) const;
    void converse(int fd) const;
};

}  // namespace rth
#endif
