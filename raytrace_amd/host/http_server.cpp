// http_server.cpp -- see http_server.h.
#include "http_server.h"

#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <thread>

namespace rth {

namespace {

const char* reason(int status) {
    switch (status) {
    case 200: return "OK";
    case 400: return "Bad Request";
    case 404: return "Not Found";
    case 405: return "Method Not Allowed";
    case 413: return "Payload Too Large";
    default: return "Error";
    }
}

const char* mime_of(const std::string& path) {
    const size_t dot = path.find_last_of('.');
    const std::string ext = dot == std::string::npos ? "" : path.substr(dot + 1);
    if (ext == "html" || ext == "htm") return "text/html";
    if (ext == "png") return "image/png";
    if (ext == "jpg" || ext == "jpeg") return "image/jpeg";
    if (ext == "js") return "application/javascript";
    if (ext == "css") return "text/css";
    if (ext == "txt") return "text/plain";
    return "application/octet-stream";
}

bool send_all(int fd, const char* p, size_t n) {
    while (n) {
        const ssize_t k = ::send(fd, p, n, MSG_NOSIGNAL);
        if (k <= 0) return false;
        p += k; n -= size_t(k);
    }
    return true;
}

void send_reply(int fd, const HttpReply& r) {
    char head[256];
    const int n = std::snprintf(head, sizeof head, "HTTP/1.1 %d %s\r\nContent-Type: %s\r\nContent-Length: %zu\r\nConnection: close\r\n\r\n",
                                r.status, reason(r.status), r.type.c_str(), r.body.size());
    if (send_all(fd, head, size_t(n))) send_all(fd, r.body.data(), r.body.size());
}

}  // namespace

void HttpServer::route(const std::string& method, const std::string& path, HttpRoute action) {
    table_.push_back(Entry{method, path, action});
}

// The table first; then, for GET, a plain file below the directory (no ".." anywhere in the path).
HttpReply HttpServer::answer(const HttpRequest& request) const {
    for (const Entry& e : table_)
        if (e.method == request.method && e.path == request.path) return e.action(request);
    if (request.method == "GET" && !directory_.empty() && request.path.size() > 1 && request.path.find("..") == std::string::npos) {
        const std::string file = directory_ + request.path;
        struct stat st;
        std::ifstream in(file.c_str(), std::ios::binary);
        if (::stat(file.c_str(), &st) == 0 && S_ISREG(st.st_mode) && in)
            return HttpReply::of(mime_of(file), std::string(std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>()));
    }
    return HttpReply::text("Not Found", 404);
}

// Reads one request from the connection and answers it.
void HttpServer::converse(int fd) const {
    timeval tv = {5, 0};
    setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
    std::string buf;
    size_t head_end = std::string::npos;
    char chunk[4096];
    while (head_end == std::string::npos && buf.size() < 65536) {
        const ssize_t k = ::recv(fd, chunk, sizeof chunk, 0);
        if (k <= 0) return;
        buf.append(chunk, size_t(k));
        head_end = buf.find("\r\n\r\n");
    }
    if (head_end == std::string::npos) { send_reply(fd, HttpReply::text("Bad Request", 400)); return; }
    HttpRequest req;
    {
        const size_t sp1 = buf.find(' '), sp2 = sp1 == std::string::npos ? sp1 : buf.find(' ', sp1 + 1);
        if (sp2 == std::string::npos || sp2 > buf.find("\r\n")) { send_reply(fd, HttpReply::text("Bad Request", 400)); return; }
        req.method = buf.substr(0, sp1);
        req.path = buf.substr(sp1 + 1, sp2 - sp1 - 1);
        const size_t q = req.path.find('?');
        if (q != std::string::npos) req.path.resize(q);
    }
    size_t content_length = 0;
    {
        // header names are case-insensitive
        std::string lower = buf.substr(0, head_end);
        for (char& c : lower) c = char(std::tolower(static_cast<unsigned char>(c)));
        const size_t p = lower.find("\r\ncontent-length:");
        if (p != std::string::npos) content_length = size_t(std::strtoul(lower.c_str() + p + 17, nullptr, 10));
    }
    if (content_length > (1u << 20)) { send_reply(fd, HttpReply::text("Payload Too Large", 413)); return; }
    req.body = buf.substr(head_end + 4);
    while (req.body.size() < content_length) {
        const ssize_t k = ::recv(fd, chunk, sizeof chunk, 0);
        if (k <= 0) break;
        req.body.append(chunk, size_t(k));
    }
    req.body.resize(std::min(req.body.size(), content_length));
    send_reply(fd, answer(req));
}

bool HttpServer::serve(const std::string& address, int port) {
    addrinfo hints;
    std::memset(&hints, 0, sizeof hints);
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_STREAM;
    hints.ai_flags = AI_PASSIVE;
    addrinfo* info = nullptr;
    char service[16];
    std::snprintf(service, sizeof service, "%d", port);
    if (getaddrinfo(address.empty() ? nullptr : address.c_str(), service, &hints, &info) != 0 || !info) return false;
    const int fd = ::socket(info->ai_family, info->ai_socktype, info->ai_protocol);
    if (fd < 0) { freeaddrinfo(info); return false; }
    int one = 1;
    setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    const bool ok = ::bind(fd, info->ai_addr, info->ai_addrlen) == 0 && ::listen(fd, 16) == 0;
    freeaddrinfo(info);
    if (!ok) { ::close(fd); return false; }
    while (!stopping_.load()) {
        // wake up regularly so that shut_down() is noticed even when nobody connects
        fd_set set;
        FD_ZERO(&set);
        FD_SET(fd, &set);
        timeval tv = {0, 200000};
        const int ready = ::select(fd + 1, &set, nullptr, nullptr, &tv);
        if (ready <= 0) continue;
        const int client = ::accept(fd, nullptr, nullptr);
        if (client < 0) continue;
        // Every connection gets a short-lived thread of its own, so a client that connects and then sends nothing (it is
        // given five seconds) holds up nobody else's request.  Beyond kMaxConversations at once the accept loop answers
        // the connection itself, which bounds the threads a flood of connections can start.
        auto talk = [this, client] {
            converse(client);
            ::shutdown(client, SHUT_RDWR);
            ::close(client);
        };
        if (conversations_.fetch_add(1) < kMaxConversations) {
            std::thread([this, talk] { talk(); conversations_.fetch_sub(1); }).detach();
        } else {
            talk();
            conversations_.fetch_sub(1);
        }
    }
    ::close(fd);
    // the detached threads use this object: wait for the last of them (each ends within its receive timeout)
    while (conversations_.load() > 0) ::usleep(10000);
    return true;
}

}  // namespace rth
