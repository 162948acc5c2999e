// Tiny blocking HTTP/1.1 server (thread per connection, keep-alive) and client for the native
// control-plane binaries.  Plain TCP only: in-cluster API access goes through a
// `kubectl proxy` sidecar or the controller's own HTTP port.
#pragma once
#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <thread>

namespace vhttp {

struct Request {
  std::string method, path, query, body;
  std::map<std::string, std::string> headers;  // lower-case names
};

struct Response {
  int status = 200;
  std::string content_type = "application/json";
  std::string body;
};

inline std::string lower(std::string s) {
  for (auto& c : s) c = (char)tolower((unsigned char)c);
  return s;
}

inline bool read_message(int fd, std::string& buf, std::string& head, std::string& body) {
  size_t hdr_end;
  while ((hdr_end = buf.find("\r\n\r\n")) == std::string::npos) {
    char tmp[8192];
    ssize_t n = recv(fd, tmp, sizeof tmp, 0);
    if (n <= 0) return false;
    buf.append(tmp, (size_t)n);
    if (buf.size() > (64u << 20)) return false;
  }
  head = buf.substr(0, hdr_end);
  size_t clen = 0;
  {
    std::string lh = lower(head);
    size_t p = lh.find("content-length:");
    if (p != std::string::npos) clen = (size_t)strtoull(lh.c_str() + p + 15, nullptr, 10);
  }
  size_t need = hdr_end + 4 + clen;
  while (buf.size() < need) {
    char tmp[65536];
    ssize_t n = recv(fd, tmp, sizeof tmp, 0);
    if (n <= 0) return false;
    buf.append(tmp, (size_t)n);
  }
  body = buf.substr(hdr_end + 4, clen);
  buf.erase(0, need);
  return true;
}

inline void parse_head(const std::string& head, Request& r) {
  size_t eol = head.find("\r\n");
  std::string line = head.substr(0, eol);
  size_t a = line.find(' '), b = line.rfind(' ');
  r.method = line.substr(0, a);
  std::string target = line.substr(a + 1, b - a - 1);
  size_t q = target.find('?');
  r.path = target.substr(0, q);
  r.query = q == std::string::npos ? "" : target.substr(q + 1);
  size_t pos = eol == std::string::npos ? head.size() : eol + 2;
  while (pos < head.size()) {
    size_t e = head.find("\r\n", pos);
    if (e == std::string::npos) e = head.size();
    std::string h = head.substr(pos, e - pos);
    size_t c = h.find(':');
    if (c != std::string::npos) {
      std::string v = h.substr(c + 1);
      while (!v.empty() && v[0] == ' ') v.erase(0, 1);
      r.headers[lower(h.substr(0, c))] = v;
    }
    pos = e + 2;
  }
}

inline std::string query_param(const std::string& query, const std::string& key) {
  size_t pos = 0;
  while (pos < query.size()) {
    size_t amp = query.find('&', pos);
    std::string kv = query.substr(pos, amp == std::string::npos ? std::string::npos : amp - pos);
    size_t eq = kv.find('=');
    if (kv.substr(0, eq) == key) return eq == std::string::npos ? "" : kv.substr(eq + 1);
    if (amp == std::string::npos) break;
    pos = amp + 1;
  }
  return "";
}

using Handler = std::function<Response(const Request&)>;

class Server {
 public:
  // returns the bound port (0 on failure); serves on background threads
  int start(const std::string& host, int port, Handler h) {
    handler_ = std::move(h);
    fd_ = socket(AF_INET, SOCK_STREAM, 0);
    int one = 1;
    setsockopt(fd_, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    sockaddr_in addr{};
    addr.sin_family = AF_INET;
    addr.sin_port = htons((uint16_t)port);
    addr.sin_addr.s_addr = host.empty() || host == "0.0.0.0" ? INADDR_ANY : inet_addr(host.c_str());
    if (bind(fd_, (sockaddr*)&addr, sizeof addr) != 0 || listen(fd_, 1024) != 0) return 0;
    socklen_t len = sizeof addr;
    getsockname(fd_, (sockaddr*)&addr, &len);
    running_ = true;
    acceptor_ = std::thread([this] { accept_loop(); });
    return ntohs(addr.sin_port);
  }
  void stop() {
    running_ = false;
    if (fd_ >= 0) {
      shutdown(fd_, SHUT_RDWR);
      close(fd_);
      fd_ = -1;
    }
    if (acceptor_.joinable()) acceptor_.join();
  }
  void wait() {
    if (acceptor_.joinable()) acceptor_.join();
  }

 private:
  void accept_loop() {
    while (running_) {
      int c = accept(fd_, nullptr, nullptr);
      if (c < 0) {
        if (!running_) break;
        continue;
      }
      int one = 1;
      setsockopt(c, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
      std::thread([this, c] { serve(c); }).detach();
    }
  }
  void serve(int c) {
    std::string buf, head, body;
    while (read_message(c, buf, head, body)) {
      Request r;
      parse_head(head, r);
      r.body = std::move(body);
      Response resp;
      try {
        resp = handler_(r);
      } catch (const std::exception& e) {
        resp.status = 500;
        resp.body = std::string("{\"error\":\"") + e.what() + "\"}";
      }
      std::string out = "HTTP/1.1 " + std::to_string(resp.status) + " X\r\nContent-Type: " +
                        resp.content_type + "\r\nContent-Length: " + std::to_string(resp.body.size()) +
                        "\r\nConnection: keep-alive\r\n\r\n" + resp.body;
      size_t off = 0;
      while (off < out.size()) {
        ssize_t n = send(c, out.data() + off, out.size() - off, MSG_NOSIGNAL);
        if (n <= 0) {
          close(c);
          return;
        }
        off += (size_t)n;
      }
    }
    close(c);
  }
  Handler handler_;
  int fd_ = -1;
  std::atomic<bool> running_{false};
  std::thread acceptor_;
};

// url = http://host:port[/prefix]; returns status (0 on connection failure)
inline int request(const std::string& method, const std::string& url, const std::string& path,
                   const std::string& body, std::string* out, const std::string& content_type = "application/json",
                   const std::string& bearer = "") {
  std::string rest = url.substr(url.find("://") == std::string::npos ? 0 : url.find("://") + 3);
  std::string prefix;
  size_t slash = rest.find('/');
  if (slash != std::string::npos) {
    prefix = rest.substr(slash);
    rest = rest.substr(0, slash);
  }
  while (!prefix.empty() && prefix.back() == '/') prefix.pop_back();
  std::string host = rest, port = "80";
  size_t colon = rest.rfind(':');
  if (colon != std::string::npos) {
    host = rest.substr(0, colon);
    port = rest.substr(colon + 1);
  }
  addrinfo hints{}, *res = nullptr;
  hints.ai_family = AF_UNSPEC;
  hints.ai_socktype = SOCK_STREAM;
  if (getaddrinfo(host.c_str(), port.c_str(), &hints, &res) != 0 || !res) return 0;
  int fd = socket(res->ai_family, res->ai_socktype, res->ai_protocol);
  timeval tv{10, 0};
  setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
  setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
  if (fd < 0 || connect(fd, res->ai_addr, res->ai_addrlen) != 0) {
    if (fd >= 0) close(fd);
    freeaddrinfo(res);
    return 0;
  }
  freeaddrinfo(res);
  std::string req = method + " " + prefix + path + " HTTP/1.1\r\nHost: " + host + "\r\nConnection: close\r\n";
  if (!bearer.empty()) req += "Authorization: Bearer " + bearer + "\r\n";
  if (!body.empty() || method == "POST" || method == "PUT" || method == "PATCH")
    req += "Content-Type: " + content_type + "\r\nContent-Length: " + std::to_string(body.size()) + "\r\n";
  req += "\r\n" + body;
  size_t off = 0;
  while (off < req.size()) {
    ssize_t n = send(fd, req.data() + off, req.size() - off, MSG_NOSIGNAL);
    if (n <= 0) {
      close(fd);
      return 0;
    }
    off += (size_t)n;
  }
  std::string resp;
  char tmp[65536];
  ssize_t n;
  while ((n = recv(fd, tmp, sizeof tmp, 0)) > 0) resp.append(tmp, (size_t)n);
  close(fd);
  size_t he = resp.find("\r\n\r\n");
  if (he == std::string::npos || resp.size() < 12) return 0;
  int status = atoi(resp.c_str() + 9);
  std::string head = lower(resp.substr(0, he));
  std::string payload = resp.substr(he + 4);
  if (head.find("transfer-encoding: chunked") != std::string::npos) {
    std::string dec;
    size_t p = 0;
    while (p < payload.size()) {
      size_t e = payload.find("\r\n", p);
      if (e == std::string::npos) break;
      size_t len = strtoul(payload.substr(p, e - p).c_str(), nullptr, 16);
      if (len == 0) break;
      dec.append(payload, e + 2, len);
      p = e + 2 + len + 2;
    }
    payload.swap(dec);
  }
  if (out) *out = std::move(payload);
  return status;
}

}  // namespace vhttp
