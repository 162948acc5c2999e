// Tiny blocking HTTP/1.1 server (thread per connection, keep-alive) and client for the native
// control-plane binaries.  Plain TCP always; TLS on both sides when the translation unit is built
// with -DVHTTP_TLS and linked against OpenSSL (the operator: https:// API server with the
// service-account CA bundle and bearer token, a TLS metrics endpoint whose certificate files are
// re-read when they change).  Without the flag an https:// URL fails to connect, nothing else
// changes (the picker and the self-tests are built without it).
#pragma once
#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <sys/stat.h>

#include <atomic>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>

#ifdef VHTTP_TLS
#include <openssl/err.h>
#include <openssl/ssl.h>
#include <openssl/x509v3.h>
#endif

namespace vhttp {

// One connection: a socket, and the TLS session on top of it when there is one.
struct Conn {
  int fd = -1;
#ifdef VHTTP_TLS
  SSL* ssl = nullptr;
#endif
};

inline ssize_t conn_recv(Conn& c, void* buf, size_t n) {
#ifdef VHTTP_TLS
  if (c.ssl) {
    int r = SSL_read(c.ssl, buf, (int)n);
    return r > 0 ? r : (r == 0 ? 0 : -1);
  }
#endif
  return recv(c.fd, buf, n, 0);
}

inline ssize_t conn_send(Conn& c, const void* buf, size_t n) {
#ifdef VHTTP_TLS
  if (c.ssl) {
    int r = SSL_write(c.ssl, buf, (int)n);
    return r > 0 ? r : -1;
  }
#endif
  return send(c.fd, buf, n, MSG_NOSIGNAL);
}

inline void conn_close(Conn& c) {
#ifdef VHTTP_TLS
  if (c.ssl) {
    SSL_shutdown(c.ssl);
    SSL_free(c.ssl);
    c.ssl = nullptr;
  }
#endif
  if (c.fd >= 0) close(c.fd);
  c.fd = -1;
}

// What an https:// client connection trusts.  ca_file empty = the system's default trust store;
// insecure_skip_verify only for tests against a throw-away certificate.
struct TlsClientConfig {
  std::string ca_file;
  bool insecure_skip_verify = false;
};
inline TlsClientConfig& tls_client_config() {
  static TlsClientConfig cfg;
  return cfg;
}

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

// Limits of what a peer may make this process buffer.
constexpr size_t kMaxHeaderBytes = 1u << 20;
constexpr size_t kMaxBodyBytes = 64u << 20;

// Value of a header in a lower-cased head block ("" when absent).  Only a name at the start of a
// line counts: "x-content-length: 9" or a value that mentions the name must not match.
inline std::string header_value(const std::string& lower_head, const std::string& name) {
  size_t pos = 0;
  const std::string needle = "\r\n" + name + ":";
  size_t p = lower_head.find(needle, pos);
  if (p == std::string::npos) return "";
  size_t b = p + needle.size(), e = lower_head.find("\r\n", b);
  std::string v = lower_head.substr(b, e == std::string::npos ? std::string::npos : e - b);
  while (!v.empty() && (v[0] == ' ' || v[0] == '\t')) v.erase(0, 1);
  while (!v.empty() && (v.back() == ' ' || v.back() == '\t')) v.pop_back();
  return v;
}

// Content-Length of a head block: -1 when absent, -2 when present but not a plain decimal number
// or larger than the body limit.
inline long long content_length(const std::string& lower_head) {
  std::string v = header_value(lower_head, "content-length");
  if (v.empty()) return lower_head.find("\r\ncontent-length:") == std::string::npos ? -1 : -2;
  if (v.size() > 12) return -2;
  for (char c : v)
    if (c < '0' || c > '9') return -2;
  long long n = atoll(v.c_str());
  return n > (long long)kMaxBodyBytes ? -2 : n;
}

// Reads one request (head + Content-Length body) from fd.  false = connection closed, a read
// error, or a message this server refuses to buffer (oversized head or body, bad length).
inline bool read_message(Conn& fd, std::string& buf, std::string& head, std::string& body) {
  size_t hdr_end;
  while ((hdr_end = buf.find("\r\n\r\n")) == std::string::npos) {
    if (buf.size() > kMaxHeaderBytes) return false;
    char tmp[8192];
    ssize_t n = conn_recv(fd, tmp, sizeof tmp);
    if (n <= 0) return false;
    buf.append(tmp, (size_t)n);
  }
  if (hdr_end > kMaxHeaderBytes) return false;
  head = buf.substr(0, hdr_end);
  long long cl = content_length(lower(head));
  if (cl == -2) return false;
  size_t clen = cl < 0 ? 0 : (size_t)cl;
  size_t need = hdr_end + 4 + clen;
  while (buf.size() < need) {
    char tmp[65536];
    ssize_t n = conn_recv(fd, tmp, sizeof tmp);
    if (n <= 0) return false;
    buf.append(tmp, (size_t)n);
  }
  body = buf.substr(hdr_end + 4, clen);
  buf.erase(0, need);
  return true;
}

inline bool read_message(int fd, std::string& buf, std::string& head, std::string& body) {
  Conn c;
  c.fd = fd;
  return read_message(c, buf, head, body);
}

// false when the request line is not "METHOD SP target SP HTTP/x.y".
inline bool parse_head(const std::string& head, Request& r) {
  size_t eol = head.find("\r\n");
  std::string line = head.substr(0, eol);
  size_t a = line.find(' '), b = line.rfind(' ');
  if (a == std::string::npos || b == a || a == 0 || line.compare(b + 1, 5, "HTTP/") != 0) return false;
  r.method = line.substr(0, a);
  for (char c : r.method)
    if (c < 'A' || c > 'Z') return false;
  std::string target = line.substr(a + 1, b - a - 1);
  if (target.empty() || target[0] != '/') return false;
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
  return true;
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
  // TLS: serve with this certificate chain and key (PEM files).  The files are looked at again
  // (modification time) whenever a connection arrives, and re-read when they have changed: a
  // rotated certificate is picked up without a restart.  Call before start(); false when the
  // files do not load (or the build has no TLS).
  bool use_tls(const std::string& cert_file, const std::string& key_file) {
#ifdef VHTTP_TLS
    std::lock_guard<std::mutex> g(tls_mu_);
    cert_file_ = cert_file;
    key_file_ = key_file;
    cert_mtime_ = mtime_of(cert_file_);
    key_mtime_ = mtime_of(key_file_);
    return reload_tls();
#else
    (void)cert_file;
    (void)key_file;
    return false;
#endif
  }
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
#ifdef VHTTP_TLS
  static long long mtime_of(const std::string& f) {
    struct stat st{};
    return stat(f.c_str(), &st) == 0 ? (long long)st.st_mtime * 1000000000LL + st.st_mtim.tv_nsec : -1;
  }
  bool reload_tls() {
    SSL_CTX* ctx = SSL_CTX_new(TLS_server_method());
    if (!ctx) return false;
    SSL_CTX_set_min_proto_version(ctx, TLS1_2_VERSION);
    if (SSL_CTX_use_certificate_chain_file(ctx, cert_file_.c_str()) != 1 ||
        SSL_CTX_use_PrivateKey_file(ctx, key_file_.c_str(), SSL_FILETYPE_PEM) != 1 ||
        SSL_CTX_check_private_key(ctx) != 1) {
      SSL_CTX_free(ctx);
      return false;
    }
    if (tls_ctx_) SSL_CTX_free(tls_ctx_);  // sessions in flight hold their own reference
    tls_ctx_ = ctx;
    tls_on_ = true;
    return true;
  }
  SSL* accept_tls(int c) {
    SSL* ssl;
    {
      std::lock_guard<std::mutex> g(tls_mu_);
      long long cm = mtime_of(cert_file_), km = mtime_of(key_file_);
      if (cm != cert_mtime_ || km != key_mtime_) {
        // a half-written pair fails to load and the old certificate stays; the new times are
        // remembered either way, so a broken file is not parsed again for every connection
        reload_tls();
        cert_mtime_ = cm;
        key_mtime_ = km;
      }
      ssl = SSL_new(tls_ctx_);
    }
    if (!ssl) return nullptr;
    timeval tv{10, 0};  // a peer that never finishes the handshake must not hold a thread for ever
    setsockopt(c, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
    SSL_set_fd(ssl, c);
    if (SSL_accept(ssl) != 1) {
      SSL_free(ssl);
      return nullptr;
    }
    timeval none{0, 0};
    setsockopt(c, SOL_SOCKET, SO_RCVTIMEO, &none, sizeof none);
    return ssl;
  }
#endif
  void serve(int cfd) {
    Conn c;
    c.fd = cfd;
#ifdef VHTTP_TLS
    if (tls_on_) {
      c.ssl = accept_tls(cfd);
      if (!c.ssl) {
        close(cfd);
        return;
      }
    }
#endif
    std::string buf, head, body;
    while (read_message(c, buf, head, body)) {
      Request r;
      Response resp;
      if (!parse_head(head, r)) {
        resp.status = 400;
        resp.body = "{\"error\":\"malformed request line\"}";
      } else {
        r.body = std::move(body);
        try {
          resp = handler_(r);
        } catch (const std::exception& e) {
          resp.status = 500;
          resp.body = "{\"error\":\"internal error\"}";
          (void)e;
        }
      }
      std::string out = "HTTP/1.1 " + std::to_string(resp.status) + " X\r\nContent-Type: " +
                        resp.content_type + "\r\nContent-Length: " + std::to_string(resp.body.size()) +
                        "\r\nConnection: keep-alive\r\n\r\n" + resp.body;
      size_t off = 0;
      while (off < out.size()) {
        ssize_t n = conn_send(c, out.data() + off, out.size() - off);
        if (n <= 0) {
          conn_close(c);
          return;
        }
        off += (size_t)n;
      }
    }
    conn_close(c);
  }
  Handler handler_;
  int fd_ = -1;
  std::atomic<bool> running_{false};
  std::thread acceptor_;
#ifdef VHTTP_TLS
  std::string cert_file_, key_file_;
  SSL_CTX* tls_ctx_ = nullptr;  // guarded by tls_mu_, like the two times below
  std::atomic<bool> tls_on_{false};
  std::mutex tls_mu_;
  long long cert_mtime_ = -1, key_mtime_ = -1;
#endif
};

// Splits http[s]://host:port[/prefix] and connects (https: TLS handshake, certificate verified
// against tls_client_config() and the host name or address); fd < 0 on failure.
inline Conn connect_url(const std::string& url, std::string* host_out, std::string* prefix_out, int timeout_s) {
  Conn out;
  const bool https = url.compare(0, 8, "https://") == 0;
  std::string rest = url.substr(url.find("://") == std::string::npos ? 0 : url.find("://") + 3);
  std::string prefix;
  size_t slash = rest.find('/');
  if (slash != std::string::npos) {
    prefix = rest.substr(slash);
    rest = rest.substr(0, slash);
  }
  while (!prefix.empty() && prefix.back() == '/') prefix.pop_back();
  std::string host = rest, port = https ? "443" : "80";
  size_t colon = rest.rfind(':');
  if (colon != std::string::npos && rest.find(']', colon) == std::string::npos) {
    host = rest.substr(0, colon);
    port = rest.substr(colon + 1);
  }
  if (host.size() > 2 && host.front() == '[' && host.back() == ']') host = host.substr(1, host.size() - 2);  // [v6]
  addrinfo hints{}, *res = nullptr;
  hints.ai_family = AF_UNSPEC;
  hints.ai_socktype = SOCK_STREAM;
  if (getaddrinfo(host.c_str(), port.c_str(), &hints, &res) != 0 || !res) return out;
  int fd = socket(res->ai_family, res->ai_socktype, res->ai_protocol);
  if (fd >= 0) {
    timeval tv{timeout_s, 0};
    setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
    setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
    if (connect(fd, res->ai_addr, res->ai_addrlen) != 0) {
      close(fd);
      fd = -1;
    }
  }
  freeaddrinfo(res);
  if (host_out) *host_out = host;
  if (prefix_out) *prefix_out = prefix;
  if (fd < 0) return out;
  if (https) {
#ifdef VHTTP_TLS
    static SSL_CTX* ctx = nullptr;
    static std::mutex mu;
    static std::string ctx_ca;
    const TlsClientConfig& cfg = tls_client_config();
    {
      std::lock_guard<std::mutex> g(mu);
      if (!ctx || ctx_ca != cfg.ca_file) {  // (re)built when the trust anchor is configured
        SSL_CTX* fresh = SSL_CTX_new(TLS_client_method());
        if (fresh) {
          SSL_CTX_set_min_proto_version(fresh, TLS1_2_VERSION);
          bool ok = cfg.ca_file.empty() ? SSL_CTX_set_default_verify_paths(fresh) == 1
                                        : SSL_CTX_load_verify_locations(fresh, cfg.ca_file.c_str(), nullptr) == 1;
          if (!ok) {
            SSL_CTX_free(fresh);
            fresh = nullptr;
          }
        }
        if (fresh) {
          if (ctx) SSL_CTX_free(ctx);
          ctx = fresh;
          ctx_ca = cfg.ca_file;
        }
      }
    }
    SSL* ssl = ctx ? SSL_new(ctx) : nullptr;
    bool ok = ssl != nullptr;
    if (ok) {
      SSL_set_fd(ssl, fd);
      if (!cfg.insecure_skip_verify) {
        SSL_set_verify(ssl, SSL_VERIFY_PEER, nullptr);
        in_addr a4{};
        in6_addr a6{};
        X509_VERIFY_PARAM* vp = SSL_get0_param(ssl);
        if (inet_pton(AF_INET, host.c_str(), &a4) == 1 || inet_pton(AF_INET6, host.c_str(), &a6) == 1)
          ok = X509_VERIFY_PARAM_set1_ip_asc(vp, host.c_str()) == 1;  // the API server by cluster IP
        else
          ok = SSL_set1_host(ssl, host.c_str()) == 1 && SSL_set_tlsext_host_name(ssl, host.c_str()) == 1;
      }
      ok = ok && SSL_connect(ssl) == 1;
    }
    if (!ok) {
      if (ssl) SSL_free(ssl);
      close(fd);
      return out;
    }
    out.ssl = ssl;
#else
    close(fd);
    return out;  // built without TLS
#endif
  }
  out.fd = fd;
  return out;
}

inline bool send_all(Conn& fd, const std::string& data) {
  size_t off = 0;
  while (off < data.size()) {
    ssize_t n = conn_send(fd, data.data() + off, data.size() - off);
    if (n <= 0) return false;
    off += (size_t)n;
  }
  return true;
}

// Decodes a chunked body held completely in `payload`; stops at the first malformed chunk header
// or at a chunk that claims more bytes than there are.
inline std::string dechunk(const std::string& payload) {
  std::string dec;
  size_t p = 0;
  while (p < payload.size()) {
    size_t e = payload.find("\r\n", p);
    if (e == std::string::npos || e == p || e - p > 16) break;
    char* endp = nullptr;
    std::string hex = payload.substr(p, e - p);
    unsigned long long len = strtoull(hex.c_str(), &endp, 16);
    if (endp == hex.c_str() || len == 0) break;
    if (len > payload.size() - (e + 2)) {
      dec.append(payload, e + 2, std::string::npos);
      break;
    }
    dec.append(payload, e + 2, (size_t)len);
    p = e + 2 + (size_t)len + 2;
  }
  return dec;
}

// url = http://host:port[/prefix]; returns status (0 on connection failure)
inline int request(const std::string& method, const std::string& url, const std::string& path,
                   const std::string& body, std::string* out, const std::string& content_type = "application/json",
                   const std::string& bearer = "") {
  std::string host, prefix;
  Conn fd = connect_url(url, &host, &prefix, 10);
  if (fd.fd < 0) return 0;
  std::string req = method + " " + prefix + path + " HTTP/1.1\r\nHost: " + host + "\r\nConnection: close\r\n";
  if (!bearer.empty()) req += "Authorization: Bearer " + bearer + "\r\n";
  if (!body.empty() || method == "POST" || method == "PUT" || method == "PATCH")
    req += "Content-Type: " + content_type + "\r\nContent-Length: " + std::to_string(body.size()) + "\r\n";
  req += "\r\n" + body;
  if (!send_all(fd, req)) {
    conn_close(fd);
    return 0;
  }
  std::string resp;
  char tmp[65536];
  ssize_t n;
  while ((n = conn_recv(fd, tmp, sizeof tmp)) > 0) {
    resp.append(tmp, (size_t)n);
    if (resp.size() > kMaxBodyBytes + kMaxHeaderBytes) break;
  }
  conn_close(fd);
  size_t he = resp.find("\r\n\r\n");
  if (he == std::string::npos || resp.size() < 12 || resp.compare(0, 5, "HTTP/") != 0) return 0;
  int status = atoi(resp.c_str() + 9);
  std::string head = lower(resp.substr(0, he));
  std::string payload = resp.substr(he + 4);
  if (header_value(head, "transfer-encoding").find("chunked") != std::string::npos) payload = dechunk(payload);
  if (out) *out = std::move(payload);
  return status;
}

// GET whose body is consumed line by line while it arrives (a Kubernetes watch: one JSON event
// per line, chunked, open for minutes).  `on_line` returns false to hang up.  Returns the HTTP
// status (0: could not connect / no valid response head); the call ends when the server closes
// the stream, `idle_timeout_s` passes without a byte, or a line exceeds the body limit.
inline int stream_lines(const std::string& url, const std::string& path, const std::string& bearer,
                        const std::function<bool(const std::string&)>& on_line, int idle_timeout_s = 60) {
  std::string host, prefix;
  Conn fd = connect_url(url, &host, &prefix, idle_timeout_s);
  if (fd.fd < 0) return 0;
  std::string req = "GET " + prefix + path + " HTTP/1.1\r\nHost: " + host + "\r\nConnection: close\r\nAccept: application/json\r\n";
  if (!bearer.empty()) req += "Authorization: Bearer " + bearer + "\r\n";
  req += "\r\n";
  if (!send_all(fd, req)) {
    conn_close(fd);
    return 0;
  }
  std::string buf;
  char tmp[16384];
  size_t he;
  while ((he = buf.find("\r\n\r\n")) == std::string::npos) {
    ssize_t n = conn_recv(fd, tmp, sizeof tmp);
    if (n <= 0 || buf.size() > kMaxHeaderBytes) {
      conn_close(fd);
      return 0;
    }
    buf.append(tmp, (size_t)n);
  }
  if (buf.compare(0, 5, "HTTP/") != 0 || buf.size() < 12) {
    conn_close(fd);
    return 0;
  }
  int status = atoi(buf.c_str() + 9);
  bool chunked = header_value(lower(buf.substr(0, he)), "transfer-encoding").find("chunked") != std::string::npos;
  std::string raw = buf.substr(he + 4), data;  // raw: undecoded bytes, data: decoded bytes not yet split
  bool go = status == 200, eof = false;
  while (go) {
    if (chunked) {
      // move every complete chunk from raw to data
      for (;;) {
        size_t e = raw.find("\r\n");
        if (e == std::string::npos) break;
        if (e == 0 || e > 16) {
          go = false;
          break;
        }
        char* endp = nullptr;
        std::string hex = raw.substr(0, e);
        unsigned long long len = strtoull(hex.c_str(), &endp, 16);
        if (endp == hex.c_str() || len > kMaxBodyBytes) {
          go = false;
          break;
        }
        if (len == 0) {
          eof = true;
          break;
        }
        if (raw.size() < e + 2 + len + 2) break;
        data.append(raw, e + 2, (size_t)len);
        raw.erase(0, e + 2 + (size_t)len + 2);
      }
    } else {
      data += raw;
      raw.clear();
    }
    size_t nl;
    while (go && (nl = data.find('\n')) != std::string::npos) {
      std::string line = data.substr(0, nl);
      data.erase(0, nl + 1);
      if (!line.empty() && line.back() == '\r') line.pop_back();
      if (!line.empty() && !on_line(line)) go = false;
    }
    if (!go || eof || data.size() > kMaxBodyBytes) break;
    ssize_t n = conn_recv(fd, tmp, sizeof tmp);
    if (n <= 0) break;
    raw.append(tmp, (size_t)n);
  }
  conn_close(fd);
  return status;
}

}  // namespace vhttp
