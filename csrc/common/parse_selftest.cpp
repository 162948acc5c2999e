// Self-test of the control plane's JSON parser and HTTP server/client, meant to be built with
// -fsanitize=address,undefined: what the picker and the operator read comes from the network
// (request bodies, API-server replies, watch streams), so every parser here is fed truncated,
// oversized, mis-framed and non-UTF-8 input.  Bad input may be refused; it may not crash the
// process, read or write out of bounds, overflow the stack or hit undefined behaviour.
#include <cmath>
#include <cstdio>
#include <random>
#include <thread>
#include <vector>

#include "http.hpp"
#include "json.hpp"

static int failures = 0;
#define CHECK(cond)                                                             \
  do {                                                                          \
    if (!(cond)) {                                                              \
      fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond);   \
      ++failures;                                                               \
    }                                                                           \
  } while (0)

static bool parses(const std::string& s, vjson::Value* out = nullptr) {
  try {
    vjson::Value v = vjson::parse(s);
    if (out) *out = v;
    return true;
  } catch (const std::runtime_error&) {
    return false;
  }
}

static void json_tests() {
  using vjson::Value;
  // valid documents survive parse -> dump -> parse
  const char* good[] = {"null", "true", " [ ] ", "{}", "0", "-0", "-12.5e-3", "1E+2", "\"\"",
                        "\"a\\n\\t\\\"\\\\\\/\\b\\f\\r\"", "\"\\u00e9\\u20ac\\ud83d\\ude00\"",
                        "{\"a\":{\"b\":[1,2,{\"c\":null}]},\"d\":\"x\"}", "[[[[[[[[[[1]]]]]]]]]]",
                        "{\"spec\":{\"replicas\":3,\"vllmConfig\":{\"port\":8000}}}", "123456789012345678901234567890",
                        "1e400", "-1e400", "\"\xc3\xa9 raw utf-8\""};
  for (const char* g : good) {
    Value v, w;
    CHECK(parses(g, &v));
    std::string d = vjson::dump(v);
    CHECK(parses(d, &w));
    CHECK(vjson::dump(w) == d);
  }
  // malformed documents are refused with an exception
  std::vector<std::string> bad = {"", " ", "{", "[", "}", "]", "\"", "\"\\", "\"\\u", "\"\\u12", "\"\\u12g4\"", "{\"a\":}",
                                  "{\"a\" 1}", "{\"a\":1,}", "{a:1}", "[1,]", "[1 2]", "tru", "nul", "fals", "nan", "NaN",
                                  "inf", "-inf", "Infinity", "+1", "-", "1.", ".5", "1e", "1e+", "0x10", "--1", "1 2",
                                  "{\"a\":1}}", "[1]]", "\"abc", "{\"a\":\"b", "[\"\\ud800"};
  bad.push_back(std::string(2000, '1') + "x");
  bad.push_back(std::string(1000, '9'));  // a number too long to be one
  bad.push_back(std::string(100000, '['));
  bad.push_back(std::string(100000, '[') + std::string(100000, ']'));
  {
    std::string deep;
    for (int i = 0; i < 50000; ++i) deep += "{\"a\":";
    bad.push_back(deep);
    deep += "1";
    deep += std::string(50000, '}');
    bad.push_back(deep);
  }
  for (auto& b : bad) CHECK(!parses(b));
  // nesting up to the limit is fine
  CHECK(parses(std::string(150, '[') + std::string(150, ']')));
  // surrogates: pairs combine, halves on their own become U+FFFD, nothing else is produced
  {
    Value v;
    CHECK(parses("\"\\ud83d\\ude00\"", &v) && v.str == "\xf0\x9f\x98\x80");
    CHECK(parses("\"\\ud800\"", &v) && v.str == "\xef\xbf\xbd");
    CHECK(parses("\"\\ud800\\u0041\"", &v) && v.str == "\xef\xbf\xbd" "A");
    CHECK(parses("\"\\udc00x\"", &v) && v.str == "\xef\xbf\xbd" "x");
    CHECK(parses("\"\\ud800\\ud800\\udc00\"", &v) && v.str == "\xef\xbf\xbd\xf0\x90\x80\x80");
  }
  // bytes that are not UTF-8, and NUL, inside strings and outside: no crash, and a string that
  // parsed dumps to something that parses again
  {
    std::string s = "{\"k\xff\xfe\":\"v\x80\xc0";
    s.push_back('\0');
    s += "tail\"}";
    Value v;
    if (parses(s, &v)) CHECK(parses(vjson::dump(v)));
    std::string nul("[1,\0 2]", 7);
    CHECK(!parses(nul));
  }
  // numbers: out-of-range values clamp on the way in and on the way to an integer
  {
    Value v;
    CHECK(parses("{\"p\":1e400,\"q\":-1e400,\"r\":3.99,\"s\":\"7\",\"t\":70000,\"u\":-5}", &v));
    CHECK(std::isfinite(v.n("p")) && std::isfinite(v.n("q")));
    CHECK(v.i("p", 0, 1, 65535) == 65535);
    CHECK(v.i("q", 0, 1, 65535) == 1);
    CHECK(v.i("r") == 3);
    CHECK(v.i("s", 42) == 42);        // a string is not a number
    CHECK(v.i("missing", 9) == 9);
    CHECK(v.i("t", 8000, 1, 65535) == 65535);
    CHECK(v.i("u", 1, 0, 100000) == 0);
    CHECK(v.i("p") == (int64_t(1) << 53));
    Value nan_v(std::nan(""));
    CHECK(vjson::dump(nan_v) == "null");
    Value inf_v(INFINITY);
    CHECK(vjson::dump(inf_v) == "null");
    CHECK(vjson::dump(Value(1e300)).size() > 3 && parses(vjson::dump(Value(1e300))));
    CHECK(vjson::dump(Value(-9007199254740993.0)) != "");
  }
  // path lookups on the wrong types
  {
    Value v;
    CHECK(parses("{\"spec\":\"a string\",\"list\":[1,2],\"n\":null}", &v));
    CHECK(v.s("spec.replicas.x") == "" && v.n("list.0") == 0 && v.at("n.a.b").is_null() && v.at("").is_null());
    CHECK(v.at("...").is_null() && v.at("spec.").is_null());
  }
  // mutation fuzzing: flips, cuts, splices of valid documents
  std::mt19937 rng(99);
  std::vector<std::string> seeds = {
      "{\"apiVersion\":\"production-stack.vllm.ai/v1alpha1\",\"kind\":\"VLLMRuntime\",\"metadata\":{\"name\":\"x\","
      "\"resourceVersion\":\"12\",\"finalizers\":[\"a\"]},\"spec\":{\"replicas\":2,\"model\":{\"modelURL\":\"m\","
      "\"maxModelLen\":4096,\"enableLoRA\":true},\"vllmConfig\":{\"port\":8000,\"extraArgs\":[\"--a\",\"b\"],"
      "\"env\":[{\"name\":\"A\",\"value\":\"\\u00e9\\ud83d\\ude00\"}]}},\"status\":{\"readyReplicas\":1.0e0}}",
      "{\"type\":\"MODIFIED\",\"object\":{\"metadata\":{\"name\":\"p\",\"labels\":{\"model\":\"llama\"}},\"status\":"
      "{\"podIP\":\"10.0.0.1\",\"containerStatuses\":[{\"name\":\"vllm\",\"ready\":false}]}}}",
      "{\"pods\":[{\"name\":\"a\",\"address\":\"1.2.3.4\"},{\"name\":\"b\",\"address\":\"1.2.3.5\"}],\"prompt\":\"hi\","
      "\"session\":\"s\",\"model\":\"m\"}"};
  long accepted = 0;
  for (int it = 0; it < 30000; ++it) {
    std::string s = seeds[rng() % seeds.size()];
    int edits = 1 + (int)(rng() % 4);
    for (int e = 0; e < edits && !s.empty(); ++e) {
      size_t at = rng() % s.size();
      switch (rng() % 6) {
        case 0: s[at] = (char)rng(); break;
        case 1: s.erase(at, 1 + rng() % 8); break;
        case 2: s.resize(at); break;
        case 3: s.insert(at, std::string(1 + rng() % 300, "[{\"\\u:,0-e.\xff"[rng() % 13])); break;
        case 4: s.insert(at, s.substr(rng() % s.size(), rng() % 40)); break;
        case 5: s[at] = "{}[]\",:\\"[rng() % 8]; break;
      }
    }
    vjson::Value v;
    if (parses(s, &v)) {
      ++accepted;
      std::string d = vjson::dump(v);
      CHECK(parses(d));
      (void)v.i("spec.replicas", 1, 0, 100000);
      (void)v.s("object.status.podIP");
      (void)v.at("pods").size();
    }
  }
  CHECK(accepted > 100);  // the fuzzer does reach the accepting paths too
}

static void header_tests() {
  using namespace vhttp;
  CHECK(content_length("post / http/1.1\r\nhost: x") == -1);
  CHECK(content_length("post / http/1.1\r\ncontent-length: 12") == 12);
  CHECK(content_length("post / http/1.1\r\ncontent-length:\t 12 \r\nx: y") == 12);
  CHECK(content_length("post / http/1.1\r\ncontent-length: -1") == -2);
  CHECK(content_length("post / http/1.1\r\ncontent-length: 12abc") == -2);
  CHECK(content_length("post / http/1.1\r\ncontent-length: 99999999999999999999999") == -2);
  CHECK(content_length("post / http/1.1\r\ncontent-length: 999999999999") == -2);
  CHECK(content_length("post / http/1.1\r\ncontent-length:") == -2);
  CHECK(content_length("post / http/1.1\r\ncontent-length: 0x10") == -2);
  CHECK(content_length("post / http/1.1\r\nx-content-length: 5") == -1);
  CHECK(content_length("post / http/1.1\r\nx: content-length: 5") == -1);
  CHECK(header_value("get / http/1.1\r\ntransfer-encoding: chunked", "transfer-encoding") == "chunked");
  Request r;
  CHECK(parse_head("GET /a/b?x=1&y=2 HTTP/1.1\r\nHost: h\r\nX-A:  v \r\nbroken line\r\n:novalue", r));
  CHECK(r.path == "/a/b" && query_param(r.query, "y") == "2" && query_param(r.query, "z") == "" && r.headers["x-a"] == "v ");
  const char* bad_heads[] = {"", "\r\n", "GET", "GET /", "GET / ", " / HTTP/1.1", "get / HTTP/1.1", "GET x HTTP/1.1",
                             "GET  HTTP/1.1", "GET / FTP/1.1", "G\xc3T / HTTP/1.1", "GET /\r\nHTTP/1.1", "\xff\xfe\xfd"};
  for (const char* h : bad_heads) {
    Request q;
    CHECK(!parse_head(h, q));
  }
  {
    Request q;  // a path of raw bytes is carried as it is, not interpreted
    CHECK(parse_head("GET /\xff\xfe%00?\x80 HTTP/1.1", q) && q.path == "/\xff\xfe%00");
  }
  CHECK(query_param("", "a") == "" && query_param("&&&", "a") == "" && query_param("a", "a") == "" && query_param("a=&a=2", "a") == "");
  // chunked bodies: good, cut short, lying about their size, not hex, absurdly long size line
  CHECK(dechunk("5\r\nhello\r\n0\r\n\r\n") == "hello");
  CHECK(dechunk("5\r\nhel") == "hel");
  CHECK(dechunk("ffffffffffffffff\r\nabc") == "abc");
  CHECK(dechunk("zz\r\nabc\r\n") == "");
  CHECK(dechunk("11111111111111111111111111\r\nabc") == "");
  CHECK(dechunk("\r\n\r\n") == "" && dechunk("5") == "" && dechunk("") == "");
  CHECK(dechunk("3\r\nabc\r\n3\r\nde") == "abcde");
}

// ------------------------------------------------------------------ live server under attack
static std::string raw_exchange(int port, const std::string& bytes, bool half_close = true) {
  int fd = socket(AF_INET, SOCK_STREAM, 0);
  sockaddr_in a{};
  a.sin_family = AF_INET;
  a.sin_port = htons((uint16_t)port);
  a.sin_addr.s_addr = inet_addr("127.0.0.1");
  timeval tv{5, 0};
  setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
  if (connect(fd, (sockaddr*)&a, sizeof a) != 0) {
    close(fd);
    return "<connect failed>";
  }
  size_t off = 0;
  while (off < bytes.size()) {
    ssize_t n = send(fd, bytes.data() + off, std::min<size_t>(bytes.size() - off, 1 << 16), MSG_NOSIGNAL);
    if (n <= 0) break;  // the server hung up on us: allowed
    off += (size_t)n;
  }
  if (half_close) shutdown(fd, SHUT_WR);
  std::string resp;
  char tmp[4096];
  ssize_t n;
  while ((n = recv(fd, tmp, sizeof tmp, 0)) > 0) resp.append(tmp, (size_t)n);
  close(fd);
  return resp;
}

static vhttp::Server g_server;  // outlives main's locals: connection threads are detached

static void server_tests() {
  int port = g_server.start("127.0.0.1", 0, [](const vhttp::Request& r) {
    vhttp::Response resp;
    if (r.path == "/throw") throw std::runtime_error("secret detail");
    vjson::Value v = vjson::parse(r.body.empty() ? "{}" : r.body);  // may throw: becomes a 500
    resp.body = vjson::dump(v);
    return resp;
  });
  CHECK(port > 0);
  auto status_of = [](const std::string& resp) { return resp.size() >= 12 ? atoi(resp.c_str() + 9) : 0; };
  const std::string ok = "POST /x HTTP/1.1\r\nContent-Length: 7\r\n\r\n{\"a\":1}";
  CHECK(status_of(raw_exchange(port, ok)) == 200);
  CHECK(status_of(raw_exchange(port, "GET /throw HTTP/1.1\r\n\r\n")) == 500);
  CHECK(raw_exchange(port, "GET /throw HTTP/1.1\r\n\r\n").find("secret") == std::string::npos);
  CHECK(status_of(raw_exchange(port, "POST /x HTTP/1.1\r\nContent-Length: 5\r\n\r\n{\"a\":")) == 500);  // bad JSON
  CHECK(status_of(raw_exchange(port, "BOGUS\r\n\r\n")) == 400);
  CHECK(status_of(raw_exchange(port, "get / HTTP/1.1\r\n\r\n")) == 400);
  CHECK(status_of(raw_exchange(port, std::string("\xff\xfe\x00\x01\r\n\r\n", 8))) == 400);
  // framing the server refuses to buffer: the connection is dropped, nothing is answered
  CHECK(raw_exchange(port, "POST /x HTTP/1.1\r\nContent-Length: 99999999999\r\n\r\n{}") == "");
  CHECK(raw_exchange(port, "POST /x HTTP/1.1\r\nContent-Length: -7\r\n\r\n{}") == "");
  CHECK(raw_exchange(port, "POST /x HTTP/1.1\r\nContent-Length: 1e3\r\n\r\n{}") == "");
  CHECK(raw_exchange(port, "POST /x HTTP/1.1\r\nContent-Length: 100\r\n\r\nshort") == "");  // body never arrives
  CHECK(raw_exchange(port, "POST /x HTTP/1.1\r\nHost: never finished") == "");
  CHECK(raw_exchange(port, "") == "");
  CHECK(raw_exchange(port, "GET / HTTP/1.1\r\nX: " + std::string(3u << 20, 'a')) == "");  // 3 MiB of header, no end
  // a body of 70 MiB is over the limit: refused from its Content-Length alone, before it is read
  CHECK(raw_exchange(port, "POST /x HTTP/1.1\r\nContent-Length: 73400320\r\n\r\n" + std::string(1 << 20, 'x')) == "");
  // 100000 levels of nesting in a body of legal size: a clean 500, not a stack overflow
  {
    std::string body(100000, '[');
    std::string resp = raw_exchange(port, "POST /x HTTP/1.1\r\nContent-Length: " + std::to_string(body.size()) + "\r\n\r\n" + body);
    CHECK(status_of(resp) == 500);
  }
  // pipelined requests on one connection, the second malformed
  {
    std::string resp = raw_exchange(port, ok + "NOPE\r\n\r\n" + ok);
    size_t first = resp.find("HTTP/1.1 200"), second = resp.find("HTTP/1.1 400");
    CHECK(first == 0 && second != std::string::npos && resp.find("HTTP/1.1 200", second) != std::string::npos);
  }
  // random garbage, many connections at once
  std::vector<std::thread> ts;
  for (int t = 0; t < 8; ++t)
    ts.emplace_back([port, t] {
      std::mt19937 rng(1000 + t);
      for (int i = 0; i < 60; ++i) {
        std::string junk(rng() % 3000, 0);
        for (auto& c : junk) c = (char)rng();
        if (i % 3 == 0) junk = "POST /x HTTP/1.1\r\nContent-Length: " + std::to_string(junk.size()) + "\r\n\r\n" + junk;
        if (i % 3 == 1) junk += "\r\n\r\n";
        (void)raw_exchange(port, junk);
      }
    });
  for (auto& t : ts) t.join();
  CHECK(status_of(raw_exchange(port, ok)) == 200);  // still serving
}

// ------------------------------------------------------------------ client against a hostile upstream
static void client_tests() {
  std::vector<std::string> replies = {
      "",
      "HTTP",
      "HTTP/1.1",
      "HTTP/1.1 200 OK\r\n",
      "garbage\r\n\r\nbody",
      "HTTP/1.1 abc OK\r\n\r\n{}",
      "HTTP/1.1 200 OK\r\nTransfer-Encoding: chunked\r\n\r\nffffffffffffffffffff\r\nabc",
      "HTTP/1.1 200 OK\r\nTransfer-Encoding: chunked\r\n\r\nzz\r\nabc\r\n0\r\n\r\n",
      "HTTP/1.1 200 OK\r\nTransfer-Encoding: chunked\r\n\r\n10\r\nshort",
      "HTTP/1.1 200 OK\r\nTransfer-Encoding: chunked\r\n\r\n4\r\n{\"a\"\r\n3\r\n:1}\r\n1\r\n\n\r\n0\r\n\r\n",
      "HTTP/1.1 200 OK\r\nContent-Length: 5\r\n\r\n{\"type\":\"ADDED\",\"object\":{}}\n{\"type\":\xff\xfe}\nnot json\n\n\r\n{\"type\":",
      "HTTP/1.1 200 OK\r\nX: " + std::string(2u << 20, 'h'),
      "HTTP/1.1 200 OK\r\n\r\n" + std::string(300000, 'x'),  // one endless line
      "HTTP/1.1 500 Oops\r\n\r\n{\"kind\":\"Status\"}\n",
  };
  int lfd = socket(AF_INET, SOCK_STREAM, 0);
  int one = 1;
  setsockopt(lfd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
  sockaddr_in a{};
  a.sin_family = AF_INET;
  a.sin_addr.s_addr = inet_addr("127.0.0.1");
  CHECK(bind(lfd, (sockaddr*)&a, sizeof a) == 0 && listen(lfd, 64) == 0);
  socklen_t len = sizeof a;
  getsockname(lfd, (sockaddr*)&a, &len);
  int port = ntohs(a.sin_port);
  const size_t total = replies.size() * 2;
  std::thread upstream([&] {
    for (size_t i = 0; i < total; ++i) {
      int c = accept(lfd, nullptr, nullptr);
      if (c < 0) return;
      char tmp[4096];
      (void)recv(c, tmp, sizeof tmp, 0);
      const std::string& r = replies[i % replies.size()];
      size_t off = 0;
      while (off < r.size()) {
        ssize_t n = send(c, r.data() + off, r.size() - off, MSG_NOSIGNAL);
        if (n <= 0) break;
        off += (size_t)n;
      }
      close(c);
    }
  });
  const std::string url = "http://127.0.0.1:" + std::to_string(port);
  int lines_seen = 0, parsed = 0;
  for (size_t i = 0; i < replies.size(); ++i) {
    std::string out;
    int st = vhttp::request("GET", url, "/x", "", &out);
    CHECK(st == 0 || (st >= 100 && st < 600));
    if (st == 200 && !out.empty()) {
      try {
        (void)vjson::parse(out);
      } catch (const std::runtime_error&) {
      }
    }
  }
  for (size_t i = 0; i < replies.size(); ++i) {
    int st = vhttp::stream_lines(url, "/watch", "token", [&](const std::string& line) {
      ++lines_seen;
      try {
        vjson::Value ev = vjson::parse(line);
        (void)ev.s("type");
        (void)ev.at("object.metadata.resourceVersion");
        ++parsed;
      } catch (const std::runtime_error&) {
      }
      return true;
    }, 5);
    CHECK(st == 0 || (st >= 100 && st < 600));
  }
  upstream.join();
  close(lfd);
  CHECK(lines_seen >= 4 && parsed >= 2);
  // nothing listening, unresolvable host, URL that is not one
  CHECK(vhttp::request("GET", url, "/x", "", nullptr) == 0);
  CHECK(vhttp::request("GET", "http://", "/x", "", nullptr) == 0);
  CHECK(vhttp::request("GET", "", "/x", "", nullptr) == 0);
  CHECK(vhttp::request("GET", "http://127.0.0.1:notaport/p", "/x", "", nullptr) == 0);
}

int main() {
  json_tests();
  header_tests();
  server_tests();
  client_tests();
  if (failures) {
    fprintf(stderr, "%d checks failed\n", failures);
    _exit(1);
  }
  printf("parse selftest ok\n");
  fflush(stdout);
  _exit(0);  // connection threads of the test server are detached and may still be parked in recv
}
